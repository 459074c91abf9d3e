"""Per-workgroup timeline of the fused-loss GEMM (last forward layer + MSE loss from the accumulators; CODAE_GEMM_DBG=8
build): entry, first MFMA phase, K loop done, loss arithmetic done / dy staged, dy stores retired.
Usage: python tools/timeline_loss.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "mui-deepautoencoder_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
os.environ["CODAE_GEMM_DBG"] = "8"
from codae import hip
from codae.train import HipEmbeddingTrainer
import bench
S, E, B = 3, 512, 8192
io = S * E
dev = torch.device("cuda:0")
data, blank = bench.make_inputs(4 * B, io, S)
table = np.ones((S, io), dtype=np.uint8)
for s_ in range(S):
    table[s_, s_ * E:(s_ + 1) * E] = 0
tr = HipEmbeddingTrainer(bench.square_schedule(io, 4, 4), torch.from_numpy(data), torch.from_numpy(table),
                         torch.from_numpy(blank.reshape(-1, 1).copy()), 1e-5, 1e-4, 1.0, max_batch=B, precision="bf16", device=dev)
tr.init_params(seed=0)
idx = torch.randperm(4 * B)[:B].to(torch.int32).to(dev)
eng = tr.engine
for rnd in range(4):
    batch = tr._batch(idx, 0)
    eng.step_forward_loss(batch, eng.hyper(1e-5, 1e-4, 1.0, global_rows=B))
torch.cuda.synchronize()
NWG = (B // 256) * (io // 192)
out = np.zeros((NWG, 6), dtype=np.uint64)
hip.check(hip.lib().codae_debug_gemm_timeline(out.ctypes.data, NWG))
us = (out[:, :5].astype(np.int64) - out[:, 0].astype(np.int64).min()) / 100.0
for i, n in enumerate(["entry", "first MFMA phase", "K loop done", "loss done, dy staged", "dy stores retired"]):
    print("  %-22s min %6.2f  median %6.2f  max %6.2f us after the first entry" % (n, us[:, i].min(), np.median(us[:, i]), us[:, i].max()))
d = np.diff(us, axis=1)
for i, n in enumerate(["prologue (fill)", "K loop", "gather x + loss math", "dy write-out"]):
    print("  %-22s min %6.2f  median %6.2f  max %6.2f us" % (n, d[:, i].min(), np.median(d[:, i]), d[:, i].max()))
print("loss", eng.read_scalars()[3])
