"""Timeline of the persistent chain kernel (workgroup 0, thread 0; 100 MHz clock) from the CHAIN_ABL=9 build
(tools/abl/libcodae_ch9.so: set CODAE_HIP_LIB to it).  Prints the time between stamps: prologue pieces, then per matrix
the multiply and the epilogue."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mui-deepautoencoder_amd"))
import numpy as np, torch
import bench
from codae.train import HipEmbeddingTrainer

slots, emb, B = 3, int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
io = slots * emb
dev = torch.device("cuda:0")
schedule = bench.square_schedule(io, bench.N_IN, bench.N_OUT)
data, blank = bench.make_inputs(16 * B, io, slots)
table = np.ones((slots, io), dtype=np.uint8)
for s_ in range(slots):
    table[s_, s_ * emb:(s_ + 1) * emb] = 0
tr = HipEmbeddingTrainer(schedule, torch.from_numpy(data), torch.from_numpy(table), torch.from_numpy(blank.reshape(-1, 1).copy()),
                         bench.LR, bench.WD, bench.CLIP, max_batch=B, precision="bf16", device=dev)
tr.init_params(seed=0)
idx = torch.randperm(16 * B)[:B].to(torch.int32).to(dev)
for it in range(5):
    tr.train_batch(idx, run=0)
torch.cuda.synchronize()
raw = tr.engine.bias_parts[:2 * 66].cpu().numpy().view(np.uint64)
n = int(raw[0]); st = raw[1:1 + min(n, 64)].astype(np.int64)
d = np.diff(st) * 10          # ns
L = len(schedule)
names = ["loads issue", "ring start", "bias->lds", "gather+act0", "unit0 wait"]
for l in range(L): names += ["fwd %d mul" % l, "fwd %d epi" % l]
for l in range(L - 1, 0, -1): names += ["bwd %d mul" % l, "bwd %d epi" % l]
print("stamps", n, "total %.1f us" % ((st[-1] - st[0]) / 100))
for nm, v in zip(names, d): print("%-12s %6.2f us" % (nm, v / 1e3))
