"""One C3 step of the fp32 engine with the fp32-MFMA GEMMs and with the bf16-plane GEMMs: loss, gradient norm, per-layer weight /
bias gradient differences; then bench-like training for N steps each: loss curves.  Usage: python tools/abl/f32x3_engine_cmp.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mui-deepautoencoder_amd"))
import numpy as np, torch
import bench
from codae import hip as H
from codae.train import HipEmbeddingTrainer
S, E, B = 3, 512, 8192
io = S * E
sched = bench.square_schedule(io, 4, 4)
data, blank = bench.make_inputs(2 * B, io, S)
table = np.ones((S, io), dtype=np.uint8)
for s in range(S): table[s, s * E:(s + 1) * E] = 0
dev = "cuda:0"
idx = [torch.tensor(np.random.default_rng(i).permutation(2 * B)[:B], dtype=torch.int32, device=dev) for i in range(4)]
out = {}
for mode in ("native", "x3"):
    os.environ["CODAE_F32_GEMM"] = mode
    H.check(H.lib().codae_reload_env())
    tr = HipEmbeddingTrainer(sched, torch.from_numpy(data), torch.from_numpy(table), torch.from_numpy(blank.reshape(-1, 1).astype(np.int32)),
                             bench.LR, bench.WD, bench.CLIP, max_batch=B, precision="f32", device=dev)
    tr.init_params(seed=0)
    eng = tr.engine
    tr.train_batch(idx[0], run=0)
    sc = eng.read_scalars()
    g = [(eng.weight_grad(l).clone(), eng.bias_grad(l).clone()) for l in range(eng.L)]
    losses = [sc[3]]
    for i in range(40):
        tr.train_batch(idx[(i + 1) % 4], run=0)
        losses.append(eng.read_scalars()[3])
    out[mode] = (sc, g, losses)
    print(mode, "step 0: sq %.9g sqp %.9g gsq %.9g loss %.9g" % tuple(sc), " losses", " ".join("%.6f" % v for v in losses[::8]))
    del tr
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for l in range(len(out["native"][1])):
    print("layer %d: dW rel L2 x3 vs native %.3e, db %.3e" % (l, rel(out["x3"][1][l][0], out["native"][1][l][0]), rel(out["x3"][1][l][1], out["native"][1][l][1])))
