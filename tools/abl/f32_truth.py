"""Who is closest to the truth at C3?  One optimizer step's weight gradients (the rows tests/test_gpu_parity.py samples) from
(a) the fp32 engine on the fp32-MFMA GEMMs, (b) the fp32 engine on the bf16-plane GEMMs (gemm_f32x3.hip), (c) the numpy fp32 oracle,
each against the SAME step evaluated in float64 (torch on the GPU).  Usage: python tools/abl/f32_truth.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mui-deepautoencoder_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from codae import hip as H
from codae.train import HipEmbeddingTrainer
from oracle import dae_oracle as O
from test_gpu_parity import _c3_problem
S, E, B, io, sched, data, blank, params, bm, idx = _c3_problem()
dev = "cuda:0"
fmask = bm[blank[idx]].astype(np.float32)
rows = [0, 511, 1029, io - 1]
# float64 truth
x = torch.tensor(data[idx], dtype=torch.float64, device=dev); m = torch.tensor(fmask, dtype=torch.float64, device=dev)
Ws = [torch.tensor(w, dtype=torch.float64, device=dev, requires_grad=True) for w, _ in params]
bs = [torch.tensor(b, dtype=torch.float64, device=dev, requires_grad=True) for _, b in params]
h = x * m
for l, (_, _, relu) in enumerate(sched):
    h = h @ Ws[l].T + bs[l]
    if relu: h = torch.relu(h)
loss = ((h - x) ** 2).mean()
loss.backward()
truth = [(W.grad[rows].cpu().numpy(), b.grad.cpu().numpy()) for W, b in zip(Ws, bs)]
print("float64 loss %.12g" % float(loss))
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
def report(name, grads):
    worst_w = max(rel(gw[k], truth[l][0][k]) for l, (gw, _) in enumerate(grads) for k in range(len(rows)))
    worst_b = max(rel(gb, truth[l][1]) for l, (_, gb) in enumerate(grads))
    per_layer = " ".join("%.1e" % max(rel(gw[k], truth[l][0][k]) for k in range(len(rows))) for l, (gw, _) in enumerate(grads))
    print("%-34s worst sampled dW row %.3e, worst db %.3e | per layer: %s" % (name, worst_w, worst_b, per_layer))
orc = O.EmbeddingTrainer(params, [r for _, _, r in sched], bench.LR, bench.WD)
orc.step(data[idx], fmask)
report("numpy fp32 oracle", [(gw[rows], gb) for gw, gb in orc.last_grads])
for mode in ("native", "x3"):
    os.environ["CODAE_F32_GEMM"] = mode
    H.check(H.lib().codae_reload_env())
    tr = HipEmbeddingTrainer(sched, torch.from_numpy(data), torch.from_numpy(bm).to(torch.uint8), torch.from_numpy(blank.reshape(-1, 1).astype(np.int32)),
                             bench.LR, bench.WD, bench.CLIP, max_batch=B, precision="f32", device=dev)
    tr.load_params(params)
    tr.train_batch(torch.tensor(idx, dtype=torch.int32, device=dev), run=0)
    eng = tr.engine
    g = [(eng.weight_grad(l)[rows].cpu().numpy(), eng.bias_grad(l).cpu().numpy()) for l in range(eng.L)]
    report("fp32 engine, CODAE_F32_GEMM=%s" % mode, g)
    o = [(gw[rows], gb) for gw, gb in orc.last_grads]
    print("   against the oracle: worst sampled dW row %.3e" % max(rel(g[l][0][k], o[l][0][k].astype(np.float64)) for l in range(eng.L) for k in range(len(rows))))
    del tr
