"""First-step weight gradients of the 3 x 128 / batch 1024 stack (tests/test_gpu_parity.py::_oracle_vs_engine) from the numpy fp32 oracle,
the fp32 engine on the fp32-MFMA GEMMs and on the bf16-plane GEMMs, against the float64 gradient: relative L2 and how many elements
have the WRONG SIGN (Adam's first step moves every element by lr in the direction of that sign: a wrong sign is a 2 lr difference).
Usage: python tools/abl/grad_sign_vs_fp64.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mui-deepautoencoder_amd"))
import numpy as np, torch
from codae import hip as H
from codae.train import HipEmbeddingTrainer
from oracle import dae_oracle as O
S, E, B = 3, 128, 1024
io = S * E
rng = np.random.default_rng(42)
N = 4 * B
data = rng.random((N, io), dtype=np.float32); data = data / (data.max() - data.min())
sched = O.layer_schedule(io, io, 4, 4, False, "embedding")
params = O.init_params(sched, rng)
bm, nmr, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
mtu = np.stack([rng.permutation(S) for _ in range(N)])
idx = rng.permutation(N)[:B]
_, fmask = O.get_masks(bm, nmr, mtu, 1, idx, 0)
dev = "cuda:0"
x = torch.tensor(data[idx], dtype=torch.float64, device=dev); m = torch.tensor(fmask, dtype=torch.float64, device=dev)
Ws = [torch.tensor(w, dtype=torch.float64, device=dev, requires_grad=True) for w, _ in params]
bs = [torch.tensor(b, dtype=torch.float64, device=dev, requires_grad=True) for _, b in params]
h = x * m
for l, (_, _, relu) in enumerate(sched):
    h = h @ Ws[l].T + bs[l]
    if relu: h = torch.relu(h)
((h - x) ** 2).mean().backward()
truth = [W.grad.cpu().numpy() for W in Ws]
wd = 1e-4
def report(name, grads):
    wrong = tot = 0; num = den = 0.0
    for l, g in enumerate(grads):
        t = truth[l]
        # Adam sees g + wd * w
        ga = g.astype(np.float64) + wd * params[l][0]; ta = t + wd * params[l][0]
        wrong += int((np.sign(ga) != np.sign(ta)).sum()); tot += t.size
        num += float(((g - t) ** 2).sum()); den += float((t ** 2).sum())
    print("%-36s rel L2 of all weight gradients vs float64 %.3e; wrong sign of (g + wd w): %d of %d" % (name, (num / den) ** 0.5, wrong, tot))
orc = O.EmbeddingTrainer(params, [r for _, _, r in sched], 1e-3, wd)
orc.step(data[idx], fmask)
report("numpy fp32 oracle", [gw for gw, _ in orc.last_grads])
for mode in ("native", "x3"):
    os.environ["CODAE_F32_GEMM"] = mode; H.check(H.lib().codae_reload_env())
    tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu).to(torch.int32), 1e-3, wd, 1.0,
                             max_batch=B, precision="f32", device=dev)
    tr.load_params(params)
    tr.train_batch(torch.tensor(idx, dtype=torch.int32, device=dev), run=0)
    report("fp32 engine, CODAE_F32_GEMM=%s" % mode, [tr.engine.weight_grad(l).cpu().numpy() for l in range(tr.engine.L)])
    del tr
