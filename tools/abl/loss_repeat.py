"""Repeat the SAME evaluation step (forward + fused loss, no update) 200 times at 3 x 256 / batch 4096 and count the distinct
values of the reported loss and metric sums: any count above 1 is a race inside the step."""
import os, sys, collections
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mui-deepautoencoder_amd"))
import numpy as np, torch
from codae.train import HipEmbeddingTrainer
from oracle import dae_oracle as O
DEV = "cuda:0"
S, E, B = 3, 256, 4096
io = S * E
rng = np.random.default_rng(77)
N = 2 * B
data = rng.random((N, io), dtype=np.float32)
sched = O.layer_schedule(io, io, 2, 2, False, "embedding")
params = O.init_params(sched, rng)
bm, _, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
mtu = rng.integers(0, S, (N, 1)).astype(np.int32)
idx = torch.tensor(rng.permutation(N)[:B], dtype=torch.int32, device=DEV)
junk = [torch.full((64 << 20,), float("nan"), device=DEV) for _ in range(8)]; del junk
tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu), 1e-3, 1e-4, 1.0,
                         max_batch=B, precision="bf16", device=DEV)
tr.load_params(params)
seen = collections.Counter()
import ctypes as C
from codae import hip
eng = tr.engine
batch = tr._batch(idx, 0)
hyper = eng.hyper(1e-3, 1e-4, 1.0, global_rows=B)
train = len(sys.argv) > 1          # any argument: the TRAINING forward (loss fused into the last GEMM) instead of the evaluation step
for i in range(200):
    eng.zero_metric_sums()
    if train:
        hip.check(hip.lib().codae_step_forward_loss(eng._h, C.byref(eng.bufs), C.byref(batch), C.byref(hyper), None, hip.current_stream()))
    else:
        tr.eval_batch(idx, run=0)
    sq, sqp, gsq, loss = eng.read_scalars()
    seen[(sq, sqp, loss)] += 1
    if train:
        # the fused-loss kernel's per-workgroup sums: the LAST 2 * 768 doubles in front of the tail padding of bias_parts
        allp = eng.bias_parts.view(torch.float64).clone()
        if i == 0: first = allp
        else:
            d = (allp != first).nonzero().flatten()
            d = d[~torch.isnan(allp[d]) | ~torch.isnan(first[d])]
            if d.numel() and globals().setdefault("shown", 0) < 6:
                globals()["shown"] += 1
                print("iter", i, "differs at double index", d.tolist()[:8], "of", allp.numel(), "values", allp[d][:4].tolist(), "first", first[d][:4].tolist())
        dy = eng.dacts.view(torch.int16)
        key = (int(dy.to(torch.int64).sum()), float(eng.bias_parts.double().sum()))
        seen2 = globals().setdefault("seen2", collections.Counter()); seen2[key] += 1
print("distinct (sq_full, sq_partial, loss):", len(seen))
for k, v in seen.most_common(5): print(v, k)

if train:
    print("distinct (checksum of the dY workspace, sum of all partial-sum rows):", len(seen2))
    for k, v in seen2.most_common(4): print(v, k)
