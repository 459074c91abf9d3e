"""Soak / race check at the full C3 size: N steps of the fused bf16 step with the default stream choreography vs
CODAE_SINGLE_STREAM=1 (everything on one stream), no host sync inside the loop; prints the loss curves.
Usage: python tools/soak.py [steps]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mui-deepautoencoder_amd"))
import numpy as np, torch
import bench
from codae.train import HipEmbeddingTrainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
S, E, B = 3, 512, 8192
io = S * E
sched = bench.square_schedule(io, 4, 4)
data, blank = bench.make_inputs(4 * B, io, S)
dev = torch.device("cuda:0")
table = np.ones((S, io), dtype=np.uint8)
for s in range(S): table[s, s * E:(s + 1) * E] = 0
rng = np.random.default_rng(3)
order = [torch.tensor(rng.permutation(4 * B)[:B], dtype=torch.int32, device=dev) for _ in range(steps)]
curves = {}
for mode in ("single", "dual"):
    if mode == "single": os.environ["CODAE_SINGLE_STREAM"] = "1"
    else: os.environ.pop("CODAE_SINGLE_STREAM", None)
    tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(table), torch.tensor(blank.reshape(-1, 1)), 1e-4, 1e-4, 1.0,
                             max_batch=B, precision="bf16", device=str(dev))
    tr.init_params(0)
    ls = []
    for s in range(steps):
        tr.train_batch(order[s], run=0)
        if s % 50 == 49: ls.append(tr.engine.read_scalars()[3])
    curves[mode] = ls
    print(mode, " ".join("%.6f" % v for v in ls))
rel = max(abs(a - b) / abs(a) for a, b in zip(curves["single"], curves["dual"]))
print("max relative loss difference: %.2e" % rel)
