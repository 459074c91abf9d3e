"""Run the 3 x 256 / batch 4096 training loop of test_two_stream_step_matches_single_stream_over_many_steps several times in
each stream mode and print a checksum of the parameters: tells a kernel-internal race (runs of ONE mode differ) from a
cross-stream ordering bug (modes differ, each reproducible)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mui-deepautoencoder_amd"))
import numpy as np, torch
from codae.train import HipEmbeddingTrainer
from oracle import dae_oracle as O
DEV = "cuda:0"
S, E, B, steps = 3, 256, 4096, 40
io = S * E
rng = np.random.default_rng(77)
N = 2 * B
data = rng.random((N, io), dtype=np.float32)
sched = O.layer_schedule(io, io, 2, 2, False, "embedding")
params = O.init_params(sched, rng)
bm, _, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
mtu = rng.integers(0, S, (N, 1)).astype(np.int32)
order = [torch.tensor(rng.permutation(N)[:B], dtype=torch.int32, device=DEV) for _ in range(steps)]
if len(sys.argv) > 1:          # poison the caching allocator's free blocks: stale partial sums etc. would show as NaN / changed bits
    junk = [torch.full((64 << 20,), float("nan"), device=DEV) for _ in range(8)]
    del junk
for rep in range(3):
    for single in (True, False):
        if single: os.environ["CODAE_SINGLE_STREAM"] = "1"
        else: os.environ.pop("CODAE_SINGLE_STREAM", None)
        tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu), 1e-3, 1e-4, 1.0,
                                 max_batch=B, precision="bf16", device=DEV)
        tr.load_params(params)
        first_bad = None
        for s in range(steps):
            tr.train_batch(order[s], run=0)
        p = tr.engine.params
        print("rep %d %s  loss %.17g  sum|p| %.17g" % (rep, "single" if single else "two   ", tr.engine.read_scalars()[3], float(p.double().abs().sum())))
