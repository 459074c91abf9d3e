"""Does a host-to-device copy on the compute stream slow the steps that follow it?  (tools/bench_script_loop.py: the embedding
script's epoch loop ran at 2.2-2.3 ms/step against 1.18 for the same steps from pre-built index vectors.)  C3 steps from
pre-built device index vectors, 16 per "epoch", with one small copy between epochs in several forms."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mui-deepautoencoder_amd"))
import numpy as np, torch
import bench
from codae.train import HipEmbeddingTrainer
S, E, B = 3, 512, 8192
io = S * E
dev = torch.device("cuda:0")
sched = bench.square_schedule(io, 4, 4)
data, blank = bench.make_inputs(16 * B, io, S)
table = np.ones((S, io), dtype=np.uint8)
for s in range(S): table[s, s * E:(s + 1) * E] = 0
tr = HipEmbeddingTrainer(sched, torch.from_numpy(data), torch.from_numpy(table), torch.from_numpy(blank.reshape(-1, 1).copy()), 1e-5, 1e-4, 1.0,
                         max_batch=B, precision="bf16", device=dev)
tr.init_params(seed=0)
fixed = [torch.randperm(16 * B)[:B].to(dev, torch.int32) for _ in range(16)]
for r in range(20):
    for b in fixed: tr.train_batch(b, run=0)
torch.cuda.synchronize()
pinned = torch.zeros(16 * B, dtype=torch.int32).pin_memory()
pageable = torch.zeros(16 * B, dtype=torch.int32)
dbuf = torch.zeros(16 * B, dtype=torch.int32, device=dev)
dsrc = torch.zeros(16 * B, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()


def none(): pass
def h2d_pinned_async(): dbuf.copy_(pinned, non_blocking=True)
def h2d_pageable(): dbuf.copy_(pageable)
def d2d(): dbuf.copy_(dsrc)
def h2d_pinned_side_stream():
    with torch.cuda.stream(side):
        dbuf.copy_(pinned, non_blocking=True)
def fill_kernel(): dbuf.fill_(1)
def d2h_scalar(): float(tr.engine.scalars[0].item())


for name, between in (("nothing between epochs", none), ("pinned H2D, async, compute stream", h2d_pinned_async), ("pageable H2D (synchronous)", h2d_pageable),
                      ("device-to-device copy", d2d), ("pinned H2D on another stream", h2d_pinned_side_stream), ("a fill kernel", fill_kernel),
                      ("one scalar read back (.item())", d2h_scalar), ("nothing between epochs", none)):
    for r in range(2):
        between()
        for b in fixed: tr.train_batch(b, run=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for r in range(8):
        between()
        for b in fixed: tr.train_batch(b, run=0)
    torch.cuda.synchronize()
    print("%-40s %.3f ms/step" % (name, (time.perf_counter() - t0) / 128 * 1e3))
