#!/usr/bin/env python3
"""Where does a fresh process spend its first steps?  (`bench.py --steps 3 --warmup 1` on a box that
had just been leased ran its 3 timed steps at ~25 ms each against 1.38 ms steady.)

Builds the C3 trainer exactly as bench.py does, then runs N steps ONE AT A TIME: host enqueue time, then the wait for
the device, per step, plus the device-side time of the step between two hipEvents.  Run it as the FIRST GPU process of
a gpurun call (that is the state the driver's pytest run found)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mui-deepautoencoder_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

t_start = time.perf_counter()
import numpy as np
import torch
import bench

t_import = time.perf_counter() - t_start
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 14
cfg = sys.argv[2] if len(sys.argv) > 2 else "c3"
S, E, B = bench.CONFIGS[cfg]
io = S * E
dev = torch.device("cuda", 0)
t0 = time.perf_counter()
torch.cuda.set_device(0)
torch.zeros(1, device=dev)
torch.cuda.synchronize()
t_ctx = time.perf_counter() - t0

from codae.train import HipEmbeddingTrainer
schedule = bench.square_schedule(io, 4, 4)
data, blank = bench.make_inputs(16 * B, io, S)
table = np.ones((S, io), dtype=np.uint8)
for s_ in range(S):
    table[s_, s_ * E:(s_ + 1) * E] = 0
t0 = time.perf_counter()
tr = HipEmbeddingTrainer(schedule, torch.from_numpy(data), torch.from_numpy(table), torch.from_numpy(blank.reshape(-1, 1).copy()),
                         bench.LR, bench.WD, bench.CLIP, max_batch=B, precision="bf16", device=dev)
tr.init_params(seed=0)
torch.cuda.synchronize()
t_build = time.perf_counter() - t0
g = torch.Generator(device="cpu").manual_seed(1)
perm = torch.randperm(16 * B, generator=g)
idx = [perm[(i % 16) * B:(i % 16 + 1) * B].to(torch.int32).to(dev) for i in range(n_steps)]
torch.cuda.synchronize()
rows = []
for i in range(n_steps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    tr.train_batch(idx[i], run=0)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append({"step": i, "enqueue_ms": 1e3 * (t1 - t0), "wait_ms": 1e3 * (t2 - t1), "device_ms": e0.elapsed_time(e1)})
print(json.dumps({"import_s": t_import, "context_s": t_ctx, "build_trainer_s": t_build, "config": cfg, "steps": rows}))
