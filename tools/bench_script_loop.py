"""What the embedding script's epoch loop costs around the fused step (script/train_dae_on_embedding.py of this build), and
why: one epoch = the sampler's order (torch.randperm + an index op + a dtype cast, each wide enough for torch to run it on its
intra-op pool) then 16 fused steps at the C3 shape.  Measured with the pool at torch's default (one thread per VISIBLE cpu: 256 on
this pool's boxes) and capped at the container's CPU quota (codae.train.fit_host_threads: 16), reading the cgroup's
cpu.stat around each leg: the wide pool's workers spin past the quota, the kernel throttles every thread of the container
- the enqueueing thread and the ROCm runtime's included - and the GPU starves.
Usage: python tools/bench_script_loop.py"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mui-deepautoencoder_amd"))
import numpy as np, torch
import bench
from codae.train import HipEmbeddingTrainer, SubsetEpochSampler, host_cpu_share

S, E, B = 3, 512, 8192
io = S * E
dev = torch.device("cuda:0")
visible = torch.get_num_threads()
share = host_cpu_share()
print("torch intra-op threads by default %d, cpus visible %d, cpu quota of this container %d" % (visible, os.cpu_count(), share))
torch.set_num_threads(min(visible, share))
sched = bench.square_schedule(io, 4, 4)
data, blank = bench.make_inputs(16 * B, io, S)
table = np.ones((S, io), dtype=np.uint8)
for s in range(S): table[s, s * E:(s + 1) * E] = 0
tr = HipEmbeddingTrainer(sched, torch.from_numpy(data), torch.from_numpy(table), torch.from_numpy(blank.reshape(-1, 1).copy()), 1e-5, 1e-4, 1.0,
                         max_batch=B, precision="bf16", device=dev)
tr.init_params(seed=0)
sampler = SubsetEpochSampler(range(16 * B), B)


def throttled():
    try:
        with open("/sys/fs/cgroup/cpu.stat") as f:
            d = dict(line.split() for line in f)
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except OSError:
        return 0, 0


def epoch_host():
    for batch_indices in sampler:
        tr.train_batch(batch_indices.to(device=dev, dtype=torch.int32), run=0)


def epoch_device():
    for idx in sampler.device_batches(dev):
        tr.train_batch(idx, run=0)


fixed = [torch.randperm(16 * B)[:B].to(dev, torch.int32) for _ in range(16)]


def epoch_fixed():
    for i in range(16): tr.train_batch(fixed[i], run=0)


for _ in range(20):
    epoch_fixed()
torch.cuda.synchronize()
for threads in (share, visible, share):
    torch.set_num_threads(threads)
    for name, fn in (("pre-built device index vectors", epoch_fixed), ("sampler, per-step host -> device copy", epoch_host),
                     ("sampler.device_batches (one copy/epoch)", epoch_device)):
        fn(); torch.cuda.synchronize()
        n0, u0 = throttled()
        t0 = time.perf_counter()
        n = 12
        for _ in range(n): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (n * 16)
        n1, u1 = throttled()
        print("threads %3d  %-40s %.3f ms/step = %.2f M samples/s   cfs periods throttled +%d (%.1f ms)"
              % (threads, name, dt * 1e3, B / dt / 1e6, n1 - n0, (u1 - u0) / 1e3))
