#!/usr/bin/env python3
"""Headline benchmark: training samples/s of the DAE hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[2], SURVEY.md 8d "C3"): embedding.yaml topology — 3 slots x 512
(io 1536), z = io, 4+4 hidden layers, steep False => 10 x Linear(1536,1536); batch 8192 rows PER
GPU (weak scaling); bf16 operands / fp32 accumulate; Adam lr 1e-5, wd 1e-4, clip 1.0
(config/embedding.yaml:8-11 of the reference).  Synthetic inputs: dataset[16*B, io] U[0,1) from
default_rng(1234) scaled by (max-min); per-row blanked slot from default_rng(5678); Xavier
weights from seed 0.  The dataset, mask tables and all per-step index vectors are resident in HBM
before the timed region; a step is one call of the fused C-ABI step (gather+corrupt, 10 forward
GEMMs, MSE loss+grad+metric sums, 10 weight-gradient + 9 data-gradient GEMMs, grad-norm clip,
Adam + bf16 shadow refresh); with N > 1 the gradient buckets are all-reduced over RCCL between
backward and update, overlapped with the remaining backward GEMMs.

Presets: --config c2 (3x128, batch 1024), c3 (default: 3x512, batch 8192 = BASELINE.json's metric), c5 (6x1024,
batch 16384).  One JSON line on rank 0.  `f32_parity` (N = 1, c3 only): the same workload on the exact-fp32 engine
(the mode the rtol 1e-3 / atol 1e-5 reference replays run in), timed after the bf16 region.  `roofline`: the forward GEMM class (the 3-slot x 512 encoder GEMM), timed
with hipEvent pairs recorded around every launch of the timed region on the launch stream
(codae_profile_begin/_end); achieved = 2*M*N*K / mean launch time.  (dgrad and wgrad launches of a
layer run concurrently on two streams; their overlapping times are listed under by_kernel.)  `cpu_baseline`: the numpy
oracle (oracle/dae_oracle.py, a port of the reference's math; used here only as the thing timed)
on the host cores, bounded sample; `cpu_baseline.torch_cpu`: the same step as plain torch CPU ops
(the stack the reference runs on), timed beside it.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "mui-deepautoencoder_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

S, E, BATCH = 3, 512, 8192
N_IN, N_OUT = 4, 4
LR, WD, CLIP = 1e-5, 1e-4, 1.0
CONFIGS = {"c2": (3, 128, 1024), "c3": (3, 512, 8192), "c5": (6, 1024, 16384)}
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_F32_TFLOPS = 157.3        # dense fp32 MFMA (v_mfma_f32_32x32x2_f32): the roofline of gemm_f32.hip
# the fp32 engine's large GEMMs run on gemm_f32x3.hip: each fp32 product = six bf16 MFMA products (three bf16 planes per operand,
# fp32 accumulation), so their roofline is a sixth of the bf16 MFMA peak; CODAE_F32_GEMM=native puts them back on the fp32 MFMA
PEAK_F32_VIA_BF16_TFLOPS = PEAK_BF16_TFLOPS / 6.0


def f32_peak_tflops():
    return PEAK_F32_TFLOPS if os.environ.get("CODAE_F32_GEMM", "")[:1] == "n" else PEAK_F32_VIA_BF16_TFLOPS


def square_schedule(io, nb_in, nb_out):
    """z = io, steep False => inc 0 => (nb_in + nb_out + 2) square Linears, ReLU after all but the
    code layer and the output layer (embedding_denoising_autoencoder.py:59-126)."""
    relu = [True] * nb_in + [False] + [True] * nb_out + [False]
    return [(io, io, r) for r in relu]


def flops_per_sample(schedule):
    """2 * (3 * sum K_l N_l - K_1 N_1): forward + wgrad for every layer, dgrad for all but the first."""
    tot = sum(k * n for k, n, _ in schedule)
    return 2 * (3 * tot - schedule[0][0] * schedule[0][1])


def make_inputs(n_rows, io, slots):
    import numpy as np
    rng = np.random.default_rng(1234)
    data = rng.random((n_rows, io), dtype=np.float32)
    data /= (data.max() - data.min())
    blank = np.random.default_rng(5678).integers(0, slots, size=n_rows)
    return data, blank.astype(np.int32)


def cpu_baseline(schedule, data, blank, io, slots, B=BATCH, budget_s=18.0):
    """Time the oracle's training step (numpy, BLAS threads = host cores given to this process)."""
    import numpy as np
    from oracle import dae_oracle as O
    try:
        from threadpoolctl import threadpool_info
        threads = max([d.get("num_threads", 1) for d in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    rng = np.random.default_rng(0)
    params = O.init_params(schedule, rng)
    tr = O.EmbeddingTrainer(params, [r for _, _, r in schedule], LR, WD)
    arch = [{"size": io // slots, "position": s * (io // slots)} for s in range(slots)]
    bm, _, _ = O.corrupter_tables(arch, 1)
    x = data[:B]
    fmask = bm[blank[:B]]
    t0 = time.perf_counter()
    tr.step(x, fmask)                      # warm-up (BLAS thread pool, page faults)
    warm = time.perf_counter() - t0
    n, t_sum = 0, 0.0
    while n < 2 or (t_sum + warm < budget_s and n < 20):
        t0 = time.perf_counter()
        tr.step(x, fmask)
        t_sum += time.perf_counter() - t0
        n += 1
        if t_sum > 2 * budget_s:
            break
    out = {"value": B * n / t_sum, "unit": "samples/s", "cores": int(threads), "kind": "port",
           "sample": "%d timed steps (1 warm-up) of the numpy oracle step at the same %dx%d / batch %d workload, fp32"
                     % (n, slots, io // slots, B),
           "ms_per_step": 1e3 * t_sum / n}
    out["torch_cpu"] = torch_cpu_baseline(schedule, x, fmask, budget_s=8.0)
    return out


def torch_cpu_baseline(schedule, x_np, fmask_np, budget_s=8.0):
    """The same step as plain torch CPU ops (nn.Linear stack, MSELoss, clip_grad_norm_, Adam): the
    stack the reference itself runs on (script/train_dae_on_embedding.py:198-215), timed beside the oracle."""
    import torch
    torch.manual_seed(0)
    layers = []
    for k, n, relu in schedule:
        lin = torch.nn.Linear(k, n)
        torch.nn.init.xavier_uniform_(lin.weight)
        torch.nn.init.zeros_(lin.bias)
        layers.append(lin)
        if relu:
            layers.append(torch.nn.ReLU(inplace=True))
    model = torch.nn.Sequential(*layers)
    opt = torch.optim.Adam(model.parameters(), lr=LR, weight_decay=WD)
    crit = torch.nn.MSELoss(reduction="mean")
    x = torch.from_numpy(x_np)
    m = torch.from_numpy(fmask_np).to(torch.float32)

    def step():
        opt.zero_grad()
        y = model(x * m)
        loss = crit(x, y)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), CLIP)
        opt.step()
        return float(loss.detach())

    t0 = time.perf_counter()
    step()
    warm = time.perf_counter() - t0
    n, t_sum = 0, 0.0
    while n < 2 or (t_sum + warm < budget_s and n < 20):
        t0 = time.perf_counter()
        step()
        t_sum += time.perf_counter() - t0
        n += 1
    return {"value": x.shape[0] * n / t_sum, "unit": "samples/s", "threads": int(torch.get_num_threads()),
            "ms_per_step": 1e3 * t_sum / n, "sample": "%d timed steps (1 warm-up), torch %s CPU fp32" % (n, torch.__version__)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS), help="workload preset (SURVEY.md 8d)")
    ap.add_argument("--slots", type=int, default=None)
    ap.add_argument("--embedding", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--no-f32-parity", action="store_true", help="skip the exact-fp32 (parity mode) leg")
    ap.add_argument("--kernel-events-every", type=int, default=16,
                    help="record per-launch hipEvent pairs in every n-th timed step only (a sampled C3 step carries ~60 event "
                         "records: every 4th step cost 4 %% of the reported rate, every 16th 0.7 %%, none 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not record per-launch hipEvents in the timed region (roofline then null)")
    ap.add_argument("--buckets", type=int, default=0,
                    help="N > 1: gradient buckets of the all-reduce (0 = one per layer, SURVEY.md 8e: the first collective starts "
                         "one layer into the backward and the exposed tail is one layer's 9.4 MB; 4 with --sharded-update)")
    ap.add_argument("--sharded-update", action="store_true",
                    help="N > 1: reduce-scatter the gradient buckets, Adam on 1/N of the parameters, all-gather the bf16 shadows")
    ap.add_argument("--native-rccl", action="store_true",
                    help="N > 1: the library's own RCCL communicator issues the bucket all-reduces inside one call per step "
                         "(codae_train_step_dp; opt-in: only ever run with one rank on this build's test boxes)")
    ap.add_argument("--fwd-events-only", action="store_true", help="time only the forward GEMM class")
    ap.add_argument("--graph", action="store_true", help="replay the step from a hipGraph (single GPU; no live kernel events)")
    args = ap.parse_args()
    cs, ce, cb = CONFIGS[args.config]
    args.slots = cs if args.slots is None else args.slots
    args.embedding = ce if args.embedding is None else args.embedding
    args.batch = cb if args.batch is None else args.batch

    # thread pools no wider than the container's CPU quota, set before numpy / torch create them (codae/hostcpu.py;
    # BENCH_NO_THREAD_FIT=1: ablation - the default pools of one thread per visible cpu)
    from codae.hostcpu import cap_thread_env, fit_host_threads
    fit = os.environ.get("BENCH_NO_THREAD_FIT") != "1"
    if fit:
        cap_thread_env()
    import numpy as np
    import torch
    host_share = fit_host_threads() if fit else None      # before the first wide torch CPU op: see its docstring (CFS throttling = "cold" 24 ms steps)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (libcodae_hip.so has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # BENCH_FORCE_DIST=1: initialise the process group and run the bucketed all-reduce path even with one
    # rank (rehearsal of the multi-GPU launch on a single-GPU box)
    distributed = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        from codae.train import init_rccl_process_group
        init_rccl_process_group(dev)

    from codae.train import HipEmbeddingTrainer

    slots, emb, B = args.slots, args.embedding, args.batch
    io = slots * emb
    schedule = square_schedule(io, N_IN, N_OUT)
    n_rows = 16 * B
    data, blank = make_inputs(n_rows, io, slots)

    # Corrupter tables for k_max = 1: mask s blanks slot s; run 0 only (train_dae_on_embedding.py:198)
    table = np.ones((slots, io), dtype=np.uint8)
    for s_ in range(slots):
        table[s_, s_ * emb:(s_ + 1) * emb] = 0
    mask_to_use = torch.from_numpy(blank.reshape(-1, 1).copy())

    tr = HipEmbeddingTrainer(schedule, torch.from_numpy(data), torch.from_numpy(table), mask_to_use, LR, WD, CLIP,
                             max_batch=B, precision=args.precision, device=dev, distributed=distributed,
                             n_buckets=(args.buckets if args.buckets > 0 else None), use_graph=args.graph and not distributed,
                             sharded_update=args.sharded_update, native_dp=args.native_rccl)
    tr.init_params(seed=0)

    # per-step row indices, resident before timing: one permutation of the dataset per epoch, the
    # global batch of a step split into contiguous per-rank shards
    total_steps = args.warmup + args.steps
    g = torch.Generator(device="cpu").manual_seed(1)
    steps_per_epoch = max(1, n_rows // (B * world))
    idx_steps = []
    perm = None
    for st in range(total_steps):
        if st % steps_per_epoch == 0:
            perm = torch.randperm(n_rows, generator=g)
        o = (st % steps_per_epoch) * B * world + rank * B
        idx_steps.append(perm[o:o + B].to(torch.int32).to(dev))

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for st in range(args.warmup):
        tr.train_batch(idx_steps[st], run=0)
    barrier()
    # Ramp-up guard (untimed, on top of the W warm-up steps the caller asked for): a process whose first kernels have
    # just run is not in steady state - code objects of kernels that only the 2nd step needs, the power state after an
    # idle lease (tools/cold_start.py).  Extra steps run one at a time until two consecutive ones
    # agree within 10 %, at most MAX_EXTRA; they re-use the warm-up batches and are reported as `warmup_effective`.
    extra, prev_ms, ramp_ms = 0, None, []
    MAX_EXTRA = 12
    while extra < MAX_EXTRA:
        ts = time.perf_counter()
        tr.train_batch(idx_steps[extra % max(1, args.warmup or 1)], run=0)
        barrier()
        ms = 1e3 * (time.perf_counter() - ts)
        ramp_ms.append(ms)
        extra += 1
        if prev_ms is not None and abs(ms - prev_ms) <= 0.10 * min(ms, prev_ms) and extra >= 3:
            break
        prev_ms = ms
    # ... and a sustained burst, still untimed: the FIRST back-to-back run of a process that is the first on a freshly leased
    # box has been seen to lose ~8 ms somewhere inside (kernel durations unchanged, host enqueue 0.08 ms / step: 1.396 vs 1.236
    # ms / step over 50 steps, the next process on the same box clean) - the power management settling under its first
    # sustained load is the working explanation.  64 steps (~80 ms) absorb it; the stall detector below stays as the backstop.
    burst = 0
    if not os.environ.get("BENCH_NO_BURST"):
        burst = 64
        for i in range(burst):
            tr.train_batch(idx_steps[i % max(1, args.warmup or 1)], run=0)
        barrier()
    kernel_events = not args.no_kernel_events and not args.graph
    # at least three timed steps carry the per-launch event pairs (mean != min in by_kernel), spread over the region
    every = max(1, min(args.kernel_events_every, args.steps // 3 if args.steps >= 3 else 1))
    if kernel_events:
        # A hipEvent pair costs 2-4 us of stream time INSIDE the timed region (it breaks back-to-back dispatch).  The
        # forward launches of a step (the class the roofline is quoted on) are dependent, gap-free kernels on one
        # stream, so the engine brackets the whole run of them with ONE pair and reports elapsed / launches.  Every
        # other class gets a pair per launch; all of it only in every `every`-th step (codae_profile_stride), so the
        # timed region as a whole is perturbed by ~8 % / every.
        from codae.hip import KERNEL_CLASSES
        classes = ("gemm_fwd",) if args.fwd_events_only else KERNEL_CLASSES
        tr.engine.profile_begin(classes, max_records=64 * (args.steps // every + 2), every=every)
        if tr.dp is not None:
            tr.dp.time_waits(every)
    step_times = [] if os.environ.get("BENCH_STEP_TIMES") == "1" else None      # diagnosis only: a sync per timed step

    def timed_region(step_events=None):
        """Exactly K steps between two barriers.  -> (elapsed s, enqueue s, host time at which each of the first steps
        had been enqueued).  step_events: list that receives one end-of-step event per step (the re-timed pass only)."""
        t0 = time.perf_counter()
        marks = []
        for st in range(args.warmup, total_steps):
            tr.train_batch(idx_steps[st], run=0)
            if len(marks) < 8:
                marks.append(1e3 * (time.perf_counter() - t0))
            if step_events is not None and len(step_events) < 32:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                step_events.append(ev)
            if step_times is not None:
                torch.cuda.synchronize()
                step_times.append(1e3 * (time.perf_counter() - t0))
        enq = time.perf_counter() - t0            # host time to ENQUEUE the timed steps (no device sync yet)
        barrier()
        return time.perf_counter() - t0, enq, marks

    elapsed, enqueue_s, enq_marks = timed_region()
    prof = tr.engine.profile_end() if kernel_events else {}
    # Stall detector.  Twice (round 2's driver run, one full-suite run of round 3) a 3-step timed region took ~24 ms per
    # step on a box whose kernels ran at their usual durations inside that very region (by_kernel summed to ~2 ms / step,
    # host enqueue 0.28 ms / step): the device sat idle between launches.  Not reproduced in 12 later attempts on 5 boxes
    # (DESIGN.md section 6).  If the timed region is more than 1.25 x the ramp-up steps measured one at a time (each with its own sync) just
    # before it, the K steps are timed AGAIN (no per-launch event pairs this time, one end-of-step event per step), the second
    # pass is what `value` reports and the first is kept under `retimed`.
    retimed = None
    steady = sorted(ramp_ms[1:])[len(ramp_ms[1:]) // 2] if len(ramp_ms) > 1 else None
    if steady and world == 1 and step_times is None and 1e3 * elapsed / args.steps > 1.25 * steady:
        first = {"ms_per_step": 1e3 * elapsed / args.steps, "host_enqueue_ms_per_step": 1e3 * enqueue_s / args.steps,
                 "ramp_up_median_ms": steady}
        evs = []
        start = torch.cuda.Event(enable_timing=True)
        start.record()
        elapsed, enqueue_s, enq_marks = timed_region(evs)
        first["second_pass_step_done_device_ms"] = [start.elapsed_time(e) for e in evs]
        retimed = first
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    loss, gnorm = tr.last_loss_and_grad_norm()
    step_path = "layers (bucketed backward + collectives)" if distributed else tr.engine.step_path(B)
    dp_info = None
    if distributed:
        dp_info = {"rccl_ranks": int(dist.get_world_size()), "backend": dist.get_backend(),
                   "buckets": [list(b) for b in tr.dp.buckets], "exposed_wait_ms_per_step": tr.dp.wait_report(),
                   "sharded_update": bool(tr.dp.sharded), "native_rccl": bool(tr.dp.native)}

    # exact-fp32 (parity) mode on the same workload: the mode every reference-pinned rtol 1e-3 / atol 1e-5 replay runs in
    f32_parity = None
    if (world == 1 and not distributed and args.precision == "bf16" and args.config == "c3" and not args.no_f32_parity
            and (slots, emb, B) == CONFIGS["c3"] and not args.graph):
        del tr
        torch.cuda.empty_cache()
        tr32 = HipEmbeddingTrainer(schedule, torch.from_numpy(data), torch.from_numpy(table), mask_to_use, LR, WD, CLIP,
                                   max_batch=B, precision="f32", device=dev)
        tr32.init_params(seed=0)
        n32, w32 = 10, 3
        for st in range(w32):
            tr32.train_batch(idx_steps[st % total_steps], run=0)
        torch.cuda.synchronize()
        t32 = time.perf_counter()
        for st in range(n32):
            tr32.train_batch(idx_steps[(w32 + st) % total_steps], run=0)
        torch.cuda.synchronize()
        t32 = time.perf_counter() - t32
        fps32 = flops_per_sample(schedule)
        f32_parity = {"ms_per_step": 1e3 * t32 / n32, "samples_per_s": B * n32 / t32, "steps": n32, "warmup": w32,
                      "tflops": fps32 * B * n32 / t32 / 1e12,
                      "frac_of_157.3TF": fps32 * B * n32 / t32 / (PEAK_F32_TFLOPS * 1e12),
                      "roofline_tflops": f32_peak_tflops(), "frac": fps32 * B * n32 / t32 / (f32_peak_tflops() * 1e12),
                      "final_loss": tr32.last_loss_and_grad_norm()[0],
                      "note": "same workload, CODAE_PREC_F32 engine: the mode the reference replays at rtol 1e-3 / atol 1e-5 run "
                              "in.  Its GEMMs take fp32 operands and give fp32-accurate products from three bf16 planes per "
                              "operand / six bf16 MFMA products / fp32 accumulation (gemm_f32x3.hip; roofline = bf16 MFMA peak / 6 "
                              "= 416.7 TFLOP/s; frac_of_157.3TF > 1 = faster than the fp32 MFMA could go); CODAE_F32_GEMM=native "
                              "= v_mfma_f32_32x32x2_f32 throughout"}
        del tr32

    if rank == 0:
        fps = flops_per_sample(schedule)
        value = B * world * args.steps / elapsed
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else f32_peak_tflops()
        out = {
            "metric": "training samples/sec at 3x512-dim input, batch 8192" if (slots, emb, B) == (S, E, BATCH)
                      else "training samples/sec at %dx%d-dim input, batch %d" % (slots, emb, B),
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "warmup_effective": args.warmup + extra + burst,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "embedding.yaml topology: %d slots x %d (io %d), z=io, 4+4 layers -> 10 x Linear(%d,%d); "
                                   "batch %d rows per GPU, Adam lr 1e-5 wd 1e-4 clip 1.0, slot-blanking k=1"
                                   % (slots, emb, io, io, io, B),
                       "global_batch": B * world, "parallelism": "dp%d" % world,
                       "step_tflops_algorithmic": fps * B / 1e12,
                       "mfma_roofline_frac_whole_step": (fps * value / world) / (peak * 1e12)},
            "final_loss": loss, "final_grad_norm": gnorm,
            "host_enqueue_ms_per_step": 1e3 * enqueue_s / args.steps,
            "host_enqueue_done_ms_first_steps": enq_marks, "ramp_up_step_ms": ramp_ms,
            "host_cpu_share": host_share, "host_threads": torch.get_num_threads(),
            "step_path": step_path,
            "f32_parity": f32_parity,
            **({"step_done_ms": step_times} if step_times is not None else {}),
            **({"retimed": retimed} if retimed is not None else {}),
        }
        if dp_info is not None:
            out["data_parallel"] = dp_info
        roof = None
        if prof:
            by = {}
            sampled_steps = max(1, args.steps // every)      # (steps every, 2 every, ... carry the pairs)
            gemm_flops = 2.0 * B * io * io
            for name, ms in prof.items():
                mean_ms = float(np.mean(ms))
                by[name] = {"launches": len(ms), "mean_ms": mean_ms, "min_ms": float(np.min(ms)),
                            "ms_per_step": float(np.sum(ms)) / max(1, sampled_steps), "sampled_steps": sampled_steps}
                if name in ("gemm_fwd", "gemm_dgrad", "gemm_wgrad", "loss"):
                    by[name]["tflops"] = gemm_flops / (mean_ms * 1e-3) / 1e12
            # ("loss" in the fused bf16 step = the LAST forward GEMM with the MSE loss in its epilogue: GEMM flops
            # plus a gather of the 50 MB target rows, so it is listed apart from the plain forward launches)
            # `roofline` is quoted on the DOMINANT kernel class: the GEMM class with the largest share of the step's
            # time (by_kernel[..].ms_per_step), not the best one.  `roofline_fwd` keeps the forward class (the "3-slot x
            # 512 encoder GEMM" BASELINE.json's target names) and `roofline_step` the whole step (algorithmic flops of the
            # step / wall time per step).  Classes that run concurrently on two streams (deferred-wgrad path: none; DP
            # path: dgrad beside wgrad) have overlapping event times, which by_kernel keeps as measured.
            gemm_classes = [k for k in by if k.startswith("gemm_")]
            dom = max(gemm_classes, key=lambda k: by[k]["ms_per_step"]) if gemm_classes else max(by, key=lambda k: by[k]["ms_per_step"])
            if "chain" in by:
                dom = "chain"
            if dom == "chain":
                # narrow stack: the persistent fused chain is the step's dominant kernel.  Its arithmetic is tiny
                # (forward + data gradients); what it has to MOVE through HBM per launch: the gathered batch (fp32), every
                # layer's saved activation and activation gradient (bf16, written once), both weight shadows once.
                rows = (B + 63) // 64 * 64
                wbytes = sum(k * n for k, n, _ in schedule) * 2
                abytes = sum(rows * k for k, _, _ in schedule) * 2 + sum(rows * n for _, n, _ in schedule) * 2
                alg = B * io * 4 + abytes + 2 * wbytes
                gbs = alg / (by[dom]["mean_ms"] * 1e-3) / 1e9
                by[dom]["tflops"] = 2.0 * B * (2 * sum(k * n for k, n, _ in schedule) - schedule[0][0] * schedule[0][1]) / (by[dom]["mean_ms"] * 1e-3) / 1e12
                roof = {"bound": "hbm", "kernel": dom, "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                        "traffic": None, "traffic_source": None, "bytes_per_launch": alg, "by_kernel": by,
                        "note": "launch- and L2-latency-bound shape: every workgroup (16 batch rows) streams all weights "
                                "from L2 once per direction; neither HBM nor MFMA is near its roofline at this size"}
            # HBM-side bytes per launch of the dominant kernel: an OFFLINE measurement (two rocprofv3 --pmc passes of this
            # very command, reduced by tools/hbm_traffic.py), keyed by workload + precision; null for any other shape
            traffic, traffic_src = None, None
            wkey = "%dx%d_b%d_%s" % (slots, emb, B, args.precision)
            tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(tfile):
                try:
                    tj = json.load(open(tfile))
                    rec = tj.get("workloads", {}).get(wkey)
                    if rec is not None and dom in rec:
                        traffic = rec[dom]
                        traffic_src = "offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/hbm_traffic.json[%s] (%s)" % (
                            wkey, rec.get("_source", "?"))
                except Exception:
                    traffic = None
            if roof is None:
                roof = {"bound": "mfma", "kernel": dom, "achieved": by[dom]["tflops"], "peak": peak, "unit": "TFLOP/s",
                        "frac": by[dom]["tflops"] / peak, "traffic": traffic, "traffic_source": traffic_src,
                        "flops_per_launch": gemm_flops * by[dom].get("gemms_per_launch", 1), "by_kernel": by}
                if "gemm_fwd" in by:
                    out["roofline_fwd"] = {"bound": "mfma", "kernel": "gemm_fwd", "achieved": by["gemm_fwd"]["tflops"], "peak": peak,
                                           "unit": "TFLOP/s", "frac": by["gemm_fwd"]["tflops"] / peak}
        out["roofline"] = roof
        step_tf = fps * value / world / 1e12
        out["roofline_step"] = {"bound": "mfma", "achieved": step_tf, "peak": peak, "unit": "TFLOP/s", "frac": step_tf / peak,
                                "note": "algorithmic flops of the whole step (29 GEMMs' worth at C3) / wall time per step"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(schedule, data, blank, io, slots, B)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
