"""How many host CPUs the process really has, and pools sized to it.  No torch / numpy import at module level: bench.py and the
scripts call cap_thread_env() BEFORE importing either, so that OpenMP / OpenBLAS / MKL create their pools at the right width
in the first place."""
import os


def host_cpu_share():
    """CPUs this process may actually burn: the smaller of its affinity mask and its cgroup's CPU quota.  On the GPU boxes of
    this pool the two differ 16-fold (256 logical CPUs visible, cpu.max = 1600000 100000): torch sizes its intra-op pool, and
    OpenBLAS its own, from the first.  Under a one-process-per-GPU launcher (LOCAL_WORLD_SIZE) the share is split between the
    ranks of the node."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()[:2]),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            with open(path) as f:
                text = f.read()
            if parse is None:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    quota, period = text.strip(), f.read().strip()
            else:
                quota, period = parse(text)
            if quota not in ("max", "-1") and int(period) > 0:
                n = min(n, max(1, int(quota) // int(period)))
            break
        except (OSError, ValueError):
            continue
    try:
        n //= max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        pass
    return max(1, n)


def fit_host_threads(limit=None):
    """Cap torch's intra-op pool and the BLAS pools at host_cpu_share() (never raises them; OMP_NUM_THREADS set lower by the
    user stays).  Why a GPU training loop cares: one torch CPU op wide enough to go parallel (torch.randperm over the train
    split once per epoch, the xavier init of a 1536^2 layer) wakes one worker per VISIBLE cpu; the workers spin after the
    region, the cgroup's 100 ms CFS quota is gone within a few ms, and the kernel then throttles EVERY thread of the container -
    the thread enqueueing kernels and the ROCm runtime's signal handlers included - until the period ends.  Measured at C3 (round 3,
    tools/bench_script_loop.py, profiles/r03_e_script_loop.txt): 2.1-6.7 ms/step for an epoch loop that draws its order with
    torch.randperm on the default 128-thread pool (4-13 CFS periods throttled per 200 steps), 1.21 with the pool capped at the
    quota (none throttled); cpu.stat of a box after a full test run: 144 of 1138 periods throttled.  The 24 ms/step "cold" bench lines of rounds
    2 and 3 were the same thing.  Returns the cap."""
    import torch
    share = host_cpu_share() if limit is None else int(limit)
    if torch.get_num_threads() > share:
        torch.set_num_threads(share)
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        if any(d.get("num_threads", 1) > share for d in threadpool_info()):
            threadpool_limits(limits=share)
    except ImportError:
        pass
    return share


def cap_thread_env(limit=None):
    """OMP_NUM_THREADS / OPENBLAS_NUM_THREADS / MKL_NUM_THREADS = host_cpu_share() unless already set: for the top of a program,
    before numpy and torch are imported (their pools are then never created wider than the quota; fit_host_threads() does
    the same to pools that already exist).  Child processes inherit it.  Returns the share."""
    share = host_cpu_share() if limit is None else int(limit)
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(var, str(share))
    return share
