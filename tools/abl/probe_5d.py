"""DESIGN.md section 5d probe: which value goes wrong when the 64 x 64 fused-loss tile loses a contribution?

Needs a probe build of the library (CODAE_HIP_LIB=...libcodae_hip_5dN.so, built with EXTRA=-DCODAE_DBG_5D=N: bit 0 removes the
scalar-chain guard, bit 1 dumps every thread's {sq, sqp} before the cross-lane reduction).  Repeats the SAME training forward
(3 x 256, batch 4096: 768 workgroups of 64 x 64) and, whenever a workgroup's partial sum differs from the first launch, says
whether a THREAD's value differed (accumulation) or only the workgroup's total (reduction / LDS)."""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mui-deepautoencoder_amd"))
import numpy as np
import torch
from codae import hip
from codae.train import HipEmbeddingTrainer
from oracle import dae_oracle as O

DEV = "cuda:0"
S, E, B = 3, 256, 4096
N_LAUNCH = int(sys.argv[1]) if len(sys.argv) > 1 else 300
io = S * E
rng = np.random.default_rng(77)
N = 2 * B
data = rng.random((N, io), dtype=np.float32)
sched = O.layer_schedule(io, io, 2, 2, False, "embedding")
params = O.init_params(sched, rng)
bm, _, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
mtu = rng.integers(0, S, (N, 1)).astype(np.int32)
idx = torch.tensor(rng.permutation(N)[:B], dtype=torch.int32, device=DEV)
tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu), 1e-3, 1e-4, 1.0,
                         max_batch=B, precision="bf16", device=DEV)
tr.load_params(params)
eng = tr.engine
lib = hip.lib()
have_dump = hasattr(lib, "codae_debug_5d")
n_wg = (B // 64) * (io // 64)
batch = tr._batch(idx, 0)
hyper = eng.hyper(1e-3, 1e-4, 1.0, global_rows=B)


def dump():
    if not have_dump:
        return None
    buf = (C.c_float * (n_wg * 512))()
    assert lib.codae_debug_5d(buf, n_wg * 512) == 0
    return np.frombuffer(buf, dtype=np.float32).reshape(n_wg, 256, 2).copy()


first_parts = first_dump = None
bad_launches = thread_level = total_only = 0
for i in range(N_LAUNCH):
    eng.zero_metric_sums()
    hip.check(lib.codae_step_forward_loss(eng._h, C.byref(eng.bufs), C.byref(batch), C.byref(hyper), None, hip.current_stream()))
    torch.cuda.synchronize()
    allp = eng.bias_parts.view(torch.float64).clone().cpu().numpy()
    d = dump()
    if i == 0:
        first_parts, first_dump = allp, d
        continue
    diff = np.nonzero((allp != first_parts) & ~(np.isnan(allp) & np.isnan(first_parts)))[0]
    if diff.size == 0:
        continue
    bad_launches += 1
    msg = "launch %d: %d doubles differ, first at %s: %r vs %r (delta %.6g)" % (
        i, diff.size, diff[:4].tolist(), allp[diff[0]], first_parts[diff[0]], allp[diff[0]] - first_parts[diff[0]])
    if d is not None:
        w, t, c = np.nonzero(d != first_dump)
        if w.size:
            thread_level += 1
            k = 0
            msg += "; THREAD values differ: wg %d thread %d (wave %d lane %d) comp %s now %r first %r" % (
                w[k], t[k], t[k] // 64, t[k] % 64, "sq" if c[k] == 0 else "sqp", d[w[k], t[k], c[k]], first_dump[w[k], t[k], c[k]])
            msg += " (%d thread values in all)" % w.size
        else:
            total_only += 1
            msg += "; every thread's {sq, sqp} equals the first launch: the REDUCTION lost it"
    if bad_launches <= 8:
        print(msg, flush=True)
print("probe_5d: lib %s, %d launches, %d with a differing partial sum (%d with differing thread values, %d reduction-only)" % (
    os.path.basename(hip.LIB_PATH), N_LAUNCH, bad_launches, thread_level, total_only))
