"""Per-workgroup timeline of the forward-form bf16 GEMM (CODAE_GEMM_DBG=8 build of the 8-wave pipelined
kernel): where a launch's time goes between entry, first MFMA phase, end of the K loop and the output
stores.  Usage: python tools/timeline_gemm.py [K ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mui-deepautoencoder_amd"))
import numpy as np
import torch
from codae import hip
L = hip.lib()
M, N = 8192, 1536
dev = torch.device("cuda:0")
os.environ["CODAE_GEMM_DBG"] = os.environ.get("TIMELINE_DBG", "8")   # 9: + no LDS-DMA, 10: + no MFMA (ablations)
hip.lib().codae_reload_env()
g = torch.Generator(device="cpu").manual_seed(0)
st = hip.current_stream()
NWG = (M // 256) * (N // 192)
for K in [int(a) for a in sys.argv[1:]] or [1536]:
    x = (torch.rand(M, K, generator=g) * 2 - 1).to(dev).bfloat16()
    W = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).to(dev).bfloat16()
    b = torch.randn(N, generator=g).to(dev)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for rnd in range(6):
        hip.check(L.codae_linear_bf16(hip.ptr(x), hip.ptr(W), hip.ptr(b), hip.ptr(y), 0, M, N, K, 1, st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hip.check(L.codae_linear_bf16(hip.ptr(x), hip.ptr(W), hip.ptr(b), hip.ptr(y), 0, M, N, K, 1, st))
    e1.record(); torch.cuda.synchronize()
    ref = torch.relu(x.float() @ W.float().T + b)
    print("  max |y - torch fp32 reference| = %.4f (bf16 output, |y| up to %.1f)" % (float((y.float() - ref).abs().max()), float(ref.abs().max())))
    out = np.zeros((NWG, 6), dtype=np.uint64)
    hip.check(L.codae_debug_gemm_timeline(out.ctypes.data, NWG))
    t = out[:, :5].astype(np.int64)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0                       # 100 MHz
    names = ["entry", "first MFMA phase", "K loop done", "stores issued", "stores retired"]
    print("K %d: event-timed launch %.1f us; %d workgroups; XCC histogram %s" %
          (K, e0.elapsed_time(e1) * 1e3, NWG, np.bincount(out[:, 5].astype(np.int64), minlength=8).tolist()))
    for i, n in enumerate(names):
        print("  %-18s min %6.2f  median %6.2f  max %6.2f us after the first entry" % (n, us[:, i].min(), np.median(us[:, i]), us[:, i].max()))
    d = np.diff(us, axis=1)
    for i, n in enumerate(["prologue (fill)", "K loop", "epilogue to issue", "store drain"]):
        print("  %-18s min %6.2f  median %6.2f  max %6.2f us" % (n, d[:, i].min(), np.median(d[:, i]), d[:, i].max()))
    # are the 8 column tiles of a row panel on one XCC?  (bid remap in the kernel: xcd = blockIdx & 7)
    same = sum(len(set(out[p * 8:(p + 1) * 8, 5].tolist())) == 1 for p in range(NWG // 8))
    print("  blockIdx groups of 8 on a single XCC: %d of %d; blockIdx %% 8 == XCC for %d of %d" %
          (same, NWG // 8, int((out[:, 5].astype(np.int64) == np.arange(NWG) % 8).sum()), NWG))
