"""fwd GEMM time vs K for fixed M x N: the intercept is prologue + epilogue + launch, the slope the
K loop.  Usage: python tools/bench_gemm_k.py [cfgs]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mui-deepautoencoder_amd"))
import torch
from codae import hip
L = hip.lib()
M, N = 8192, 1536
dev = torch.device("cuda:0")
cfgs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["b", "p"]
dbgs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"]
g = torch.Generator(device="cpu").manual_seed(0)
st = hip.current_stream()
for K in (64, 1536, 3072):
    x = (torch.rand(M, K, generator=g) * 2 - 1).to(dev).bfloat16()
    W = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).to(dev).bfloat16()
    b = torch.randn(N, generator=g).to(dev)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for c, d in [(c, d) for c in cfgs for d in dbgs]:
        os.environ["CODAE_GEMM_TILE"] = c
        os.environ["CODAE_GEMM_DBG"] = d
        hip.lib().codae_reload_env()
        ts = []
        for rnd in range(5):
            for _ in range(3): hip.check(L.codae_linear_bf16(hip.ptr(x), hip.ptr(W), hip.ptr(b), hip.ptr(y), 0, M, N, K, 1, st))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): hip.check(L.codae_linear_bf16(hip.ptr(x), hip.ptr(W), hip.ptr(b), hip.ptr(y), 0, M, N, K, 1, st))
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        ts.sort()
        print("K %5d tile %s dbg %s  median %7.2f us   per K-tile %6.3f us" % (K, c, d, ts[2], ts[2] / (K / 64)))
