"""Forward-form GEMM y = relu(x W^T + b) at M N K for a list of forced tiles (CODAE_GEMM_TILE letters; '-' = automatic): time and
bit-comparison of the outputs.  Usage: python tools/bench_gemm_fwd.py M N K s,m,x,-"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mui-deepautoencoder_amd"))
import torch
from codae import hip
L = hip.lib()
M, N, K = (int(v) for v in sys.argv[1:4])
tiles = sys.argv[4].split(",")
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
x = (torch.rand(M, K, generator=g) * 2 - 1).to(dev).bfloat16()
W = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).to(dev).bfloat16()
b = torch.randn(N, generator=g).to(dev)
st = hip.current_stream()
outs, res = {}, {t: [] for t in tiles}
for rnd in range(5):
    for t in tiles:
        if t == "-": os.environ.pop("CODAE_GEMM_TILE", None)
        else: os.environ["CODAE_GEMM_TILE"] = t
        L.codae_reload_env()
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        def f(): hip.check(L.codae_linear_bf16(hip.ptr(x), hip.ptr(W), hip.ptr(b), hip.ptr(y), 0, M, N, K, 1, st))
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        res[t].append(e0.elapsed_time(e1) / 20 * 1e3)
        outs[t] = y.clone()
ref = outs[tiles[0]]
for t in tiles:
    v = sorted(res[t][1:])
    print("%d x %d x %d tile %s  median %6.1f us  %6.0f TFLOP/s  same bits as tile %s: %s" % (M, N, K, t, v[len(v) // 2], 2.0 * M * N * K / v[len(v) // 2] / 1e6, tiles[0], bool(torch.equal(outs[t], ref))))
