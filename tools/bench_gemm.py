"""Time the three bf16 GEMM forms at the C3 shape for each tile configuration (one process,
interleaved rounds; MI355X guide rule 24).  Usage: python tools/bench_gemm.py [M N K]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mui-deepautoencoder_amd"))
import torch
from codae import hip
L = hip.lib()
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8192, 1536, 1536)
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
x = (torch.rand(M, K, generator=g) * 2 - 1).to(dev).bfloat16()
W = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).to(dev).bfloat16()
b = torch.randn(N, generator=g).to(dev)
dy = (torch.randn(M, N, generator=g) * 1e-2).to(dev).bfloat16()
h = (torch.rand(M, K, generator=g) - 0.3).to(dev).bfloat16()
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
db = torch.zeros(K, device=dev); dbws = torch.zeros((M + 127) // 128 * K, device=dev)
dW = torch.empty(N, K, device=dev)
slabs = torch.empty(8 * N * K, device=dev)
st = hip.current_stream()
def fwd(): hip.check(L.codae_linear_bf16(hip.ptr(x), hip.ptr(W), hip.ptr(b), hip.ptr(y), 0, M, N, K, 1, st))
def dgrad(): hip.check(L.codae_dgrad_bf16(hip.ptr(dy), hip.ptr(W), hip.ptr(h), hip.ptr(dx), hip.ptr(db), hip.ptr(dbws), M, N, K, st))
def wgrad(): hip.check(L.codae_wgrad_bf16(hip.ptr(dy), hip.ptr(x), hip.ptr(dW), hip.ptr(slabs), slabs.numel() * 4, M, N, K, st))
ops = {"fwd": fwd, "dgrad": dgrad, "wgrad(+reduce)": wgrad}
cfgs = sys.argv[4].split(",") if len(sys.argv) > 4 else ["s", "q", "x"]
res = {(o, c): [] for o in ops for c in cfgs}
for rnd in range(6):
    for c in cfgs:
        os.environ["CODAE_GEMM_TILE"] = c
        hip.lib().codae_reload_env()
        for o, f in ops.items():
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            res[(o, c)].append(e0.elapsed_time(e1) / 20 * 1e3)
fl = 2.0 * M * N * K
for o in ops:
    for c in cfgs:
        v = sorted(res[(o, c)][1:])
        med = v[len(v) // 2]
        print("%-16s tile %s  median %7.1f us  min %7.1f us  %7.0f TFLOP/s (median)" % (o, c, med, v[0], fl / med / 1e6))
