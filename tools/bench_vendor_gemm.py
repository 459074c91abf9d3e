"""Context for the forward-GEMM roofline fraction: what torch's own bf16 GEMM (hipBLASLt / rocBLAS behind torch.matmul
and F.linear) reaches at the same shape on the same box, next to this build's forward kernel.
Usage: python tools/bench_vendor_gemm.py [M N K]"""
import os, sys
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
bb = b.bfloat16()
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
st = hip.current_stream()
ops = {
    "this build: codae_linear_bf16 (bias + ReLU fused)": lambda: hip.check(L.codae_linear_bf16(hip.ptr(x), hip.ptr(W), hip.ptr(b), hip.ptr(y), 0, M, N, K, 1, st)),
    "torch.matmul(x, W.t()) (no bias, no ReLU)": lambda: torch.matmul(x, W.t(), out=y),
    "torch F.linear(x, W, bias) + relu_": lambda: torch.nn.functional.linear(x, W, bb).relu_(),
}
res = {k: [] for k in ops}
for rnd in range(6):
    for name, f in ops.items():
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 20 * 1e3)
fl = 2.0 * M * N * K
for name, v in res.items():
    v = sorted(v[1:]); med = v[len(v) // 2]
    print("%-52s median %7.1f us  %6.0f TFLOP/s" % (name, med, fl / med / 1e6))

# the weight-gradient shape: dW[N][K] = dY[M][N]^T X[M][K] (fp32 accumulate; the library writes bf16 or fp32)
dy = (torch.randn(M, N, generator=g) * 1e-2).to(dev).bfloat16()
dW32 = torch.empty(N, K, device=dev)
slabs = torch.empty(8 * N * K, device=dev)
ops2 = {
    "this build: codae_wgrad_bf16 (split-K slabs + reduce, fp32 out)": lambda: hip.check(L.codae_wgrad_bf16(hip.ptr(dy), hip.ptr(x), hip.ptr(dW32), hip.ptr(slabs), slabs.numel() * 4, M, N, K, st)),
    "torch.matmul(dY.t(), X) (bf16 out)": lambda: torch.matmul(dy.t(), x),
}
res2 = {k: [] for k in ops2}
for rnd in range(6):
    for name, f in ops2.items():
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        res2[name].append(e0.elapsed_time(e1) / 20 * 1e3)
for name, v in res2.items():
    v = sorted(v[1:]); med = v[len(v) // 2]
    print("%-66s median %7.1f us  %6.0f TFLOP/s" % (name, med, fl / med / 1e6))
