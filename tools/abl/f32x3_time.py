"""Time one GEMM form of the fp32-from-bf16-planes kernel (CODAE_F32_GEMM=x3 forced) at a given shape; used with the
timing-only ablation builds of gemm_f32x3.hip (make EXTRA=-DX3_DBG=n OUT=...; CODAE_HIP_LIB=that library).
Usage: python tools/abl/f32x3_time.py [M N K]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "mui-deepautoencoder_amd"))
os.environ["CODAE_F32_GEMM"] = os.environ.get("CODAE_F32_GEMM", "x3")
import torch
from codae import hip as H
L = H.lib()
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8192, 1536, 1536)
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
dy = torch.randn(M, N, device=dev); h = torch.randn(M, K, device=dev); dx = torch.empty(M, K, device=dev)
s = H.current_stream()
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
tf = timed(lambda: H.check(L.codae_linear_f32(H.ptr(x), H.ptr(W), H.ptr(b), H.ptr(y), M, N, K, 1, s)))
td = timed(lambda: H.check(L.codae_dgrad_f32(H.ptr(dy), H.ptr(W), H.ptr(h), H.ptr(dx), M, N, K, s)))
print("%s  %d x %d x %d: forward form %.1f us, data-gradient form %.1f us" % (os.environ.get("CODAE_HIP_LIB", "shipped library").split("/")[-1], M, N, K, tf, td))
