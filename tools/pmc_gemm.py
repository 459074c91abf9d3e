"""Run each GEMM form once per tile config (for rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mui-deepautoencoder_amd"))
import torch
from codae import hip
L = hip.lib()
M, N, K = 8192, 1536, 1536
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
x = (torch.rand(M, K, generator=g) * 2 - 1).to(dev).bfloat16()
W = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).to(dev).bfloat16()
b = torch.randn(N, generator=g).to(dev)
dy = (torch.randn(M, N, generator=g) * 1e-2).to(dev).bfloat16()
h = (torch.rand(M, K, generator=g) - 0.3).to(dev).bfloat16()
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16); dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
db = torch.zeros(K, device=dev); dbws = torch.zeros((M + 127) // 128 * K, device=dev); dW = torch.empty(N, K, device=dev); slabs = torch.empty(8 * N * K, device=dev)
st = hip.current_stream()
for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["c", "q"]):
    os.environ["CODAE_GEMM_TILE"] = c
    hip.lib().codae_reload_env()
    for _ in range(3):
        hip.check(L.codae_linear_bf16(hip.ptr(x), hip.ptr(W), hip.ptr(b), hip.ptr(y), 0, M, N, K, 1, st))
        hip.check(L.codae_dgrad_bf16(hip.ptr(dy), hip.ptr(W), hip.ptr(h), hip.ptr(dx), hip.ptr(db), hip.ptr(dbws), M, N, K, st))
        hip.check(L.codae_wgrad_bf16(hip.ptr(dy), hip.ptr(x), hip.ptr(dW), hip.ptr(slabs), slabs.numel() * 4, M, N, K, st))
    torch.cuda.synchronize()
