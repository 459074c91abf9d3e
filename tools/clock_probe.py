"""Long forward GEMM (8192 x 1536 x K, K = 24576 -> ~1 ms) in the ablation variants of the 8-wave
phase-pipelined kernel; run under `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` the counter / 8 /
duration is the shader clock the chip holds in each regime (MI355X guide, DVFS give-back)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mui-deepautoencoder_amd"))
import torch
from codae import hip
L = hip.lib()
M, N, K = 8192, 1536, 24576
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
x = (torch.rand(M, K, generator=g) * 2 - 1).to(dev).bfloat16()
W = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).to(dev).bfloat16()
b = torch.zeros(N, device=dev)
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
st = hip.current_stream()
os.environ["CODAE_GEMM_TILE"] = "q"
for d in ("0", "1", "2", "0", "1", "2"):
    os.environ["CODAE_GEMM_DBG"] = d
    hip.lib().codae_reload_env()
    for _ in range(3):
        hip.check(L.codae_linear_bf16(hip.ptr(x), hip.ptr(W), hip.ptr(b), hip.ptr(y), 0, M, N, K, 0, st))
    torch.cuda.synchronize()
print("done")
