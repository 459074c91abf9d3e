"""Forward GEMM (8192 x 1536 x 1536) under the timing-only ablation builds (CODAE_GEMM_DBG): 0 shipped, 1 no LDS-DMA,
2 no MFMA, 4 no epilogue stores, 16 no A-operand LDS-DMA, 32 no B-operand LDS-DMA.  One process, interleaved rounds."""
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
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
st = hip.current_stream()
def fwd(): hip.check(L.codae_linear_bf16(hip.ptr(x), hip.ptr(W), hip.ptr(b), hip.ptr(y), 0, M, N, K, 1, st))
modes = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2, 4, 16, 32]
res = {m: [] for m in modes}
for rnd in range(6):
    for m in modes:
        os.environ["CODAE_GEMM_DBG"] = str(m)
        L.codae_reload_env()
        for _ in range(3): fwd()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fwd()
        e1.record(); torch.cuda.synchronize()
        res[m].append(e0.elapsed_time(e1) / 20 * 1e3)
for m in modes:
    v = sorted(res[m][1:])
    print("dbg %2d  median %6.1f us  min %6.1f us" % (m, v[len(v) // 2], v[0]))
