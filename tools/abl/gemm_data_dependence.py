import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "mui-deepautoencoder_amd"))
import torch
from codae import hip
L = hip.lib()
M, N, K = 8192, 1536, 1536
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
W = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).to(dev).bfloat16()
b = torch.randn(N, generator=g).to(dev)
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
st = hip.current_stream()
xs = {"dense U(-1,1)": (torch.rand(M, K, generator=g) * 2 - 1),
      "relu'd (50 % zeros)": torch.relu(torch.rand(M, K, generator=g) * 2 - 1),
      "tiny gradients 1e-6": (torch.randn(M, K, generator=g) * 1e-6),
      "all zeros": torch.zeros(M, K)}
res = {k: [] for k in xs}
xd = {k: v.to(dev).bfloat16() for k, v in xs.items()}
for rnd in range(6):
    for k, x in xd.items():
        f = lambda: hip.check(L.codae_linear_bf16(hip.ptr(x), hip.ptr(W), hip.ptr(b), hip.ptr(y), 0, M, N, K, 1, st))
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 30 * 1e3)
for k, v in res.items():
    v = sorted(v[1:]); print("%-24s median %.1f us  min %.1f" % (k, v[len(v) // 2], v[0]))
