"""Dependent chain of 10 forward GEMMs (8192 x 1536 x 1536): how much of the gap between an isolated launch and a layer of
the real step comes from the WEIGHTS being cold.  Variants: a different weight matrix per layer (the step: each W is read
once per step, 47 MB in all), the same matrix for every layer (always warm in L2 / Infinity Cache), and the same
activation buffers ping-ponged instead of one per layer."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mui-deepautoencoder_amd"))
import torch
from codae import hip
L = hip.lib()
M, N, K, LAYERS = 8192, 1536, 1536, 10
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
acts = [(torch.rand(M, K, generator=g)).to(dev).bfloat16() for _ in range(LAYERS + 1)]
Ws = [((torch.rand(N, K, generator=g) * 2 - 1) * 0.06).to(dev).bfloat16() for _ in range(LAYERS)]
b = torch.zeros(N, device=dev)
st = hip.current_stream()
tile = sys.argv[1] if len(sys.argv) > 1 else "q"
os.environ["CODAE_GEMM_TILE"] = tile
L.codae_reload_env()

def chain(same_w, pingpong):
    for l in range(LAYERS):
        x = acts[l % 2] if pingpong else acts[l]
        y = acts[(l + 1) % 2] if pingpong else acts[l + 1]
        W = Ws[0] if same_w else Ws[l]
        hip.check(L.codae_linear_bf16(hip.ptr(x), hip.ptr(W), hip.ptr(b), hip.ptr(y), 0, M, N, K, 1, st))

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3 / LAYERS

for rnd in range(3):
    for same_w in (False, True):
        for pingpong in (False, True):
            print("tile %s  %s, %s: %6.2f us per layer" % (tile, "same W every layer" if same_w else "10 different W   ",
                  "2 activation buffers " if pingpong else "11 activation buffers", timeit(lambda: chain(same_w, pingpong))))
