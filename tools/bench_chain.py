"""Chain of 10 forward GEMMs (8192 x 1536 x 1536, bias + ReLU): one full-batch chain on one stream vs
two half-batch chains on two streams (each CU then holds a workgroup of each chain when the 128x192 tile
is used).  Prints wall time per chain of 10 layers.  A third argument starts the second half-chain half a layer late, so
that one chain's output drain meets the other's K loop (r02: no gain either way: 407 vs 371-400 us per 10 layers)."""
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

def gemm(l, r0, rows, stream):
    x = acts[l][r0:r0 + rows]; y = acts[l + 1][r0:r0 + rows]
    hip.check(L.codae_linear_bf16(hip.ptr(x), hip.ptr(Ws[l]), hip.ptr(b), hip.ptr(y), 0, rows, N, K, 1,
                                  hip.C.c_void_p(stream.cuda_stream)))

s0 = torch.cuda.current_stream(); s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream()
def full():
    for l in range(LAYERS): gemm(l, 0, M, s0)
OFFSET = len(sys.argv) > 3        # third argument: start the second chain half a layer late (a K/2 GEMM in front of it)
xh = acts[0][:4096, :K // 2].contiguous(); Wh = Ws[0][:, :K // 2].contiguous(); yh = torch.empty(4096, N, device=dev, dtype=torch.bfloat16)
def halves(parts):
    streams = [s0, s1, s2][:len(parts)]
    ev = torch.cuda.Event(); ev.record(s0)
    for st in streams[1:]: st.wait_event(ev)
    if OFFSET:
        hip.check(L.codae_linear_bf16(hip.ptr(xh), hip.ptr(Wh), hip.ptr(b), hip.ptr(yh), 0, 4096, N, K // 2, 1, hip.C.c_void_p(s1.cuda_stream)))
    for l in range(LAYERS):
        for (r0, rows), st in zip(parts, streams): gemm(l, r0, rows, st)
    for st in streams[1:]:
        e = torch.cuda.Event(); e.record(st); s0.wait_event(e)

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

tiles = sys.argv[1].split(",") if len(sys.argv) > 1 else ["q", "x"]
for rnd in range(3):
    for tile in tiles:
        os.environ["CODAE_GEMM_TILE"] = tile
        hip.lib().codae_reload_env()
        print("tile %s  full batch, 1 stream : %7.1f us per 10 layers" % (tile, timeit(full)))
if len(sys.argv) > 2:
    for tile in tiles:
        os.environ["CODAE_GEMM_TILE"] = tile
        hip.lib().codae_reload_env()
        print("tile %s  2 halves, 2 streams  : %7.1f us" % (tile, timeit(lambda: halves([(0, 4096), (4096, 4096)]))))
