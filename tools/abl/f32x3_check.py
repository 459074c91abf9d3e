"""gemm_f32x3.hip (fp32 product from three bf16 planes per operand, six bf16 MFMA products) against gemm_f32.hip (fp32 MFMA):
error against an fp64 product and time per launch, the three GEMM forms of the step, ragged shapes and the C3 / C5 shapes.
Usage: python tools/abl/f32x3_check.py [quick]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "mui-deepautoencoder_amd"))
import numpy as np, torch
from codae import hip as H

L = H.lib()
dev = torch.device("cuda:0")
quick = len(sys.argv) > 1


def mode(name):
    os.environ["CODAE_F32_GEMM"] = name
    H.check(L.codae_reload_env())


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def f64(t): return t.double()


shapes = [(1, 1, 1), (37, 11, 11), (130, 200, 77), (300, 129, 1000), (1024, 384, 384), (2048, 1536, 1536), (8192, 1536, 1536)]
if not quick: shapes.append((16384, 6144, 6144))
ok = True
for M, N, K in shapes:
    g = torch.Generator(device="cpu").manual_seed(M + 3 * N + 7 * K)
    x = torch.randn(M, K, generator=g).to(dev); W = torch.randn(N, K, generator=g).to(dev); b = torch.randn(N, generator=g).to(dev)
    dy = torch.randn(M, N, generator=g).to(dev); h = torch.randn(M, K, generator=g).to(dev)
    big = M * N * K > 2e9
    rows = torch.arange(0, M, max(1, M // 64), device=dev) if big else torch.arange(M, device=dev)
    ref_y = torch.relu(f64(x[rows]) @ f64(W).T + f64(b))
    ref_dx = (f64(dy[rows]) @ f64(W)) * (f64(h[rows]) > 0)
    cols = torch.arange(0, N, max(1, N // 64), device=dev) if big else torch.arange(N, device=dev)
    ref_dw = f64(dy[:, cols]).T @ f64(x)
    out = {}
    for name in ("native", "x3"):
        mode(name)
        y = torch.full((M, N), float("nan"), device=dev); dx = torch.full((M, K), float("nan"), device=dev)
        dW = torch.full((N, K), float("nan"), device=dev); db = torch.empty(N, device=dev)
        s = H.current_stream()
        fwd = lambda: H.check(L.codae_linear_f32(H.ptr(x), H.ptr(W), H.ptr(b), H.ptr(y), M, N, K, 1, s))
        dg = lambda: H.check(L.codae_dgrad_f32(H.ptr(dy), H.ptr(W), H.ptr(h), H.ptr(dx), M, N, K, s))
        wg = lambda: H.check(L.codae_wgrad_f32(H.ptr(dy), H.ptr(x), H.ptr(dW), H.ptr(db), M, N, K, s))
        t = [timed(f, 3 if big else 10) for f in (fwd, dg, wg)]
        e = [float((f64(y[rows]) - ref_y).abs().max()), float((f64(dx[rows]) - ref_dx).abs().max()), float((f64(dW[cols]) - ref_dw).abs().max())]
        rms = [float((f64(y[rows]) - ref_y).pow(2).mean().sqrt()), float((f64(dx[rows]) - ref_dx).pow(2).mean().sqrt()), float((f64(dW[cols]) - ref_dw).pow(2).mean().sqrt())]
        out[name] = (t, e, rms, (y, dx, dW))
        fl = 2.0 * M * N * K
        print("%-6s M %5d N %5d K %5d | us fwd %8.1f dgrad %8.1f wgrad %8.1f | TFLOP/s %6.1f %6.1f %6.1f | max err %.2e %.2e %.2e | rms err %.2e %.2e %.2e"
              % (name, M, N, K, t[0], t[1], t[2], fl / t[0] / 1e6, fl / t[1] / 1e6, fl / t[2] / 1e6, e[0], e[1], e[2], rms[0], rms[1], rms[2]))
    for i in range(3):
        # the plane kernel may not be further from fp64 than 1.5 x the fp32-MFMA kernel (+ a floor for tiny problems)
        if out["x3"][2][i] > 1.5 * out["native"][2][i] + 1e-7:
            ok = False; print("   form %d: x3 rms error %.3e against native %.3e" % (i, out["x3"][2][i], out["native"][2][i]))
        if not torch.isfinite(out["x3"][3][i]).all():
            ok = False; print("   form %d: non-finite / unwritten output" % i)

# integer-valued operands: every product and partial sum exact -> bit-equal to the fp64 product
for M, N, K in [(256, 256, 256), (300, 129, 1000)]:
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randint(-300, 300, (M, K), generator=g).float().to(dev); W = torch.randint(-40, 40, (N, K), generator=g).float().to(dev)
    y = torch.empty(M, N, device=dev)
    mode("x3")
    H.check(L.codae_linear_f32(H.ptr(x), H.ptr(W), None, H.ptr(y), M, N, K, 0, H.current_stream()))
    exact = bool((f64(y) == f64(x) @ f64(W).T).all())
    print("integer operands %d x %d x %d: exact %s" % (M, N, K, exact))
    ok = ok and exact
mode("auto")
print("F32X3", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
