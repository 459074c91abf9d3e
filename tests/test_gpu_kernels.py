"""Kernel-level parity on a real MI355X, through the C ABI (ctypes).

fp32 kernels: against float64 references (tolerance rtol 1e-3 / atol 1e-5 of north_star; the
observed error is ~1e-6).  bf16 kernels: exact equality on small-integer data (every product and
partial sum is exactly representable, so any wrong lane/fragment mapping shows) plus a tolerance
check on random data.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def hip():
    from codae import hip as H
    H.lib()
    return H


def dev():
    return torch.device("cuda:0")


def sync():
    torch.cuda.synchronize()


def f64(t):
    return t.detach().double().cpu().numpy()


SHAPES_F32 = [(1, 1, 1), (37, 11, 11), (64, 11, 11), (130, 200, 77), (256, 256, 256), (300, 129, 1000), (1024, 384, 384)]


@pytest.mark.parametrize("M,N,K", SHAPES_F32)
@pytest.mark.parametrize("relu", [0, 1])
def test_linear_f32(hip, M, N, K, relu):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    x = torch.randn(M, K, generator=g).to(dev())
    W = torch.randn(N, K, generator=g).to(dev())
    b = torch.randn(N, generator=g).to(dev())
    y = torch.full((M, N), float("nan"), device=dev())
    hip.check(hip.lib().codae_linear_f32(hip.ptr(x), hip.ptr(W), hip.ptr(b), hip.ptr(y), M, N, K, relu, hip.current_stream()))
    sync()
    ref = f64(x) @ f64(W).T + f64(b)
    if relu:
        ref = np.maximum(ref, 0)
    assert np.allclose(f64(y), ref, rtol=1e-3, atol=1e-4 * np.sqrt(K))
    assert np.abs(f64(y) - ref).max() < 2e-5 * K ** 0.5 * 4


@pytest.mark.parametrize("M,N,K", SHAPES_F32)
def test_dgrad_f32(hip, M, N, K):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    dy = torch.randn(M, N, generator=g).to(dev())
    W = torch.randn(N, K, generator=g).to(dev())
    h = torch.randn(M, K, generator=g).to(dev())
    dx = torch.full((M, K), float("nan"), device=dev())
    hip.check(hip.lib().codae_dgrad_f32(hip.ptr(dy), hip.ptr(W), hip.ptr(h), hip.ptr(dx), M, N, K, hip.current_stream()))
    sync()
    ref = (f64(dy) @ f64(W)) * (f64(h) > 0)
    assert np.abs(f64(dx) - ref).max() < 1e-4 * N ** 0.5
    hip.check(hip.lib().codae_dgrad_f32(hip.ptr(dy), hip.ptr(W), None, hip.ptr(dx), M, N, K, hip.current_stream()))
    sync()
    assert np.abs(f64(dx) - f64(dy) @ f64(W)).max() < 1e-4 * N ** 0.5


@pytest.mark.parametrize("M,N,K", SHAPES_F32)
def test_wgrad_f32(hip, M, N, K):
    g = torch.Generator(device="cpu").manual_seed(M * N + K)
    dy = torch.randn(M, N, generator=g).to(dev())
    x = torch.randn(M, K, generator=g).to(dev())
    dW = torch.full((N, K), float("nan"), device=dev())
    db = torch.full((N,), float("nan"), device=dev())
    hip.check(hip.lib().codae_wgrad_f32(hip.ptr(dy), hip.ptr(x), hip.ptr(dW), hip.ptr(db), M, N, K, hip.current_stream()))
    sync()
    assert np.abs(f64(dW) - f64(dy).T @ f64(x)).max() < 1e-4 * M ** 0.5
    assert np.abs(f64(db) - f64(dy).sum(0)).max() < 1e-4 * M ** 0.5


@pytest.fixture
def f32_gemm_mode(hip, monkeypatch):
    """CODAE_F32_GEMM for one test (native = fp32 MFMA, x3 = three bf16 planes per operand), restored afterwards"""
    def set_mode(mode):
        monkeypatch.setenv("CODAE_F32_GEMM", mode)
        hip.check(hip.lib().codae_reload_env())
    yield set_mode
    monkeypatch.delenv("CODAE_F32_GEMM", raising=False)
    hip.check(hip.lib().codae_reload_env())


# (M, N, K): K in whole 32-deep tiles and 16-B aligned rows are what gemm_f32x3.hip takes; ragged M / N exercise its clamped loads
SHAPES_X3 = [(128, 128, 32), (200, 132, 64), (1000, 388, 384), (513, 1536, 1536), (2048, 1536, 512)]


@pytest.mark.parametrize("M,N,K", SHAPES_X3)
def test_f32_gemm_from_bf16_planes_matches_fp64_as_closely_as_the_fp32_mfma_kernel(hip, f32_gemm_mode, M, N, K):
    """gemm_f32x3.hip (fp32 operands cut into three bf16 planes, six bf16 MFMA products, fp32 accumulation) in the three GEMM
    forms of the step - forward (bias + ReLU), data gradient (ReLU mask), weight gradient (both operands k-strided) - against
    float64, and against gemm_f32.hip on the same inputs: its rms error may not exceed the fp32-MFMA kernel's by more than 10 %
    (measured: 10-15 % BELOW it), and the north-star tolerance rtol 1e-3 / atol 1e-5 * sqrt(K) holds outright."""
    g = torch.Generator(device="cpu").manual_seed(M + 3 * N + 7 * K)
    x = torch.randn(M, K, generator=g).to(dev()); W = torch.randn(N, K, generator=g).to(dev()); b = torch.randn(N, generator=g).to(dev())
    dy = torch.randn(M, N, generator=g).to(dev()); h = torch.randn(M, K, generator=g).to(dev())
    ref = [np.maximum(f64(x) @ f64(W).T + f64(b), 0), (f64(dy) @ f64(W)) * (f64(h) > 0), f64(dy).T @ f64(x)]
    rms = {}
    for mode in ("native", "x3"):
        f32_gemm_mode(mode)
        y = torch.full((M, N), float("nan"), device=dev()); dx = torch.full((M, K), float("nan"), device=dev())
        dW = torch.full((N, K), float("nan"), device=dev()); db = torch.full((N,), float("nan"), device=dev())
        L, s = hip.lib(), hip.current_stream()
        hip.check(L.codae_linear_f32(hip.ptr(x), hip.ptr(W), hip.ptr(b), hip.ptr(y), M, N, K, 1, s))
        hip.check(L.codae_dgrad_f32(hip.ptr(dy), hip.ptr(W), hip.ptr(h), hip.ptr(dx), M, N, K, s))
        hip.check(L.codae_wgrad_f32(hip.ptr(dy), hip.ptr(x), hip.ptr(dW), hip.ptr(db), M, N, K, s))
        sync()
        got = [f64(y), f64(dx), f64(dW)]
        for a, r, depth in zip(got, ref, (K, N, M)):
            assert np.allclose(a, r, rtol=1e-3, atol=1e-5 * np.sqrt(depth)), (mode, np.abs(a - r).max())
        rms[mode] = [float(np.sqrt(np.mean((a - r) ** 2))) for a, r in zip(got, ref)]
    for form in range(3):
        assert rms["x3"][form] <= 1.1 * rms["native"][form] + 1e-9, (form, rms)


def test_f32_gemm_from_bf16_planes_is_exact_on_integers_and_deterministic(hip, f32_gemm_mode):
    """integer-valued operands (|a| < 2^24 splits exactly into three bf16 planes; every product and partial sum is exact): the
    result equals the float64 product BIT FOR BIT - any wrong plane pairing, lane or fragment mapping shows - and two launches
    give the same bits."""
    f32_gemm_mode("x3")
    for M, N, K in [(256, 256, 256), (300, 132, 992), (128, 384, 4096)]:
        g = torch.Generator(device="cpu").manual_seed(M)
        x = torch.randint(-3000, 3000, (M, K), generator=g).float().to(dev())          # 12 significant bits: planes 0 and 1 both in use
        W = torch.randint(-40, 40, (N, K), generator=g).float().to(dev())
        y = torch.empty(M, N, device=dev()); y2 = torch.empty(M, N, device=dev())
        hip.check(hip.lib().codae_linear_f32(hip.ptr(x), hip.ptr(W), None, hip.ptr(y), M, N, K, 0, hip.current_stream()))
        hip.check(hip.lib().codae_linear_f32(hip.ptr(x), hip.ptr(W), None, hip.ptr(y2), M, N, K, 0, hip.current_stream()))
        sync()
        assert np.array_equal(f64(y), f64(x) @ f64(W).T), (M, N, K)
        assert torch.equal(y, y2)


def test_f32_gemm_dispatch_keeps_unaligned_and_small_shapes_on_the_fp32_mfma_kernel(hip, f32_gemm_mode):
    """CODAE_F32_GEMM=x3 forces the plane kernel only where it is legal: K not in whole 32-deep tiles (abalone's 11-wide
    layers), rows that are not 16-byte aligned - those launches still run, on gemm_f32.hip, and give the fp64 answer."""
    f32_gemm_mode("x3")
    for M, N, K in [(37, 11, 11), (130, 200, 77), (64, 48, 30)]:
        g = torch.Generator(device="cpu").manual_seed(K)
        x = torch.randn(M, K, generator=g).to(dev()); W = torch.randn(N, K, generator=g).to(dev()); y = torch.empty(M, N, device=dev())
        hip.check(hip.lib().codae_linear_f32(hip.ptr(x), hip.ptr(W), None, hip.ptr(y), M, N, K, 0, hip.current_stream()))
        sync()
        assert np.allclose(f64(y), f64(x) @ f64(W).T, rtol=1e-4, atol=1e-4)


def test_linear_f32_full_size(hip):
    """BASELINE config 3x512, batch 8192: one encoder GEMM; checked on a row/column sample in float64
    and by linearity (f(a x1 + x2) = a f(x1) + f(x2) without bias)."""
    M, N, K = 8192, 1536, 1536
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.rand(M, K, generator=g).to(dev())
    W = ((torch.rand(N, K, generator=g) * 2 - 1) * (6 / (N + K)) ** 0.5).to(dev())
    b = torch.randn(N, generator=g).to(dev())
    y = torch.empty(M, N, device=dev())
    L = hip.lib()
    hip.check(L.codae_linear_f32(hip.ptr(x), hip.ptr(W), hip.ptr(b), hip.ptr(y), M, N, K, 0, hip.current_stream()))
    sync()
    rows = [0, 1, 127, 128, 4095, 8191]
    ref = f64(x[rows]) @ f64(W).T + f64(b)
    assert np.allclose(f64(y[rows]), ref, rtol=1e-3, atol=1e-5)
    x2 = torch.rand(M, K, generator=g).to(dev())
    y2 = torch.empty_like(y); y3 = torch.empty_like(y)
    hip.check(L.codae_linear_f32(hip.ptr(x2), hip.ptr(W), None, hip.ptr(y2), M, N, K, 0, hip.current_stream()))
    x3 = (0.5 * x + x2).contiguous()
    hip.check(L.codae_linear_f32(hip.ptr(x3), hip.ptr(W), None, hip.ptr(y3), M, N, K, 0, hip.current_stream()))
    sync()
    assert torch.allclose(y3, 0.5 * (y - b) + y2, rtol=1e-3, atol=1e-4)


# ---------------------------------------------------------------------------------------------
# bf16 MFMA kernels
# ---------------------------------------------------------------------------------------------

def ints(shape, g, lo=-3, hi=4):
    return torch.randint(lo, hi, shape, generator=g).float()


SHAPES_BF16 = [(128, 128, 64), (64, 64, 64), (200, 192, 128), (256, 384, 384), (1000, 832, 128), (8, 64, 64), (520, 448, 192)]


@pytest.fixture(params=["s", "q", "x"], ids=["tile128x128", "pipe256x192", "pipe256x192l4"])
def tile(request, monkeypatch, hip):
    """force the small / big workgroup tile of the bf16 GEMM (the library reads CODAE_GEMM_TILE at codae_reload_env)"""
    monkeypatch.setenv("CODAE_GEMM_TILE", request.param)
    hip.lib().codae_reload_env()
    yield request.param
    monkeypatch.delenv("CODAE_GEMM_TILE", raising=False)
    hip.lib().codae_reload_env()


@pytest.mark.parametrize("M,N,K", SHAPES_BF16)
@pytest.mark.parametrize("y_f32", [0, 1])
def test_linear_bf16_exact_integers(hip, tile, M, N, K, y_f32):
    g = torch.Generator(device="cpu").manual_seed(M + 2 * N + 3 * K)
    x, W, b = ints((M, K), g), ints((N, K), g), ints((N,), g)
    xb, Wb = x.to(dev()).bfloat16(), W.to(dev()).bfloat16()
    bd = b.to(dev())
    y = torch.full((M, N), float("nan"), device=dev(), dtype=torch.float32 if y_f32 else torch.bfloat16)
    hip.check(hip.lib().codae_linear_bf16(hip.ptr(xb), hip.ptr(Wb), hip.ptr(bd), hip.ptr(y), y_f32, M, N, K, 1, hip.current_stream()))
    sync()
    ref = np.maximum(f64(x) @ f64(W).T + f64(b), 0)
    if not y_f32:
        ref = f64(torch.from_numpy(ref).bfloat16())
    assert np.array_equal(f64(y.float()), ref)


@pytest.mark.parametrize("M,N,K", SHAPES_BF16 + [(3000, 1536, 1536)])
def test_linear_bf16_mid_tile_exact_integers(hip, monkeypatch, M, N, K):
    """The pipelined kernel's 128 x 192 instantiation (forward form only; taken automatically between the 64 x 64 and the
    256 x 192 regime), forced here for every shape of the list, ragged ones included."""
    monkeypatch.setenv("CODAE_GEMM_TILE", "m")
    hip.lib().codae_reload_env()
    try:
        g = torch.Generator(device="cpu").manual_seed(M + 2 * N + 3 * K)
        x, W, b = ints((M, K), g), ints((N, K), g), ints((N,), g)
        xb, Wb = x.to(dev()).bfloat16(), W.to(dev()).bfloat16()
        bd = b.to(dev())
        y = torch.full((M, N), float("nan"), device=dev(), dtype=torch.bfloat16)
        hip.check(hip.lib().codae_linear_bf16(hip.ptr(xb), hip.ptr(Wb), hip.ptr(bd), hip.ptr(y), 0, M, N, K, 1, hip.current_stream()))
        sync()
        ref = f64(torch.from_numpy(np.maximum(f64(x) @ f64(W).T + f64(b), 0)).bfloat16())
        assert np.array_equal(f64(y.float()), ref)
    finally:
        monkeypatch.delenv("CODAE_GEMM_TILE", raising=False)
        hip.lib().codae_reload_env()


@pytest.mark.parametrize("M,N,K", SHAPES_BF16)
def test_dgrad_bf16_exact_integers(hip, tile, M, N, K):
    # dx[M][K] = (dy[M][N] . W[N][K]) * [h > 0]; here the reduction dim is N (must be % 64)
    if N % 64:
        pytest.skip("reduction dim must be a multiple of 64")
    g = torch.Generator(device="cpu").manual_seed(11 * M + N + K)
    dy, W, h = ints((M, N), g, -2, 3), ints((N, K), g, -2, 3), ints((M, K), g, -1, 2)
    dyb, Wb, hb = dy.to(dev()).bfloat16(), W.to(dev()).bfloat16(), h.to(dev()).bfloat16()
    dx = torch.full((M, K), float("nan"), device=dev(), dtype=torch.bfloat16)
    db = torch.full((K,), float("nan"), device=dev())                         # overwritten, not accumulated into
    ws = torch.full(((M + 127) // 128 * K,), float("nan"), device=dev())     # partial column sums (scratch)
    hip.check(hip.lib().codae_dgrad_bf16(hip.ptr(dyb), hip.ptr(Wb), hip.ptr(hb), hip.ptr(dx), hip.ptr(db), hip.ptr(ws), M, N, K,
                                         hip.current_stream()))
    sync()
    ref = (f64(dy) @ f64(W)) * (f64(h) > 0)
    refb = f64(torch.from_numpy(ref).bfloat16())
    assert np.array_equal(f64(dx.float()), refb)
    # column sums are taken on the fp32 values before rounding to bf16
    assert np.allclose(f64(db), ref.sum(0), rtol=0, atol=1e-3)


@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (128, 128, 128), (512, 192, 128), (1024, 384, 384), (8192, 128, 832), (1536, 520, 200)])
def test_wgrad_bf16_exact_integers(hip, tile, M, N, K):
    g = torch.Generator(device="cpu").manual_seed(M + N * K)
    dy, x = ints((M, N), g, -2, 3), ints((M, K), g, -2, 3)
    dyb, xb = dy.to(dev()).bfloat16(), x.to(dev()).bfloat16()
    dW = torch.full((N, K), float("nan"), device=dev())
    slabs = torch.empty(8 * N * K, device=dev())
    hip.check(hip.lib().codae_wgrad_bf16(hip.ptr(dyb), hip.ptr(xb), hip.ptr(dW), hip.ptr(slabs), slabs.numel() * 4, M, N, K, hip.current_stream()))
    sync()
    assert np.array_equal(f64(dW), f64(dy).T @ f64(x))
    # and without the slab workspace (split-K disabled)
    dW2 = torch.full((N, K), float("nan"), device=dev())
    hip.check(hip.lib().codae_wgrad_bf16(hip.ptr(dyb), hip.ptr(xb), hip.ptr(dW2), None, 0, M, N, K, hip.current_stream()))
    sync()
    assert np.array_equal(f64(dW2), f64(dW))


def test_bf16_gemms_full_size_random(hip):
    """3x512 / batch 8192 shapes on random data; reference = float64 product of the bf16-rounded
    operands on sampled rows; error bound = fp32 accumulation + one bf16 rounding of the output."""
    M, N, K = 8192, 1536, 1536
    L = hip.lib()
    g = torch.Generator(device="cpu").manual_seed(17)
    x = torch.rand(M, K, generator=g).to(dev()).bfloat16()
    W = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).to(dev()).bfloat16()
    b = torch.randn(N, generator=g).to(dev())
    y = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    hip.check(L.codae_linear_bf16(hip.ptr(x), hip.ptr(W), hip.ptr(b), hip.ptr(y), 0, M, N, K, 0, hip.current_stream()))
    sync()
    rows = [0, 63, 64, 4097, 8191]
    ref = f64(x[rows].float()) @ f64(W.float()).T + f64(b)
    assert np.allclose(f64(y[rows].float()), ref, rtol=1e-2, atol=1e-2)
    dy = (torch.randn(M, N, generator=g) * 1e-3).to(dev()).bfloat16()
    dx = torch.empty(M, K, device=dev(), dtype=torch.bfloat16)
    hip.check(L.codae_dgrad_bf16(hip.ptr(dy), hip.ptr(W), None, hip.ptr(dx), None, None, M, N, K, hip.current_stream()))
    dW = torch.empty(N, K, device=dev())
    slabs = torch.empty(8 * N * K, device=dev())
    hip.check(L.codae_wgrad_bf16(hip.ptr(dy), hip.ptr(x), hip.ptr(dW), hip.ptr(slabs), slabs.numel() * 4, M, N, K, hip.current_stream()))
    sync()
    ref = f64(dy[rows].float()) @ f64(W.float())
    assert np.allclose(f64(dx[rows].float()), ref, rtol=1e-2, atol=1e-5)
    cols = [0, 5, 777, 1535]
    ref = f64(dy.float())[:, cols].T @ f64(x.float())
    assert np.allclose(f64(dW[cols]), ref, rtol=1e-3, atol=1e-5)


def test_bf16_gemms_c5_size_random(hip):
    """BASELINE config C5's GEMM shape (6 slots x 1024 = 6144 wide, batch 16384): the three forms on random data against
    a float64 product of the bf16-rounded operands on sampled rows / columns (the full product is 1.2 TFLOP: too much
    for the host), plus a size-independent property - linearity in the left operand."""
    M, N, K = 16384, 6144, 6144
    L = hip.lib()
    g = torch.Generator(device="cpu").manual_seed(23)
    x = torch.rand(M, K, generator=g).to(dev()).bfloat16()
    W = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.02).to(dev()).bfloat16()
    b = torch.randn(N, generator=g).to(dev())
    y = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    hip.check(L.codae_linear_bf16(hip.ptr(x), hip.ptr(W), hip.ptr(b), hip.ptr(y), 0, M, N, K, 1, hip.current_stream()))
    sync()
    rows = [0, 255, 256, 8191, 12345, 16383]
    ref = np.maximum(f64(x[rows].float()) @ f64(W.float()).T + f64(b), 0)
    assert np.allclose(f64(y[rows].float()), ref, rtol=1e-2, atol=2e-2)
    # y(2x) - b = 2 (y(x) - b) exactly in bf16 x fp32-accumulate arithmetic (scaling by 2 is exact) where ReLU is off
    y1 = torch.empty(M, N, device=dev(), dtype=torch.float32); y2 = torch.empty_like(y1)
    x2 = (x.float() * 2).bfloat16()
    hip.check(L.codae_linear_bf16(hip.ptr(x), hip.ptr(W), None, hip.ptr(y1), 1, M, N, K, 0, hip.current_stream()))
    hip.check(L.codae_linear_bf16(hip.ptr(x2), hip.ptr(W), None, hip.ptr(y2), 1, M, N, K, 0, hip.current_stream()))
    sync()
    assert torch.equal(y2, 2 * y1)
    del y1, y2, x2
    dy = (torch.randn(M, N, generator=g) * 1e-3).to(dev()).bfloat16()
    dx = torch.empty(M, K, device=dev(), dtype=torch.bfloat16)
    hip.check(L.codae_dgrad_bf16(hip.ptr(dy), hip.ptr(W), None, hip.ptr(dx), None, None, M, N, K, hip.current_stream()))
    dW = torch.empty(N, K, device=dev())
    hip.check(L.codae_wgrad_bf16(hip.ptr(dy), hip.ptr(x), hip.ptr(dW), None, 0, M, N, K, hip.current_stream()))
    sync()
    ref = f64(dy[rows].float()) @ f64(W.float())
    assert np.allclose(f64(dx[rows].float()), ref, rtol=1e-2, atol=1e-5)
    cols = [0, 191, 192, 3071, 6143]
    ref = f64(dy.float()[:, cols]).T @ f64(x.float())
    assert np.allclose(f64(dW[cols]), ref, rtol=1e-3, atol=1e-5)


# ---------------------------------------------------------------------------------------------
# elementwise kernels vs the oracle
# ---------------------------------------------------------------------------------------------

def test_corrupt_and_expand_masks(hip):
    from oracle import dae_oracle as O
    arch = [{"size": 3, "position": 0}] + [{"size": 1, "position": 3 + i} for i in range(8)]
    bm, nmr, per_k = O.corrupter_tables(arch, 2)
    rng = np.random.default_rng(0)
    N, B = 50, 37
    mtu = np.stack([rng.permutation(bm.shape[0]) for _ in range(N)])
    idx = rng.integers(0, N, B)
    masks, fmask = O.get_masks(bm, nmr, mtu, 2, idx, 3)
    ids = torch.tensor(mtu[idx, 3], dtype=torch.int32, device=dev())
    table = torch.tensor(bm, dtype=torch.uint8, device=dev())
    kof = torch.tensor(nmr, dtype=torch.int32, device=dev())
    mo = torch.empty(2, B, 11, device=dev()); fo = torch.empty(B, 11, device=dev())
    hip.check(hip.lib().codae_expand_masks(hip.ptr(ids), hip.ptr(table), hip.ptr(kof), B, 11, 2, hip.ptr(mo), hip.ptr(fo), hip.current_stream()))
    x = torch.tensor(rng.random((B, 11)), dtype=torch.float32, device=dev())
    out = torch.empty_like(x)
    hip.check(hip.lib().codae_corrupt(hip.ptr(x), hip.ptr(fo), hip.ptr(out), x.numel(), hip.current_stream()))
    sync()
    assert np.array_equal(mo[0].cpu().numpy(), masks[0]) and np.array_equal(mo[1].cpu().numpy(), masks[1])
    assert np.array_equal(fo.cpu().numpy(), fmask)
    assert np.array_equal(out.cpu().numpy(), O.corrupt(x.cpu().numpy(), fmask))


@pytest.mark.parametrize("n", [1, 5, 64, 1000, 23608320 // 8])
def test_clip_adam_matches_torch(hip, n):
    g = torch.Generator(device="cpu").manual_seed(n)
    p0 = torch.randn(n, generator=g); g0 = torch.randn(n, generator=g) * 3
    ref_p = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.Adam([ref_p], lr=1e-3, weight_decay=1e-2)
    p = p0.clone().to(dev()); m = torch.zeros(n, device=dev()); v = torch.zeros(n, device=dev())
    sc = torch.zeros(hip.S_COUNT, dtype=torch.float64, device=dev())
    for step in range(1, 4):
        grad = g0 * step
        ref_p.grad = grad.clone().double()
        total = torch.nn.utils.clip_grad_norm_([ref_p], 1.0)
        opt.step()
        gd = grad.clone().to(dev())
        hp = hip.Hyper(1e-3, 1e-2, 0.9, 0.999, 1e-8, 1.0, step, 0.0)
        hip.check(hip.lib().codae_clip_adam(hip.ptr(p), hip.ptr(gd), hip.ptr(m), hip.ptr(v), n, C.byref(hp), hip.ptr(sc), hip.current_stream()))
        sync()
        gsq = float(sc[hip.S_GRAD_SQ]) + float(sc[hip.S_GRAD_SQ_SLOTS:hip.S_GRAD_SQ_SLOTS + hip.S_N_SLOTS].sum())
        assert abs(gsq ** 0.5 - float(total)) < 1e-4 * float(total) + 1e-6
        assert np.allclose(f64(p), f64(ref_p), rtol=1e-3, atol=1e-5)


def test_mse_loss_dense(hip):
    rng = np.random.default_rng(3)
    B, io = 77, 48
    x = rng.random((B, io), dtype=np.float32); y = rng.random((B, io), dtype=np.float32)
    fm = (rng.random((B, io)) > 0.3).astype(np.float32)
    xd, yd, fd = (torch.tensor(a, device=dev()) for a in (x, y, fm))
    dy = torch.empty_like(xd); sc = torch.zeros(hip.S_COUNT, dtype=torch.float64, device=dev())
    hip.check(hip.lib().codae_mse_loss_fwd_bwd(hip.ptr(xd), hip.ptr(yd), hip.ptr(fd), hip.ptr(dy), B * io, 1.0 / (B * io), hip.ptr(sc), hip.current_stream()))
    sync()
    d = x.astype(np.float64) - y
    assert np.allclose(f64(dy), -2 * d / (B * io), rtol=1e-5, atol=1e-9)
    assert abs(float(sc[hip.S_SQ_FULL]) - (d ** 2).sum()) < 1e-3
    assert abs(float(sc[hip.S_SQ_PARTIAL]) - ((1 - fm) * d ** 2).sum()) < 1e-3
    assert abs(float(sc[hip.S_LAST_LOSS]) - (d ** 2).mean()) < 1e-6


def test_combined_criterion_kernels_match_oracle():
    """CombinedCriterion on device tensors (codae_combined_loss_fwd_bwd / _full) vs the oracle:
    loss, gradient (incl. the 1/(B s rmse) factor and softmax - onehot), monitor matrix."""
    from golden_util import Golden, close
    from codae.tool import CombinedCriterion
    from oracle import dae_oracle as O
    g = Golden("abalone_k2")
    m = g.meta
    arch = m["arch"]
    rng = np.random.default_rng(8)
    for B in (1, 37, 300):
        x = g["data"][:B]
        y = rng.standard_normal((B, 11)).astype(np.float32)
        xt = torch.tensor(x, device=dev()); yt = torch.tensor(y, device=dev(), requires_grad=True)
        crit = CombinedCriterion(arch, 2, dev(), torch.tensor(g["type_mask"]), weight=m["weight"], reduction="mean")
        loss = crit(x=xt, y=yt)
        (loss * 3.0).backward()
        assert close(float(loss), O.combined_mean(arch, m["weight"], x, y))
        assert close(yt.grad.cpu().numpy(), 3.0 * O.combined_mean_grad_y(arch, m["weight"], x, y), atol=1e-7)
        mon = CombinedCriterion(arch, 2, dev(), torch.tensor(g["type_mask"]), reduction="none")
        full = mon(xt, yt.detach(), as_numpy=True)
        assert isinstance(full, np.ndarray) and close(full, O.combined_full(arch, x, y))


def test_monitor_accumulate_matches_the_reference_accounting():
    """codae_monitor_accumulate = the per-step accounting of script/train_dae_on_abalone.py:227-236 of the reference (monitor
    criterion of the de-normalised batch, get_partial, two get_per_k, four running sums) in one launch on fp64 device tables:
    against the oracle's combined_full / get_per_k / get_partial / normalizer_undo on the abalone fixture's data, masks of
    k = 1 and k = 2 mixed, several batches accumulated, ragged batch sizes; and the same bits on a second pass."""
    from golden_util import Golden
    from codae.tool import CombinedCriterion
    from oracle import dae_oracle as O
    g = Golden("abalone_k2")
    arch = g.meta["arch"]
    nv, k_max, io = len(arch), 2, 11
    bm, per_run, _ = O.corrupter_tables(arch, k_max)
    T = O.mask_transformation(g["type_mask"], nv)
    rng = np.random.default_rng(12)
    n_onehot = arch[0]["size"]
    scale = (rng.random(io - n_onehot) * 5 + 0.5).astype(np.float32)
    dmin = rng.standard_normal(io - n_onehot).astype(np.float32)

    class Corr:
        mask_table_u8 = torch.tensor(bm).to(torch.uint8).contiguous().to(dev())
        k_of_mask_i32 = torch.tensor(per_run, dtype=torch.int32, device=dev())

    class Norm:
        pass
    Norm.scale = torch.tensor(scale); Norm.min = torch.tensor(dmin)
    mon = CombinedCriterion(arch, k_max, dev(), torch.tensor(g["type_mask"]), reduction="none")
    f = p = 0.0
    f_k = np.zeros((k_max, nv)); p_k = np.zeros((k_max, nv))
    batches = []
    for B in (1, 64, 37, 300):
        x = g["data"][rng.permutation(len(g["data"]))[:B]]
        y = (x + 0.3 * rng.standard_normal((B, io))).astype(np.float32)
        ids = rng.integers(0, len(bm), B).astype(np.int32)
        batches.append((x, y, ids))
        xu, yu = x.copy(), y.copy()
        xu[:, n_onehot:] = O.normalizer_undo(x[:, n_onehot:], scale, dmin)
        yu[:, n_onehot:] = O.normalizer_undo(y[:, n_onehot:], scale, dmin)
        loss = O.combined_full(arch, xu, yu)
        fm = bm[ids]
        masks = [fm * (per_run[ids] == k + 1)[:, None].astype(np.float32) for k in range(k_max)]
        f += float(np.sum(loss)); f_k += O.get_per_k(loss, masks, T)
        part = O.get_partial(loss, fm, T)
        p += float(np.sum(part)); p_k += O.get_per_k(part, masks, T)
    outs = []
    for _ in range(2):
        for x, y, ids in batches:
            mon.accumulate(torch.tensor(x, device=dev()), torch.tensor(y, device=dev()), torch.tensor(ids, device=dev()), Corr,
                           normalizer=Norm, first_scaled_column=n_onehot)
        outs.append(mon.accumulated())
    gf, gp, gfk, gpk = outs[0]
    assert abs(gf - f) <= 1e-5 * f and abs(gp - p) <= 1e-5 * p, (gf, f, gp, p)
    assert np.allclose(gfk, f_k, rtol=1e-5, atol=1e-6) and np.allclose(gpk, p_k, rtol=1e-5, atol=1e-6)
    assert float(gpk.sum()) > 0 and float(gfk[1].sum()) > 0                     # both k rows and the partial tables are exercised
    assert outs[1][0] == gf and outs[1][1] == gp and np.array_equal(outs[1][2], gfk) and np.array_equal(outs[1][3], gpk)
    assert mon.accumulated()[0] == 0.0                                          # reset


def test_ranking_loss_batched_never_counts_exact_copies_of_the_own_row():
    """the sample's own similarity (a wave sum) and the GEMM's columns are summed in different orders, so an
    inventory row that is an exact COPY of the sample's own row could be counted or not by an ulp; the reference never counts
    it (both values come out of one cosine_similarity call: s[idx] > s[j] is false for equal values).  The batched kernel
    skips such rows by identity (val_group).  Inventory with every validation row duplicated 3x vs the oracle."""
    from codae.tool import RankingLoss
    from oracle import dae_oracle as O
    S, E, N = 3, 32, 240
    rng = np.random.default_rng(31)
    base = rng.standard_normal((S, N // 3, E)).astype(np.float32)
    inv = np.concatenate([base, base, base], axis=1)                       # rows r, r + 80, r + 160 are identical
    val = [int(v) for v in rng.permutation(N)[:150]]
    bm, per_run, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
    mtu = rng.integers(0, S, (N, 1)).astype(np.int32)

    class DS:
        nb_predictor, nb_used_category, embedding_size = S * E, S, E
        data_per_category = {c: torch.tensor(inv[c]) for c in range(S)}

    class Corr:
        mask_table_u8 = torch.tensor(bm).to(torch.uint8).contiguous().to(dev())
        mask_to_use_i32 = torch.tensor(mtu).contiguous().to(dev())
    rl = RankingLoss(DS(), val, device=dev())
    idx = np.asarray(val[:96])
    # predictions close to the own row: the copies' similarities equal the own one to the last bit or differ by an ulp
    pred = np.concatenate([inv[c][idx] for c in range(S)], axis=1) + 1e-3 * rng.standard_normal((len(idx), S * E)).astype(np.float32)
    pred = pred.astype(np.float32)
    _, fm = O.get_masks(bm, per_run, mtu, 1, idx, 0)
    ref = O.ranking_loss(pred, fm, idx, [inv[c] for c in range(S)], E, val)
    rl.add(torch.tensor(pred, device=dev()), torch.tensor(idx, dtype=torch.int32, device=dev()), Corr, run=0, chunk=64)
    got = rl.total()
    assert rl._val_group is not None
    assert abs(got - ref) <= 1e-9 + 1e-6 * abs(ref), (got, ref)            # no slack for flipped comparisons: there are none


def test_ranking_loss_kernel_matches_oracle():
    from golden_util import Golden
    from codae.tool import RankingLoss
    from oracle import dae_oracle as O
    ge = Golden("embedding_square")

    class DS:
        nb_predictor, nb_used_category, embedding_size = 48, 3, 16
        data_per_category = {c: torch.tensor(ge["data_per_category"][c]) for c in range(3)}
    val = [int(v) for v in ge["validation_indices"]]
    rl = RankingLoss(DS(), val, device=dev())
    rng = np.random.default_rng(4)
    for call in (6, 7):
        idx, run = ge.calls()[call]
        _, fm = O.get_masks(ge["binary_masks"], ge["nb_missing_per_run"], ge["mask_to_use"], 1, idx, run)
        pred = rng.standard_normal((len(idx), 48)).astype(np.float32)
        got = rl.get(torch.tensor(pred, device=dev()), torch.tensor(fm, device=dev()), tuple(int(i) for i in idx))
        ref = O.ranking_loss(pred, fm, idx, list(ge["data_per_category"]), 16, val)
        assert abs(got - ref) <= 1e-3 * abs(ref) + 2.0 / (len(val) - 1), (got, ref)   # a near-tie may flip one comparison


@pytest.mark.gpu
def test_ranking_loss_batched_gemm_matches_oracle_and_item_kernel():
    """RankingLoss as exact-fp32 MFMA GEMMs over a whole validation batch, masks from the Corrupter's device tables,
    accumulated on the device: against the oracle on the reference-generated fixture, and against the
    wave-per-item kernel on a larger problem that spans several validation chunks and all slots.  A near-tie may flip
    one comparison per sample (different fp32 summation order), hence 2 / (V - 1) of slack per call."""
    from golden_util import Golden
    from codae.tool import RankingLoss
    from oracle import dae_oracle as O
    ge = Golden("embedding_square")

    class DS:
        nb_predictor, nb_used_category, embedding_size = 48, 3, 16
        data_per_category = {c: torch.tensor(ge["data_per_category"][c]) for c in range(3)}

    class Corr:
        mask_table_u8 = torch.tensor(ge["binary_masks"]).to(torch.uint8).contiguous().to(dev())
        mask_to_use_i32 = torch.tensor(ge["mask_to_use"]).to(torch.int32).contiguous().to(dev())
    val = [int(v) for v in ge["validation_indices"]]
    rl = RankingLoss(DS(), val, device=dev())
    rng = np.random.default_rng(4)
    ref_total = 0.0
    for call in (6, 7):
        idx, run = ge.calls()[call]
        _, fm = O.get_masks(ge["binary_masks"], ge["nb_missing_per_run"], ge["mask_to_use"], 1, idx, run)
        pred = rng.standard_normal((len(idx), 48)).astype(np.float32)
        rl.add(torch.tensor(pred, device=dev()), torch.tensor(np.asarray(idx), dtype=torch.int32, device=dev()), Corr, run=run, chunk=32)
        ref_total += O.ranking_loss(pred, fm, idx, list(ge["data_per_category"]), 16, val)
    got = rl.total()
    assert abs(got - ref_total) <= 1e-3 * abs(ref_total) + 4.0 / (len(val) - 1), (got, ref_total)
    assert rl.total() == 0.0                                     # reset

    # larger: S = 3, E = 64, 3000 observations, 700 validation rows, batch 500, chunks of 256 columns
    S, E, N, V, B = 3, 64, 3000, 700, 500
    g = torch.Generator().manual_seed(11)

    class DS2:
        nb_predictor, nb_used_category, embedding_size = S * E, S, E
        data_per_category = {c: torch.randn(N, E, generator=g) for c in range(S)}
    table = torch.ones(S, S * E, dtype=torch.uint8)
    for c in range(S):
        table[c, c * E:(c + 1) * E] = 0

    class Corr2:
        mask_table_u8 = table.to(dev())
        mask_to_use_i32 = torch.randint(0, S, (N, 2), generator=g, dtype=torch.int32).to(dev())
    val2 = torch.randperm(N, generator=g)[:V].tolist()
    rl2 = RankingLoss(DS2(), val2, device=dev())
    idx = torch.tensor(val2[:B], dtype=torch.int32)
    pred = torch.randn(B, S * E, generator=g)
    fmask = table[Corr2.mask_to_use_i32.cpu()[idx.long(), 1].long()].float()
    ref = rl2.get(pred.to(dev()), fmask.to(dev()), idx.tolist())
    rl2.add(pred.to(dev()), idx.to(dev()), Corr2, run=1, chunk=256)
    got = rl2.total()
    assert abs(got - ref) <= 2.0 * B / (V - 1) * 1e-2 + 1e-6 * abs(ref), (got, ref)     # at most a handful of flipped near-ties
    assert 0.2 * B < ref < 0.8 * B                                                       # (random predictions rank mid-field)


def test_ranking_loss_batched_at_headline_width_runs_on_the_bf16_plane_gemm(hip, f32_gemm_mode):
    """The validation rank metric at the headline embedding width (3 x 512, 3072 validation rows of batch 3072, chunks of 1536
    columns): its similarity GEMMs (24 x 12 tiles, K = 512, the row count of each slot's group read on the DEVICE: GemmF32::m_dev) are
    large enough for gemm_f32x3.hip - the one caller that hands it a device-side row count.  Against the wave-per-item kernel
    (RankingLoss.get, one similarity at a time in fp32); near-ties may flip a handful of comparisons."""
    from codae.tool import RankingLoss
    S, E, N, V, B = 3, 512, 4000, 3072, 3072
    g = torch.Generator().manual_seed(5)

    class DS:
        nb_predictor, nb_used_category, embedding_size = S * E, S, E
        data_per_category = {c: torch.randn(N, E, generator=g) for c in range(S)}
    table = torch.ones(S, S * E, dtype=torch.uint8)
    for c in range(S):
        table[c, c * E:(c + 1) * E] = 0

    class Corr:
        mask_table_u8 = table.to(dev())
        mask_to_use_i32 = torch.randint(0, S, (N, 1), generator=g, dtype=torch.int32).to(dev())
    val = torch.randperm(N, generator=g)[:V].tolist()
    rl = RankingLoss(DS(), val, device=dev())
    idx = torch.tensor(val[:B], dtype=torch.int32)
    pred = torch.randn(B, S * E, generator=g)
    fmask = table[Corr.mask_to_use_i32.cpu()[idx.long(), 0].long()].float()
    sub = list(range(0, B, 8))                                   # the item kernel on every 8th sample is reference enough
    ref = rl.get(pred[sub].to(dev()), fmask[sub].to(dev()), idx[sub].tolist())
    rl.add(pred[sub].to(dev()), idx[sub].to(dev()), Corr, run=0, chunk=1536)
    small = rl.total()
    assert abs(small - ref) <= 2.0 * len(sub) / (V - 1) * 1e-2 + 1e-6 * abs(ref), (small, ref)
    rl.add(pred.to(dev()), idx.to(dev()), Corr, run=0, chunk=1536)          # the whole batch: 24 x 12 tiles per slot -> the plane kernel
    got = rl.total()
    f32_gemm_mode("native")                                                  # the same launches on the fp32-MFMA kernel
    rl.add(pred.to(dev()), idx.to(dev()), Corr, run=0, chunk=1536)
    native = rl.total()
    assert abs(got - native) <= 20.0 / (V - 1), (got, native)               # a flipped near-tie moves the total by 1 / (V - 1)
    # every 8th sample is an unbiased sample of the batch: the two per-sample means agree to a few percent
    assert abs(got / B - ref / len(sub)) <= 0.05 * ref / len(sub), (got / B, ref / len(sub))
    assert 0.2 * B < got < 0.8 * B


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols", [(64, 64), (8, 8), (72, 200), (1536, 1536), (512, 16), (24, 1000)])
def test_transpose_bf16(rows, cols):
    """The transposed weight shadow the data-gradient GEMM reads: bit-exact W.t() (pure data movement)."""
    import torch
    from codae.hip import lib, check, ptr, current_stream
    src = torch.randn(rows, cols, device="cuda").to(torch.bfloat16)
    dst = torch.zeros(cols, rows, device="cuda", dtype=torch.bfloat16)
    check(lib().codae_transpose_bf16(ptr(src), ptr(dst), rows, cols, current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(dst, src.t().contiguous())
