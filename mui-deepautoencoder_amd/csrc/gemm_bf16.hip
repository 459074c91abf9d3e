// bf16 x bf16 -> fp32 GEMM on v_mfma_f32_16x16x32_bf16 for the three contractions of the DAE
// step (throughput mode).  One kernel template, two operand storage modes:
//
//   OP_KC  operand stored [rows][k]  (k contiguous)  : x, dy as A; W[out][in] as B of the forward
//   OP_KS  operand stored [k][rows]  (rows contiguous): W as B of dgrad (k = out index),
//                                                       dy and x as A and B of wgrad (k = batch index)
//
//   forward : y[m][n]  = sum_k x[m][k]  W[n][k]      A KC, B KC      (+bias, ReLU)
//   dgrad   : dx[m][k] = sum_n dy[m][n] W[n][k]      A KC, B KS      (* [h>0], + column sums = next bias grad)
//   wgrad   : dW[n][k] = sum_m dy[m][n] x[m][k]      A KS, B KS      (fp32 out, split-K slabs)
//
// so no transposed copy of the weights or activations is ever written to HBM.
//
// Workgroup: 256 threads = 4 waves (2 x 2), tile 128 x 128 x 64, each wave 64 x 64 = 4 x 4 MFMA
// tiles.  Both operand tiles are staged HBM -> LDS with global_load_lds_dwordx4 (LDS-DMA, 1 KiB
// per wave-instruction, no VGPR round trip), double-buffered (64 KiB LDS, 2 workgroups per CU);
// the LDS image is lane-linear, so the bank swizzle is applied to the per-lane SOURCE address and
// again on the read:
//   KC image: [128 rows][8 x 16 B chunks], chunk' = chunk ^ ((row >> 1) & 7)   -> ds_read_b128 conflict-free
//   KS image: [64 k-rows][8 x 32 B blocks], block' = block ^ key(k-row),
//             key = (kr & 3) | (((kr >> 3) & 1) << 2)                         -> ds_read_b64_tr_b16 conflict-free
// A k-strided operand is read with ds_read_b64_tr_b16 (hardware 4 x 16 transpose), two reads per
// 8-element MFMA fragment, natural k order, so KC and KS operands mix freely in one MFMA.
//
// The MFMA is issued with the operands swapped (D^T = B^T-frag x A-frag), which leaves each lane
// with 4 CONSECUTIVE output columns of one row: the epilogue stores 8 B (bf16) or 16 B (fp32)
// per lane, and bias / ReLU-mask loads are vector loads too.
#include "codae_common.h"

namespace codae {
namespace {

constexpr int BM = 128, BN = 128, BK = 64, NT = 256;
constexpr int TILE_BYTES = 128 * 64 * 2;      // one operand tile (either mode) = 16 KiB
constexpr int BUF_BYTES = 2 * TILE_BYTES;     // A + B
constexpr int SMEM_BYTES = 2 * BUF_BYTES;     // double buffer

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) const void gvoid;

__device__ __forceinline__ void glds16(const void* gsrc, lds_char* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((gvoid*)gsrc, (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
}

// Stage one operand tile.  P: operand base, ld: leading dimension (elements),
// r0: first row (KC) / first column (KS) of the tile, rmax: number of valid rows/cols,
// k0: first k of the tile.  Out-of-range rows / columns are clamped to valid memory
// (their products land in output elements that are never stored).
template <int MODE>
__device__ __forceinline__ void stage_tile(lds_char* tile, const bf16_t* __restrict__ P, int64_t ld, int r0,
                                           int rmax, int k0, int w, int lane) {
    if constexpr (MODE == OP_KC) {
        const int r8 = lane >> 3, cp = lane & 7;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rb = it * 4 + w;               // block of 8 rows
            const int row = rb * 8 + r8;
            const int c = cp ^ ((row >> 1) & 7);
            int grow = r0 + row;
            grow = grow < rmax ? grow : rmax - 1;
            glds16(P + (int64_t)grow * ld + k0 + c * 8, tile + rb * 1024);
        }
    } else {
        const int kr4 = lane >> 4, cp = lane & 15;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int kb = it * 4 + w;               // block of 4 k-rows
            const int kr = kb * 4 + kr4;
            const int key = (kr & 3) | (((kr >> 3) & 1) << 2);
            const int c = cp ^ (key << 1);
            int col = r0 + c * 8;
            col = col + 8 <= rmax ? col : rmax - 8;
            glds16(P + (int64_t)(k0 + kr) * ld + col, tile + kb * 1024);
        }
    }
}

// One 8-element MFMA fragment of 16-row/col tile `t`, k-step `s` (32 deep) of an operand tile.
template <int MODE>
__device__ __forceinline__ bf16x8 read_frag(const lds_char* tile, int t, int s, int lane) {
    if constexpr (MODE == OP_KC) {
        const int r = lane & 15, g = lane >> 4;
        const int off = (16 * t + r) * 128 + (((4 * s + g) ^ (r >> 1)) << 4);
        const s16x8 v = *reinterpret_cast<const __attribute__((address_space(3))) s16x8*>(tile + off);
        return __builtin_bit_cast(bf16x8, v);
    } else {
        const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const int key = q | ((g & 1) << 2);
        const int kr = 32 * s + 8 * g + q;
        const int off = kr * 256 + ((t ^ key) << 5) + 8 * p;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(tile + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(tile + off + 4 * 256));
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <int A_MODE, int B_MODE, bool C_F32>
__global__ __launch_bounds__(NT, 2) void gemm_bf16_kernel(GemmBf16 g, int tiles_n, int tiles_mn, int kt_total) {
    __shared__ __attribute__((aligned(16))) char smem_raw[SMEM_BYTES];
    lds_char* smem = (lds_char*)smem_raw;

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = w >> 1, wc = w & 1;

    // XCD-aware remap: consecutive tile ids (same A row panel) run on one XCD's L2
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int z = bid / tiles_mn;
    const int tmn = bid - z * tiles_mn;
    const int tm = tmn / tiles_n, tn = tmn - tm * tiles_n;
    const int i0 = tm * BM, j0 = tn * BN;
    const int kt_begin = (int)((int64_t)kt_total * z / g.split_k);
    const int kt_end = (int)((int64_t)kt_total * (z + 1) / g.split_k);
    const int nkt = kt_end - kt_begin;

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nkt > 0) {
        stage_tile<A_MODE>(smem, g.A, g.lda, i0, g.M, kt_begin * BK, w, lane);
        stage_tile<B_MODE>(smem + TILE_BYTES, g.B, g.ldb, j0, g.N, kt_begin * BK, w, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (int kt = 0; kt < nkt; ++kt) {
        lds_char* cur = smem + (kt & 1) * BUF_BYTES;
        lds_char* nxt = smem + ((kt + 1) & 1) * BUF_BYTES;
        if (kt + 1 < nkt) {
            const int k0 = (kt_begin + kt + 1) * BK;
            stage_tile<A_MODE>(nxt, g.A, g.lda, i0, g.M, k0, w, lane);
            stage_tile<B_MODE>(nxt + TILE_BYTES, g.B, g.ldb, j0, g.N, k0, w, lane);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) af[t] = read_frag<A_MODE>(cur, 4 * wr + t, s, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t) bfr[t] = read_frag<B_MODE>(cur + TILE_BYTES, 4 * wc + t, s, lane);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    // swapped operands: D rows <-> output column (n), D cols <-> output row (m)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt], af[mt], acc[mt][nt], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane holds C[i][j .. j+3], i = 16*mt + (lane & 15), j = 16*nt + 4*(lane >> 4)
    const int li = lane & 15, g4 = (lane >> 4) * 4;
    char* Cbase = reinterpret_cast<char*>(g.C);
    if (g.split_k > 1) Cbase += (int64_t)z * g.M * g.ldc * sizeof(float);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int j = j0 + 64 * wc + 16 * nt + g4;
        const bool jok = j < g.N;   // N is a multiple of 8 and j of 4: j < N => j + 3 < N
        float4 bj = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias != nullptr && jok) bj = *reinterpret_cast<const float4*>(g.bias + j);
        float cs0 = 0.f, cs1 = 0.f, cs2 = 0.f, cs3 = 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int i = i0 + 64 * wr + 16 * mt + li;
            if (i < g.M && jok) {
                float v0 = acc[mt][nt][0] + bj.x, v1 = acc[mt][nt][1] + bj.y;
                float v2 = acc[mt][nt][2] + bj.z, v3 = acc[mt][nt][3] + bj.w;
                if (g.relu) {
                    v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
                }
                if (g.relu_src != nullptr) {
                    const uint2 h = *reinterpret_cast<const uint2*>(g.relu_src + (int64_t)i * g.ld_relu + j);
                    // bf16 > 0  <=>  sign clear and magnitude non-zero
                    v0 = ((h.x & 0x8000u) == 0 && (h.x & 0x7fffu) != 0) ? v0 : 0.f;
                    v1 = ((h.x & 0x80000000u) == 0 && (h.x & 0x7fff0000u) != 0) ? v1 : 0.f;
                    v2 = ((h.y & 0x8000u) == 0 && (h.y & 0x7fffu) != 0) ? v2 : 0.f;
                    v3 = ((h.y & 0x80000000u) == 0 && (h.y & 0x7fff0000u) != 0) ? v3 : 0.f;
                }
                if constexpr (C_F32) {
                    *reinterpret_cast<float4*>(Cbase + ((int64_t)i * g.ldc + j) * 4) = make_float4(v0, v1, v2, v3);
                } else {
                    uint2 o;
                    o.x = (uint32_t)f32_to_bf16(v0) | ((uint32_t)f32_to_bf16(v1) << 16);
                    o.y = (uint32_t)f32_to_bf16(v2) | ((uint32_t)f32_to_bf16(v3) << 16);
                    *reinterpret_cast<uint2*>(Cbase + ((int64_t)i * g.ldc + j) * 2) = o;
                }
                cs0 += v0; cs1 += v1; cs2 += v2; cs3 += v3;
            }
        }
        if (g.colsum != nullptr) {
            // reduce over the 16 rows held by lanes with equal (lane >> 4)
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                cs0 += __shfl_xor(cs0, o); cs1 += __shfl_xor(cs1, o);
                cs2 += __shfl_xor(cs2, o); cs3 += __shfl_xor(cs3, o);
            }
            if (li == 0 && jok) {
                atomicAdd(&g.colsum[j + 0], cs0); atomicAdd(&g.colsum[j + 1], cs1);
                atomicAdd(&g.colsum[j + 2], cs2); atomicAdd(&g.colsum[j + 3], cs3);
            }
        }
    }
}

}  // namespace

bool gemm_bf16_supported(int M, int N, int K) {
    // K: whole BK tiles; N: 16-byte rows for vector epilogue / staged chunks
    return M > 0 && N >= 8 && K >= BK && (K % BK) == 0 && (N % 8) == 0;
}

int gemm_bf16(const GemmBf16& g, hipStream_t s) {
    CODAE_REQUIRE(gemm_bf16_supported(g.M, g.N, g.K), "gemm_bf16: unsupported shape M=%d N=%d K=%d (need K %% 64 == 0, N %% 8 == 0)",
                  g.M, g.N, g.K);
    CODAE_REQUIRE(g.a_mode == OP_KC || g.M % 8 == 0, "gemm_bf16: k-strided A needs M %% 8 == 0");
    CODAE_REQUIRE(g.a_mode == OP_KC || g.M >= 8, "gemm_bf16: k-strided A needs M >= 8");
    CODAE_REQUIRE((g.lda % 8) == 0 && (g.ldb % 8) == 0 && (g.ldc % 4) == 0, "gemm_bf16: leading dimensions must keep 16-byte rows");
    CODAE_REQUIRE(g.split_k >= 1 && (g.split_k == 1 || g.c_f32), "gemm_bf16: split-K needs fp32 output slabs");
    CODAE_REQUIRE((reinterpret_cast<uintptr_t>(g.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(g.B) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(g.C) & 15) == 0,
                  "gemm_bf16: operands must be 16-byte aligned");
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const int kt_total = g.K / BK;
    CODAE_REQUIRE(g.split_k <= kt_total, "gemm_bf16: split_k %d > k tiles %d", g.split_k, kt_total);
    const int64_t nwg = (int64_t)tiles_m * tiles_n * g.split_k;
    CODAE_REQUIRE(nwg < (1 << 30), "gemm_bf16: grid too large");
    dim3 grid((unsigned)nwg), block(NT);
#define LAUNCH(AM, BMODE, CF) \
    hipLaunchKernelGGL((gemm_bf16_kernel<AM, BMODE, CF>), grid, block, 0, s, g, tiles_n, tiles_m * tiles_n, kt_total)
    if (g.a_mode == OP_KC && g.b_mode == OP_KC) { if (g.c_f32) LAUNCH(OP_KC, OP_KC, true); else LAUNCH(OP_KC, OP_KC, false); }
    else if (g.a_mode == OP_KC && g.b_mode == OP_KS) { if (g.c_f32) LAUNCH(OP_KC, OP_KS, true); else LAUNCH(OP_KC, OP_KS, false); }
    else if (g.a_mode == OP_KS && g.b_mode == OP_KS) { if (g.c_f32) LAUNCH(OP_KS, OP_KS, true); else LAUNCH(OP_KS, OP_KS, false); }
    else { set_error("gemm_bf16: operand mode combination not built"); return CODAE_E_UNSUPPORTED; }
#undef LAUNCH
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

}  // namespace codae
