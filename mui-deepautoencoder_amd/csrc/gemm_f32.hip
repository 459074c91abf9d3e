// Exact-fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32): the parity-mode
// engine of the Linear chain, and the path for shapes the bf16 kernels cannot take
// (abalone: 11-wide layers).  The MFMA result is bit-for-bit a k-ordered fp32 fma chain,
// so this is the arithmetic of torch.nn.Linear's fp32 sgemm up to summation order.
//
//   C[i][j] = epi( sum_k A(i,k) * B(j,k) )
//     forward  (embedding_denoising_autoencoder.py:166,183): A = x [M][K], B = W [N][K]
//     dgrad    (autograd of the above):                      A = dy [M][N], B(j=k', kk=n) = W[n][k']
//     wgrad    :                                             A(i=n, kk=m) = dy[m][n], B(j=k', kk=m) = x[m][k']
//
// Tile: 128 x 128 x 32 per 256-thread workgroup; each of the 4 waves owns a 64 x 64
// sub-tile (2 x 2 MFMA 32x32 accumulators).  Operands go through LDS in a k-major image
// S[k][row] (row stride 129 / 132 floats, see LDS_LD) so that both MFMA operand reads are conflict-free
// ds_read_b32 across 32 consecutive banks; global loads follow whichever index is
// contiguous in memory and are prefetched into registers one tile ahead.
#include "codae_common.h"

namespace codae {
namespace {

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
// LDS row stride of a k-major operand image S[k][row].  Stored from a k-contiguous operand (KC: each lane scatters
// the 4 k-values of its row): 129 = 1 mod 32 puts the 4 rows x 8 k-groups of a 32-lane store on 32 different banks
// (132 gave 2-way conflicts: PMC SQ_LDS_BANK_CONFLICT 25 % of LDS cycles in the forward form).  Stored from a
// row-contiguous operand (float4 along the row): 132 keeps those stores 16-byte aligned.
template <bool KC> constexpr int LDS_LD = KC ? 129 : 132;

// Load a 128(rows) x 32(k) tile into 16 registers per thread.
//   KC : element(r,k) = P[r*rs + k]      thread -> row (t>>3)+32p, k 4*(t&7)..+3          (p = 0..3)
//   !KC: element(r,k) = P[k*ks + r]      thread -> k (t>>5)+8p,   rows 4*(t&31)..+3
template <bool KC>
__device__ __forceinline__ void load_tile(float (&reg)[16], const float* __restrict__ P, int64_t rs, int64_t ks,
                                          int r0, int k0, int R, int K, bool vec_ok, int t) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if constexpr (KC) {
            const int r = r0 + (t >> 3) + 32 * p;
            const int k = k0 + 4 * (t & 7);
            const float* src = P + (int64_t)r * rs + k;
            if (r < R && vec_ok && k + 3 < K) {
                const float4 v = *reinterpret_cast<const float4*>(src);
                reg[4 * p + 0] = v.x; reg[4 * p + 1] = v.y; reg[4 * p + 2] = v.z; reg[4 * p + 3] = v.w;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) reg[4 * p + c] = (r < R && k + c < K) ? src[c] : 0.f;
            }
        } else {
            const int k = k0 + (t >> 5) + 8 * p;
            const int r = r0 + 4 * (t & 31);
            const float* src = P + (int64_t)k * ks + r;
            if (k < K && vec_ok && r + 3 < R) {
                const float4 v = *reinterpret_cast<const float4*>(src);
                reg[4 * p + 0] = v.x; reg[4 * p + 1] = v.y; reg[4 * p + 2] = v.z; reg[4 * p + 3] = v.w;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) reg[4 * p + c] = (k < K && r + c < R) ? src[c] : 0.f;
            }
        }
    }
}

template <bool KC>
__device__ __forceinline__ void store_tile(float* S, const float (&reg)[16], int t) {
    constexpr int LD = LDS_LD<KC>;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if constexpr (KC) {
            const int r = (t >> 3) + 32 * p;
            const int k = 4 * (t & 7);
#pragma unroll
            for (int c = 0; c < 4; ++c) S[(k + c) * LD + r] = reg[4 * p + c];
        } else {
            const int k = (t >> 5) + 8 * p;
            const int r = 4 * (t & 31);
            *reinterpret_cast<float4*>(&S[k * LD + r]) =
                make_float4(reg[4 * p + 0], reg[4 * p + 1], reg[4 * p + 2], reg[4 * p + 3]);
        }
    }
}

template <bool A_KC, bool B_KC>
// (Three workgroups per CU - __launch_bounds__(NT, 3): 152 registers, no spills, 768 workgroups in one round - measured
//  no faster than two: 349 vs 346 us for the 8192 x 1536 x 1536 forward form.)
__global__ __launch_bounds__(NT) void gemm_f32_kernel(GemmF32 g, bool a_vec, bool b_vec) {
    constexpr int LDA = LDS_LD<A_KC>, LDB = LDS_LD<B_KC>;
    // two operand buffers: tile k + 1 is stored into the other one right after tile k is multiplied, so ONE barrier per
    // K-tile separates "everybody has finished reading buffer b" from "somebody overwrites buffer b" (it was store /
    // barrier / multiply / barrier on a single buffer: parity-mode step 10.4 ms)
    __shared__ __attribute__((aligned(16))) float smem[2][BK * 132 * 2];

    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int li = lane & 31, kh = lane >> 5;
    // XCD-aware tile map (as the bf16 kernels): workgroup b of the x-y grid runs on XCD b % 8; each XCD takes a contiguous
    // eighth of the tile ids, and ids walk the output in panels of 4 row tiles (down the panel, then the next column), so the
    // workgroups an XCD runs at a time share a few A row panels and B column panels in its L2 instead of one tile of every row.
    int tile_m, tile_n;
    {
        const int tiles_n = gridDim.x, tiles_m = gridDim.y, nwg = tiles_n * tiles_m;
        int id = blockIdx.y * tiles_n + blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        constexpr int GROUP_M = 4;
        const int grp = id / (GROUP_M * tiles_n);
        const int tm0 = grp * GROUP_M;
        const int gsz = tiles_m - tm0 < GROUP_M ? tiles_m - tm0 : GROUP_M;
        const int within = id - grp * (GROUP_M * tiles_n);
        tile_n = within / gsz;
        tile_m = tm0 + (within - tile_n * gsz);
    }
    const int i0 = tile_m * BM, j0 = tile_n * BN;
    if (g.m_dev != nullptr) {                 // (g is this kernel's own copy of the descriptor)
        const int m = *g.m_dev;
        if (m < g.M) g.M = m;
    }
    if (i0 >= g.M) return;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // split-K: this workgroup's range of 32-deep K-tiles and its slab
    int k_begin = 0, k_end = g.K;
    if (g.split_k > 1) {
        const int kt = (g.K + BK - 1) / BK;
        k_begin = (int)((int64_t)kt * blockIdx.z / g.split_k) * BK;
        k_end = (int)((int64_t)kt * (blockIdx.z + 1) / g.split_k) * BK;
        if (k_end > g.K) k_end = g.K;
        g.C += (int64_t)blockIdx.z * g.M * g.ldc;
    }

    float ra[16], rb[16];
    load_tile<A_KC>(ra, g.A, g.a_rs, g.a_ks, i0, k_begin, g.M, k_end, a_vec, t);
    load_tile<B_KC>(rb, g.B, g.b_rs, g.b_ks, j0, k_begin, g.N, k_end, b_vec, t);
    store_tile<A_KC>(smem[0], ra, t);
    store_tile<B_KC>(smem[0] + BK * 132, rb, t);
    __syncthreads();

    int cur = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        const float* As = smem[cur];
        const float* Bs = smem[cur] + BK * 132;
        const bool more = k0 + BK < k_end;
        if (more) {
            load_tile<A_KC>(ra, g.A, g.a_rs, g.a_ks, i0, k0 + BK, g.M, k_end, a_vec, t);
            load_tile<B_KC>(rb, g.B, g.b_rs, g.b_ks, j0, k0 + BK, g.N, k_end, b_vec, t);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int kr = 2 * kk + kh;
            const float a0 = As[kr * LDA + 64 * wr + li];
            const float a1 = As[kr * LDA + 64 * wr + 32 + li];
            const float b0 = Bs[kr * LDB + 64 * wc + li];
            const float b1 = Bs[kr * LDB + 64 * wc + 32 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) {
            store_tile<A_KC>(smem[cur ^ 1], ra, t);
            store_tile<B_KC>(smem[cur ^ 1] + BK * 132, rb, t);
        }
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int j = j0 + 64 * wc + 32 * ni + li;
        const float bj = (g.bias != nullptr && j < g.N) ? g.bias[j] : 0.f;
        float csum = 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + 64 * wr + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (i < g.M && j < g.N) {
                    float v = acc[mi][ni][r] + bj;
                    if (g.relu) v = fmaxf(v, 0.f);
                    if (g.relu_src != nullptr) v = (g.relu_src[(int64_t)i * g.ld_relu + j] > 0.f) ? v : 0.f;
                    g.C[(int64_t)i * g.ldc + j] = v;
                    csum += v;
                }
            }
        }
        if (g.colsum_part != nullptr) {
            // one partial row per 64-row wave block: no atomics, the finish kernel adds the rows in order
            csum += __shfl_xor(csum, 32);
            // (a wave whose 64-row block lies wholly past M has no partial row: gemm_f32_colsum_rows = ceil(M / 64))
            if (kh == 0 && j < g.N && i0 + 64 * wr < g.M) g.colsum_part[(int64_t)(2 * tile_m + wr) * g.N + j] = csum;
        }
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int gemm_f32(const GemmF32& g, hipStream_t s) {
    CODAE_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "gemm_f32: empty problem M=%d N=%d K=%d", g.M, g.N, g.K);
    CODAE_REQUIRE(g.a_ks == 1 || g.a_rs == 1, "gemm_f32: operand A needs a unit stride");
    CODAE_REQUIRE(g.b_ks == 1 || g.b_rs == 1, "gemm_f32: operand B needs a unit stride");
    const bool a_kc = (g.a_ks == 1);
    const bool b_kc = (g.b_ks == 1);
    const bool a_vec = aligned16(g.A) && ((a_kc ? g.a_rs : g.a_ks) % 4 == 0);
    const bool b_vec = aligned16(g.B) && ((b_kc ? g.b_rs : g.b_ks) % 4 == 0);
    const int split = g.split_k > 1 ? g.split_k : 1;
    // the bf16-plane kernel (gemm_f32x3.hip) wherever its loads are legal (K in whole 32-deep tiles, 16-byte aligned rows): it beats
    // this one at every size tried - 3 x 512 parity-mode step at batch 128: 0.85 against 1.12 ms, batch 1024: 1.63 against 3.59,
    // batch 8192: 6.0 against 10.4 (tools/abl/f32_gemm_threshold.sh); CODAE_F32_GEMM=native keeps everything here
    if (env().f32_gemm != 1 && gemm_f32x3_takes(g)) return gemm_f32x3(g, s);
    CODAE_REQUIRE(split == 1 || (g.bias == nullptr && !g.relu && g.relu_src == nullptr && g.colsum_part == nullptr && g.m_dev == nullptr &&
                                 split <= (g.K + BK - 1) / BK),
                  "gemm_f32: split-K writes plain partial products (no epilogue terms), at most one range per K-tile");
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, split);
    CODAE_REQUIRE(grid.y <= 65535, "gemm_f32: M=%d too large", g.M);
    if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(NT), 0, s, g, a_vec, b_vec);
    else if (a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(NT), 0, s, g, a_vec, b_vec);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(NT), 0, s, g, a_vec, b_vec);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(NT), 0, s, g, a_vec, b_vec);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

}  // namespace codae
