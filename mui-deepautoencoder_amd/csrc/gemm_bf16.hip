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
// Workgroup tile BM x BN x 64 with WM x WN waves.  Large problems: 128 x 192 with 4 waves and exactly
// 80 KiB of LDS, so TWO workgroups with independent barriers share a CU (8192 x 1536 outputs = 512 tiles
// = two per CU) - the default for the k-strided forms (dgrad, wgrad) and the fused-loss layer; the plain
// forward form goes to the phase-pipelined kernel (gemm_bf16_pipe.hip).  Also built: 256 x 192 with 8
// waves (one workgroup per CU) and, for small problems, 128 x 128 with 4 waves.
// Both operand tiles are staged HBM -> LDS with global_load_lds_dwordx4 (LDS-DMA, 1 KiB per
// wave-instruction, no VGPR round trip), double-buffered; the LDS image is lane-linear, so the
// bank swizzle is applied to the per-lane SOURCE address and again on the read:
//   KC image: [rows][8 x 16 B chunks], chunk' = chunk ^ ((row >> 1) & 7)       -> ds_read_b128 conflict-free
//   KS image: [64 k-rows][R/16 x 32 B blocks], block' = swizzle(block, k-row)  -> ds_read_b64_tr_b16 conflict-free
//             (XOR for 128/256-wide tiles, rotation mod 12 for the 192-wide one; see ks_to_lds_block)
// A k-strided operand is read with ds_read_b64_tr_b16 (hardware 4 x 16 transpose), two reads per
// 8-element MFMA fragment, natural k order, so KC and KS operands mix freely in one MFMA.
//
// The MFMA is issued with the operands swapped (D^T = B^T-frag x A-frag), which leaves each lane
// with 4 CONSECUTIVE output columns of one row: the epilogue stores 8 B (bf16) or 16 B (fp32)
// per lane, and bias / ReLU-mask loads are vector loads too.
#include "codae_common.h"

namespace codae {
namespace {

constexpr int BK = 64;

#if defined(CODAE_DBG_5D)
// section 5d probe builds only (tools/abl/probe_5d.py): bit 0 = no scalar-chain guard in the fused-loss epilogue, bit 1 = every
// thread's {sq, sqp} of the 64 x 64 fused-loss tile dumped here before the cross-lane reduction
constexpr int DBG5D_WGS = 1024;
__device__ float g_dbg5d[DBG5D_WGS * 256 * 2];
#endif

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) const void gvoid;

__device__ __forceinline__ void glds16(const void* gsrc, lds_char* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((gvoid*)gsrc, (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
}

// ---- k-strided (KS) image: [64 k-rows][R columns] bf16, row = 2R bytes = R/16 blocks of 32 B ----
// A 32-lane half of one ds_read_b64_tr_b16 touches 8 k-rows (kr = 8g + q [+4], g&1 in {0,1},
// q in 0..3) x one 32-B block; the 8 reads must fall on 8 different 32-B slots of the 256-B bank row.
//   R = 128 / 256 (row = 1 or 2 bank rows): block' = block ^ key,  key = q | ((g & 1) << 2)
//   R = 64 (row = half a bank row, 4 blocks: the row's parity picks the half): key = ((kr >> 1) & 1) | (((kr >> 3) & 1) << 1)
//   R = 192 (row = 1.5 bank rows, 12 blocks): slot = (4 kr + block') mod 8, so rotate:
//            block' = (block + rot) mod 12, rot = ((kr >> 1) & 1) + 2 ((kr >> 3) & 1)
template <int R>
__device__ __forceinline__ int ks_to_lds_block(int block, int kr) {
    if constexpr (R == 192) {
        const int b = block + ((kr >> 1) & 1) + 2 * ((kr >> 3) & 1);
        return b >= 12 ? b - 12 : b;
    } else if constexpr (R == 64) {
        return block ^ (((kr >> 1) & 1) | (((kr >> 3) & 1) << 1));
    } else {
        static_assert(R == 128 || R == 256, "KS image: 64, 128, 192 or 256 columns");
        return block ^ ((kr & 3) | (((kr >> 3) & 1) << 2));
    }
}
template <int R>
__device__ __forceinline__ int ks_from_lds_block(int lds_block, int kr) {
    if constexpr (R == 192) {
        const int b = lds_block - ((kr >> 1) & 1) - 2 * ((kr >> 3) & 1);
        return b < 0 ? b + 12 : b;
    } else if constexpr (R == 64) {
        return lds_block ^ (((kr >> 1) & 1) | (((kr >> 3) & 1) << 1));
    } else {
        return lds_block ^ ((kr & 3) | (((kr >> 3) & 1) << 2));
    }
}

// Stage one operand tile of R rows (KC) / R columns (KS) x 64 k with NW waves.
// P: operand base, ld: leading dimension (elements), r0: first row / column of the tile,
// rmax: number of valid rows / columns, k0: first k.  Out-of-range rows / columns are clamped to
// valid memory (their products land in output elements that are never stored).
// global source of the 16 bytes that lane `lane` of wave-instruction `i` places at tile + i*1024 + lane*16
template <int MODE, int R>
__device__ __forceinline__ const bf16_t* stage_src(const bf16_t* __restrict__ P, int64_t ld, int r0, int rmax, int k0,
                                                   int i, int lane) {
    if constexpr (MODE == OP_KC) {
        // image [R rows][8 chunks of 16 B], chunk' = chunk ^ ((row >> 1) & 7)
        const int row = i * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int grow = r0 + row;
        grow = grow < rmax ? grow : rmax - 1;
        return P + (int64_t)grow * ld + k0 + c * 8;
    } else {
        constexpr int CPR = R / 8;            // 16-B chunks per k-row
        const int q = i * 64 + lane;
        const int kr = q / CPR;
        const int cp = q - kr * CPR;
        const int c = ks_from_lds_block<R>(cp >> 1, kr) * 2 + (cp & 1);
        int col = r0 + c * 8;
        col = col + 8 <= rmax ? col : rmax - 8;
        return P + (int64_t)(k0 + kr) * ld + col;
    }
}

template <int MODE, int R, int NW>
__device__ __forceinline__ void stage_tile(lds_char* tile, const bf16_t* __restrict__ P, int64_t ld, int r0,
                                           int rmax, int k0, int w, int lane) {
    constexpr int NINSTR = R * 128 / 1024;        // 1 KiB per wave-instruction
    static_assert(NINSTR % NW == 0, "tile must split evenly over the waves");
#pragma unroll
    for (int it = 0; it < NINSTR / NW; ++it) {
        const int i = it * NW + w;
        glds16(stage_src<MODE, R>(P, ld, r0, rmax, k0, i, lane), tile + i * 1024);
    }
}

// ds_read_b64_tr_b16 issued as inline asm: a transposed LDS read the compiler knows about gets an
// `s_waitcnt vmcnt(0)` in front of it while LDS-DMA loads are in flight (it cannot tell the images apart),
// i.e. the wave would wait for the NEXT K-tile's loads before reading the current one.  The K loop
// waits for these reads itself (wait_lgkm + settle below).
template <int OFF>
__device__ __forceinline__ s16x4 lds_read_tr16(const lds_char* p) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"((uint32_t)(uintptr_t)p), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// hands fragment registers back to the compiler after the wait that covers their reads
template <int N>
__device__ __forceinline__ void settle(bf16x8 (&f)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i]));
}

// One 8-element MFMA fragment of 16-row/col tile `t`, k-step `s` (32 deep) of an operand tile.
template <int MODE, int R>
__device__ __forceinline__ bf16x8 read_frag(const lds_char* tile, int t, int s, int lane) {
    if constexpr (MODE == OP_KC) {
        const int r = lane & 15, g = lane >> 4;
        const int off = (16 * t + r) * 128 + (((4 * s + g) ^ (r >> 1)) << 4);
        const s16x8 v = *reinterpret_cast<const __attribute__((address_space(3))) s16x8*>(tile + off);
        return __builtin_bit_cast(bf16x8, v);
    } else {
        const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const int kr = 32 * s + 8 * g + q;
        const int off = kr * (2 * R) + (ks_to_lds_block<R>(t, kr) << 5) + 8 * p;
        const s16x4 lo = lds_read_tr16<0>(tile + off);
        const s16x4 hi = lds_read_tr16<4 * 2 * R>(tile + off);
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    }
}

// Tile BM x BN x 64, WM x WN waves, each wave (BM/WM) x (BN/WN) = TM x TN MFMA tiles of 16 x 16.
// One output tile of one GEMM: the whole kernel body, so that the plain kernel (one GEMM per launch) and the grouped
// kernel (several GEMMs per launch) share it.  wg / nwg: this workgroup's index among the nwg workgroups of ITS GEMM.
// NS = operand stages in LDS.  2: the one-barrier double buffer (tile t + 1 requested while t is multiplied).  4: tiles up to
// t + 3 in flight (stock batch 128 at io 1536, whole step: 0.58 ms with 2 stages, 0.43 with 4, 0.44 with 8), for launches
// too small to fill the chip - a 128-row batch of a wide layer is 12 workgroups, each
// walking its 24 K-tiles of COLD weights one memory latency at a time (27 us for 0.6 GFLOP with NS = 2).
template <int BM, int BN, int WM, int WN, int A_MODE, int B_MODE, bool C_F32, bool LOSS, int NS = 2>
__device__ __forceinline__ void gemm_bf16_tile(const GemmBf16& g, int tiles_n, int tiles_mn, int kt_total, int wg, int nwg,
                                               char* smem_raw) {
    static_assert(NS == 2 || (NS == 4 && !LOSS), "2 or 4 stages (the fused-loss epilogue keeps its row table behind 2)");
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF_BYTES = A_BYTES + B_BYTES;
    lds_char* smem = (lds_char*)smem_raw;

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = w / WN, wc = w % WN;

    // XCD-aware remap: consecutive tile ids (same A row panel) run on one XCD's L2
    int bid = wg;
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

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if constexpr (LOSS) {
        const LossFuse& L = g.loss;
        int* rowinfo = reinterpret_cast<int*>(smem_raw + 2 * BUF_BYTES);
        for (int r = threadIdx.x; r < BM; r += 64 * NW) {
            const int i = i0 + r;
            int src = -1, id = 0;
            if (i < L.B && i < g.M) {
                src = L.row_idx ? L.row_idx[i] : i;
                if (L.mask_id != nullptr) id = L.mask_id[i];
                else if (L.mask_to_use != nullptr) id = L.mask_to_use[(int64_t)src * L.nb_run + L.run];
            }
            rowinfo[2 * r] = src; rowinfo[2 * r + 1] = id;     // visible after the K loop's barriers
        }
    }

    // LDS-DMA pieces a wave issues per stage; with NS > 2 requests past the last K-tile re-read it into a slot nobody is
    // reading (slot t mod NS: no live tile shares it), so that the counted waits stay exact to the end
    constexpr int PIECES = (BM + BN) * 128 / 1024 / NW;
    auto stage = [&](int t) {
        const int tc = t < nkt ? t : nkt - 1;
        lds_char* dst = smem + (t % NS) * BUF_BYTES;
        const int k0 = (kt_begin + tc) * BK;
        stage_tile<A_MODE, BM, NW>(dst, g.A, g.lda, i0, g.M, k0, w, lane);
        stage_tile<B_MODE, BN, NW>(dst + A_BYTES, g.B, g.ldb, j0, g.N, k0, w, lane);
    };
    // (NS > 2: a raw barrier - __syncthreads() carries a fence that waits vmcnt(0), i.e. for the whole prefetch queue)
    auto tile_barrier = [&]() {
        if constexpr (NS > 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            __syncthreads();
        }
    };
    if (nkt > 0) {
#pragma unroll
        for (int t = 0; t < NS - 1; ++t) stage(t);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (NS - 2)) : "memory");
        tile_barrier();
    }
    for (int kt = 0; kt < nkt; ++kt) {
        lds_char* cur = smem + (kt % NS) * BUF_BYTES;
        if (NS > 2 || kt + 1 < nkt) stage(kt + NS - 1);
        // fragments of BOTH k-steps are requested up front: the MFMAs of k-step 0 start as soon as
        // their operands are back (counted lgkmcnt) while the reads of k-step 1 are still in flight
        bf16x8 af[2][TM], bfr[2][TN];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int t = 0; t < TM; ++t) af[s][t] = read_frag<A_MODE, BM>(cur, TM * wr + t, s, lane);
#pragma unroll
            for (int t = 0; t < TN; ++t) bfr[s][t] = read_frag<B_MODE, BN>(cur + A_BYTES, TN * wc + t, s, lane);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if constexpr (A_MODE == OP_KS || B_MODE == OP_KS) {
                // asm-issued reads: k-step 0 is back once at most k-step 1's reads are outstanding
                // (LDS returns in order; lgkmcnt saturates at 15)
                constexpr int PER_STEP = TM * (A_MODE == OP_KS ? 2 : 1) + TN * (B_MODE == OP_KS ? 2 : 1);
                if (s == 0) wait_lgkm<(PER_STEP < 15 ? PER_STEP : 15)>(); else wait_lgkm<0>();
                if constexpr (A_MODE == OP_KS) settle(af[s]);
                if constexpr (B_MODE == OP_KS) settle(bfr[s]);
            }
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt)
                    // swapped operands: D rows <-> output column (n), D cols <-> output row (m)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[s][nt], af[s][mt], acc[mt][nt], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (NS - 2)) : "memory");     // tile kt + 1 has landed
        tile_barrier();
    }
    if constexpr (NS > 2) {          // the trailing re-requests land before the epilogue reuses the buffers
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane holds C[i][j .. j+3], i = 16*mt + (lane & 15), j = 16*nt + 4*(lane >> 4)
    const int li = lane & 15, g4 = (lane >> 4) * 4;
    if constexpr (LOSS) {
        // Last forward layer of a training step: y = acc + bias never goes to HBM.  The fp32 tile is
        // staged through LDS in two row halves; each pass reads x (gathered dataset rows, fp32) and
        // the mask bytes coalesced, writes dy = 2 (y - x) / n as bf16 in whole rows, and accumulates
        // sum (x-y)^2, sum (1-m)(x-y)^2 and the column sums of dy (= bias gradient).
        constexpr int NT = 64 * NW;
        constexpr int HR = BM / 2;                              // rows per pass
        constexpr int PITCH = BN * 4 + 16;
        static_assert(HR * PITCH <= 2 * BUF_BYTES, "fp32 half tile must fit in the staging buffers");
        static_assert((BM / WM) <= HR, "a wave's rows must lie in one half");
        constexpr int CH = BN / 8, RL = NT / CH;
        const LossFuse& L = g.loss;
        const bool masked = (L.mask_id != nullptr) || (L.mask_to_use != nullptr);
        const int c = threadIdx.x % CH, rl = threadIdx.x / CH;
        const int j = j0 + c * 8;
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float sq = 0.f, sqp = 0.f;
#if defined(CODAE_DBG_5D) && (CODAE_DBG_5D & 4)
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        f32x2_t sq2 = {0.f, 0.f};
#endif
        bf16_t* Cb = reinterpret_cast<bf16_t*>(g.C);
        constexpr int ITER = (HR + RL - 1) / RL;                // rows per thread per pass
        const int* rowinfo = reinterpret_cast<const int*>(smem_raw + 2 * BUF_BYTES);
        if (nkt == 0) __syncthreads();                          // (rowinfo otherwise published by the K loop's barriers)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            // all of this pass's x rows and mask bytes are requested before the tile half is staged: one exposed gather
            // latency per pass instead of one per row.  The loads are UNCONDITIONAL, from clamped (always valid) addresses,
            // and dead lanes are masked out of the arithmetic instead: a branch around each load makes the compiler wait
            // for every load before issuing the next (round 1 had them conditional: the epilogue ran ~14 dependent HBM
            // round trips, 30 us instead of ~12).
            float4 xa[ITER], xb[ITER];
            uint2 mk[ITER];
            bool live[ITER];
            const int jc = j < g.N ? j : 0;
            const uint8_t* tb = masked ? L.table : reinterpret_cast<const uint8_t*>(L.data);   // (unmasked: any valid bytes)
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int r = rl + it * RL;
                const int rr = (rl < RL && r < HR) ? r : 0;
                const int src_row = rowinfo[2 * (hh * HR + rr)];
                const int id = rowinfo[2 * (hh * HR + rr) + 1];
                live[it] = rl < RL && j < g.N && r < HR && src_row >= 0;
                const float* xp = L.data + (int64_t)(src_row >= 0 ? src_row : 0) * L.io + jc;
#if defined(CODAE_LOSS_ABL) && (CODAE_LOSS_ABL & 1)     // timing-only ablation: no gather of the target rows
                xa[it] = make_float4(0.5f, 0.25f, 0.125f, 0.75f); xb[it] = xa[it]; (void)xp; (void)tb; (void)id;
                mk[it] = make_uint2(0x01010101u, 0x00010101u);
#else
                xa[it] = *reinterpret_cast<const float4*>(xp);
                xb[it] = *reinterpret_cast<const float4*>(xp + 4);
                const uint2 mv = *reinterpret_cast<const uint2*>(tb + (int64_t)id * L.io + jc);
                mk[it] = masked ? mv : make_uint2(0x01010101u, 0x01010101u);
#endif
            }
            if (((BM / WM) * wr) / HR == hh) {
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) {
                    const int jl = (BN / WN) * wc + 16 * nt + g4;
                    const float4 bj = load_bias4(g.bias, g.A, j0 + jl, g.N);
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt) {
                        const int il = (BM / WM) * wr + 16 * mt + li - hh * HR;
                        f32x4 v = acc[mt][nt];
                        v[0] += bj.x; v[1] += bj.y; v[2] += bj.z; v[3] += bj.w;
                        *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(smem + il * PITCH + jl * 4) = v;
                    }
                }
            }
            __syncthreads();
            if (rl < RL && j < g.N) {
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    const int r = rl + it * RL;
                    const int i = i0 + hh * HR + r;
                    if (r >= HR || i >= g.M) break;
                    uint4 o = make_uint4(0u, 0u, 0u, 0u);
                    if (live[it]) {
                        const f32x4 ya = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(smem + r * PITCH + c * 32);
                        const f32x4 yb = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(smem + r * PITCH + c * 32 + 16);
                        const uint2 m = mk[it];
                        const float xv[8] = {xa[it].x, xa[it].y, xa[it].z, xa[it].w, xb[it].x, xb[it].y, xb[it].z, xb[it].w};
                        const float yv[8] = {ya[0], ya[1], ya[2], ya[3], yb[0], yb[1], yb[2], yb[3]};
                        float gq[8];
#if defined(CODAE_DBG_5D) && (CODAE_DBG_5D & 4)
                        // probe variant: the sum of squares as an explicit packed chain WITHOUT half swaps (even / odd lanes
                        // accumulate apart, v_pk_fma_f32 with default op_sel); folded after the loop
                        {
                            typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
                            for (int k = 0; k < 8; k += 2) {
                                f32x2_t d2 = {xv[k] - yv[k], xv[k + 1] - yv[k + 1]};
#if (CODAE_DBG_5D & 8)
                                // in-place form the compiler chose in the failing build: dst = src0 = src1, accumulator as src2
                                asm volatile("v_pk_fma_f32 %0, %0, %0, %1" : "+v"(d2) : "v"(sq2));
                                sq2 = d2;
#elif (CODAE_DBG_5D & 16)
                                // the failing build's FIRST op: squares of d added to the HALF-SWAPPED src2 (op_sel on src2)
                                if (k == 0) {
                                    f32x2_t se2 = d2 * d2, r2;
                                    asm volatile("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r2) : "v"(d2), "v"(se2));
                                    sq2 += r2 * 0.5f;
                                } else {
                                    asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(sq2) : "v"(d2));
                                }
#else
                                asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(sq2) : "v"(d2));
#endif
                            }
                        }
#endif
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const float d = xv[k] - yv[k];
                            const float se = d * d;
#if !(defined(CODAE_DBG_5D) && (CODAE_DBG_5D & 4))
                            sq += se;
#endif
                            // Keeps sq a scalar chain.  Left alone, the SLP vectorizer packs this
                            // accumulation into v_pk_fma_f32 / v_pk_add_f32 pairs and folds the lane shuffle into op_sel; one of
                            // them is `v_pk_fma_f32 D, A, A, C op_sel:[0,0,1] op_sel_hi:[1,1,0]` (lo result = A.lo^2 + C.HI).
                            // On this MI355X a packed fp32 op whose LO result takes the HI half of src1 / src2 reads that half
                            // as ZERO in lanes 48-63 whenever the SIMD's matrix pipe is busy (this kernel's neighbours are in
                            // their K loops): one term of one quarter-wave lost, in 4-28 % of the launches of the 64 x 64 tile.
                            // Stand-alone proof: tools/abl/pk_fma_opsel_repro.hip; in this kernel: tools/abl/probe_5d.py.  The
                            // empty asm makes sq opaque between additions, so no vector chain can be formed from it, and
                            // tools/check_isa.py (rule 4) rejects any such instruction anywhere in the shipped code object.
#if !(defined(CODAE_DBG_5D) && (CODAE_DBG_5D & 1))
                            asm volatile("" : "+v"(sq));
#endif
                            const uint32_t mb = ((k < 4 ? m.x : m.y) >> (8 * (k & 3))) & 0xff;
                            if (mb == 0) sqp += se;
                            gq[k] = -2.f * d * L.inv_n;
                        }
                        o.x = pack_bf16x2(gq[0], gq[1]); o.y = pack_bf16x2(gq[2], gq[3]);
                        o.z = pack_bf16x2(gq[4], gq[5]); o.w = pack_bf16x2(gq[6], gq[7]);
#pragma unroll
                        for (int k = 0; k < 8; ++k) cs[k] += gq[k];
                    }
#if defined(CODAE_LOSS_ABL) && (CODAE_LOSS_ABL & 2)     // timing-only ablation: no dY stores (kept live)
                    asm volatile("" ::"v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
#else
                    *reinterpret_cast<uint4*>(Cb + (int64_t)i * g.ldc + j) = o;
#endif
                }
            }
            __syncthreads();
        }
#if defined(CODAE_DBG_5D) && (CODAE_DBG_5D & 4)
        sq = sq2[0] + sq2[1];
#endif
#if defined(CODAE_DBG_5D) && (CODAE_DBG_5D & 2)
        if (bid < DBG5D_WGS && NT == 256) { g_dbg5d[(bid * 256 + threadIdx.x) * 2] = sq; g_dbg5d[(bid * 256 + threadIdx.x) * 2 + 1] = sqp; }
#endif
        // block sums -> scalars; column sums -> bias gradient
        float* red = reinterpret_cast<float*>(smem_raw);
        static_assert(RL * BN * 4 + 64 <= 2 * BUF_BYTES, "reduction scratch must fit");
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) { sq += __shfl_xor(sq, o2); sqp += __shfl_xor(sqp, o2); }
        float* wsum = red + RL * BN;
        if (lane == 0) { wsum[2 * w] = sq; wsum[2 * w + 1] = sqp; }
        if (rl < RL) {
#pragma unroll
            for (int k = 0; k < 8; ++k) red[rl * BN + c * 8 + k] = cs[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float a = 0.f, b2 = 0.f;
            for (int ww = 0; ww < NW; ++ww) { a += wsum[2 * ww]; b2 += wsum[2 * ww + 1]; }
            L.parts[2 * bid] = (double)a;                       // (bid: tile index after the XCD remap, < gridDim.x)
            L.parts[2 * bid + 1] = masked ? (double)b2 : 0.0;
        }
        if (g.colsum_part != nullptr) {
            for (int col = threadIdx.x; col < BN; col += NT) {
                float sum = 0.f;
                for (int r = 0; r < RL; ++r) sum += red[r * BN + col];
                if (j0 + col < g.N) g.colsum_part[(int64_t)tm * g.N + j0 + col] = sum;
            }
        }
        return;
    }
    if constexpr (!C_F32) {
        // bf16 output: 8 B per lane straight from the accumulators would write 32-B row segments
        // (measured ~12 us per 25 MB output).  Stage the tile through the now idle LDS and write
        // whole rows, 16 B per lane; the ReLU-mask loads and the bias-gradient column sums ride on
        // that pass, coalesced as well.
        constexpr int NT = 64 * NW;
        constexpr int PITCH = BN * 2 + 16;                 // bytes; rows stay 16-B aligned
        static_assert(BM * PITCH <= 2 * BUF_BYTES, "output tile must fit in the staging buffers");
        // (the K loop ended with vmcnt(0) + barrier: nobody reads the operand tiles any more)
        const float floor_v = g.relu ? 0.f : -__builtin_inff();
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            const int jl = (BN / WN) * wc + 16 * nt + g4;
            const int j = j0 + jl;
            const float4 bj = load_bias4(g.bias, g.A, j, g.N);
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const int il = (BM / WM) * wr + 16 * mt + li;
                const float v0 = clamp_below(acc[mt][nt][0] + bj.x, floor_v), v1 = clamp_below(acc[mt][nt][1] + bj.y, floor_v);
                const float v2 = clamp_below(acc[mt][nt][2] + bj.z, floor_v), v3 = clamp_below(acc[mt][nt][3] + bj.w, floor_v);
                u32x2 o;
                o[0] = pack_bf16x2(v0, v1);
                o[1] = pack_bf16x2(v2, v3);
                *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(smem + il * PITCH + jl * 2) = o;
            }
        }
        __syncthreads();
        constexpr int CH = BN / 8;                         // 16-B chunks per output row
        constexpr int RL = NT / CH;                        // rows written per pass
        const int c = threadIdx.x % CH, rl = threadIdx.x / CH;
        const int j = j0 + c * 8;
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (rl < RL && j < g.N) {
            bf16_t* Cb = reinterpret_cast<bf16_t*>(g.C);
            for (int r = rl; r < BM; r += RL) {
                const int i = i0 + r;
                if (i >= g.M) break;
                const u32x4 lv = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(smem + r * PITCH + c * 16);
                uint4 v = make_uint4(lv[0], lv[1], lv[2], lv[3]);
                if (g.relu_src != nullptr) {
                    const uint4 h = *reinterpret_cast<const uint4*>(g.relu_src + (int64_t)i * g.ld_relu + j);
                    // keep where the saved activation is > 0 (sign clear, magnitude non-zero), per bf16 half
                    auto keep = [](uint32_t val, uint32_t hh) -> uint32_t {
                        const uint32_t lo = ((hh & 0x8000u) == 0 && (hh & 0x7fffu) != 0) ? 0x0000ffffu : 0u;
                        const uint32_t hi = ((hh & 0x80000000u) == 0 && (hh & 0x7fff0000u) != 0) ? 0xffff0000u : 0u;
                        return val & (lo | hi);
                    };
                    v.x = keep(v.x, h.x); v.y = keep(v.y, h.y); v.z = keep(v.z, h.z); v.w = keep(v.w, h.w);
                }
                *reinterpret_cast<uint4*>(Cb + (int64_t)i * g.ldc + j) = v;
                if (g.colsum_part != nullptr) {
                    cs[0] += bf16_to_f32((bf16_t)(v.x & 0xffff)); cs[1] += bf16_to_f32((bf16_t)(v.x >> 16));
                    cs[2] += bf16_to_f32((bf16_t)(v.y & 0xffff)); cs[3] += bf16_to_f32((bf16_t)(v.y >> 16));
                    cs[4] += bf16_to_f32((bf16_t)(v.z & 0xffff)); cs[5] += bf16_to_f32((bf16_t)(v.z >> 16));
                    cs[6] += bf16_to_f32((bf16_t)(v.w & 0xffff)); cs[7] += bf16_to_f32((bf16_t)(v.w >> 16));
                }
            }
        }
        if (g.colsum_part != nullptr) {
            // column sums of the stored tile: [RL][BN] partials through LDS, then one plain store per column into this
            // tile row's partial-sum row (added up in a fixed order by the bias-finish kernel: no float atomics)
            __syncthreads();
            float* red = reinterpret_cast<float*>(smem_raw);
            static_assert(RL * BN * 4 <= 2 * BUF_BYTES, "reduction scratch must fit");
            if (rl < RL) {
#pragma unroll
                for (int k = 0; k < 8; ++k) red[rl * BN + c * 8 + k] = cs[k];
            }
            __syncthreads();
            for (int col = threadIdx.x; col < BN; col += NT) {
                float sum = 0.f;
                for (int r = 0; r < RL; ++r) sum += red[r * BN + col];
                if (j0 + col < g.N) g.colsum_part[(int64_t)tm * g.N + j0 + col] = sum;
            }
        }
        return;
    }
    // fp32 output (split-K slabs of the weight gradient, fp32 y): same idea as the bf16 path, in two
    // row halves because the fp32 tile is twice the staging space; 16 B per lane, whole rows.
    {
        constexpr int NT = 64 * NW;
        constexpr int HR = BM / 2;
        constexpr int PITCH = BN * 4 + 16;
        static_assert(HR * PITCH <= 2 * BUF_BYTES, "fp32 half tile must fit in the staging buffers");
        static_assert((BM / WM) <= HR, "a wave's rows must lie in one half");
        constexpr int CH = BN / 4, RL = NT / CH;          // float4 chunks per row, rows per pass
        float* Cf = reinterpret_cast<float*>(g.C);
        if (g.split_k > 1) Cf += (int64_t)z * g.M * g.ldc;
        const int c = threadIdx.x % CH, rl = threadIdx.x / CH;
        const int j = j0 + c * 4;
        float sq = 0.f;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            if (((BM / WM) * wr) / HR == hh) {
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) {
                    const int jl = (BN / WN) * wc + 16 * nt + g4;
                    const float4 bj = load_bias4(g.bias, g.A, j0 + jl, g.N);
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt) {
                        const int il = (BM / WM) * wr + 16 * mt + li - hh * HR;
                        f32x4 v = acc[mt][nt];
                        v[0] += bj.x; v[1] += bj.y; v[2] += bj.z; v[3] += bj.w;
                        if (g.relu) {
                            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                        }
                        *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(smem + il * PITCH + jl * 4) = v;
                    }
                }
            }
            __syncthreads();
            if (rl < RL && j < g.N) {
                for (int r = rl; r < HR; r += RL) {
                    const int i = i0 + hh * HR + r;
                    if (i >= g.M) break;
                    const f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(smem + r * PITCH + c * 16);
                    *reinterpret_cast<float4*>(Cf + (int64_t)i * g.ldc + j) = make_float4(v[0], v[1], v[2], v[3]);
                    sq += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                }
            }
            __syncthreads();
        }
        if (g.sumsq_slots != nullptr) {                  // clip_grad_norm_'s sum g^2 over this tile (grouped weight gradients)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
            float* red = reinterpret_cast<float*>(smem_raw);
            if (lane == 0) red[w] = sq;
            __syncthreads();
            if (threadIdx.x == 0) {
                float t = 0.f;
                for (int ww = 0; ww < NW; ++ww) t += red[ww];
                atomicAdd(g.sumsq_slots + (blockIdx.x & (CODAE_S_N_SLOTS - 1)), (double)t);
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int A_MODE, int B_MODE, bool C_F32, bool LOSS = false, int NS = 2>
__global__ __launch_bounds__(64 * WM * WN, (64 * WM * WN) / 256 * ((NS * (BM + BN) * 128 <= 80 * 1024) ? 2 : 1))
void gemm_bf16_kernel(GemmBf16 g, int tiles_n, int tiles_mn, int kt_total) {
    // LOSS: + {dataset row, mask id} of the tile's rows, fetched once at entry (two dependent loads that
    // would otherwise sit in front of every row of the epilogue)
    __shared__ __attribute__((aligned(16))) char smem_raw[NS * (BM + BN) * 128 + (LOSS ? BM * 8 : 0)];
    gemm_bf16_tile<BM, BN, WM, WN, A_MODE, B_MODE, C_F32, LOSS, NS>(g, tiles_n, tiles_mn, kt_total, blockIdx.x, gridDim.x, smem_raw);
}

// Several weight-gradient GEMMs (k-strided operands, fp32 output) in one launch: workgroup -> (GEMM, tile) by the prefix
// sums in the descriptor block.  128 x 128 tiles: the narrow stacks this exists for have 9-16 tiles per layer.
template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_bf16_grouped_kernel(GemmBf16Group grp) {
    __shared__ __attribute__((aligned(16))) char smem_raw[2 * (BM + BN) * 128];
    int j = 0;
    while (j + 1 < grp.n && (int)blockIdx.x >= grp.wg_begin[j + 1]) ++j;
    const GemmBf16& g = grp.g[j];
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    gemm_bf16_tile<BM, BN, 2, 2, OP_KS, OP_KS, true, false>(g, tiles_n, tiles_m * tiles_n, g.K / BK, blockIdx.x - grp.wg_begin[j],
                                                            grp.wg_begin[j + 1] - grp.wg_begin[j], smem_raw);
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const GemmBf16& g, hipStream_t s) {
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const int kt_total = g.K / BK;
    const int64_t nwg = (int64_t)tiles_m * tiles_n * g.split_k;
    CODAE_REQUIRE(nwg < (1 << 30), "gemm_bf16: grid too large");
    dim3 grid((unsigned)nwg), block(64 * WM * WN);
#define LAUNCH(AM, BMODE, CF) \
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, AM, BMODE, CF>), grid, block, 0, s, g, tiles_n, tiles_m * tiles_n, kt_total)
    if (g.loss.enabled) {
        hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, OP_KC, OP_KC, false, true>), grid, block, 0, s, g, tiles_n,
                           tiles_m * tiles_n, kt_total);
    } else if (g.a_mode == OP_KC && g.b_mode == OP_KC) {
        // forward / data-gradient form of a launch that cannot fill the chip: four stages (see gemm_bf16_tile)
        const bool deep = !g.c_f32 && nwg <= 256 && kt_total >= 6 && !env().no_deep_small && env().small_stages >= 4;
        if (g.c_f32) LAUNCH(OP_KC, OP_KC, true);
        else if (deep)
            hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, OP_KC, OP_KC, false, false, 4>), grid, block, 0, s, g, tiles_n,
                               tiles_m * tiles_n, kt_total);
        else LAUNCH(OP_KC, OP_KC, false);
    }
    else if (g.a_mode == OP_KC && g.b_mode == OP_KS) { if (g.c_f32) LAUNCH(OP_KC, OP_KS, true); else LAUNCH(OP_KC, OP_KS, false); }
    else if (g.a_mode == OP_KS && g.b_mode == OP_KS) { if (g.c_f32) LAUNCH(OP_KS, OP_KS, true); else LAUNCH(OP_KS, OP_KS, false); }
    else { set_error("gemm_bf16: operand mode combination not built"); return CODAE_E_UNSUPPORTED; }
#undef LAUNCH
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

}  // namespace

// Tile of the grouped launch: 128 x 128 when that alone gives every CU a workgroup, else 64 x 128, else 64 x 64 (C2's ten
// 384 x 384 weight gradients: 90 / 180 / 360 workgroups).
int gemm_bf16_grouped(GemmBf16Group& grp, hipStream_t s) {
    CODAE_REQUIRE(grp.n >= 1 && grp.n <= CODAE_GROUP_MAX, "gemm_bf16_grouped: %d GEMMs", grp.n);
    auto count = [&](int bm, int bn) {
        int total = 0;
        for (int j = 0; j < grp.n; ++j) {
            const GemmBf16& g = grp.g[j];
            grp.wg_begin[j] = total;
            total += ((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * g.split_k;
        }
        grp.wg_begin[grp.n] = total;
        return total;
    };
    for (int j = 0; j < grp.n; ++j) {
        const GemmBf16& g = grp.g[j];
        CODAE_REQUIRE(g.a_mode == OP_KS && g.b_mode == OP_KS && g.c_f32 && g.split_k >= 1 && g.K % BK == 0 && g.K >= BK && g.M % 8 == 0 &&
                          g.N % 8 == 0 && !g.loss.enabled && g.relu_src == nullptr && g.colsum_part == nullptr && g.bias == nullptr,
                      "gemm_bf16_grouped: GEMM %d is not a plain weight-gradient form", j);
    }
    int tile = env().group_tile;
    if (tile < 0) tile = count(128, 128) >= 256 ? 0 : (count(64, 128) >= 256 ? 1 : 2);
    if (tile == 0) { const int total = count(128, 128); hipLaunchKernelGGL((gemm_bf16_grouped_kernel<128, 128>), dim3(total), dim3(256), 0, s, grp); }
    else if (tile == 1) { const int total = count(64, 128); hipLaunchKernelGGL((gemm_bf16_grouped_kernel<64, 128>), dim3(total), dim3(256), 0, s, grp); }
    else { const int total = count(64, 64); hipLaunchKernelGGL((gemm_bf16_grouped_kernel<64, 64>), dim3(total), dim3(256), 0, s, grp); }
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

bool gemm_bf16_supported(int M, int N, int K) {
    // K: whole BK tiles; N: 16-byte rows for vector epilogue / staged chunks
    return M > 0 && N >= 8 && K >= BK && (K % BK) == 0 && (N % 8) == 0;
}

// Tile choice: the 256 x 192 / 8-wave tile when it yields enough workgroups to cover the chip
// (its operand traffic per flop is 1.7x lower than the 128 x 128 tile's, which is what bounds
// the small tile: ~39 TB/s of L2 reads at full MFMA rate); else 128 x 128 / 4 waves.
int gemm_bf16_tile_big(int M, int N, int split_k, bool k_strided = false) {
    if (env().gemm_tile >= 0) return env().gemm_tile;
    const int64_t big = (int64_t)((M + 255) / 256) * ((N + 191) / 192) * split_k;
    // round 1 (tools/bench_gemm.py, 8192 x 1536 x 1536, us): 128 x 128 (s) 46.3 / 59.6 / 68.1 for forward / dgrad through W /
    // wgrad + reduce; one-barrier 256 x 192 (b) 38.4 / 46.4 / 55.1; two 128 x 192 workgroups per CU (c) 40.2 / 44.9 / 55.3;
    // 4-wave pipelined (p) slower than 8-wave (q) 37.0 / 43.3 / 53.9.  b, c and p were pruned in round 2.
    // r02, same box, interleaved (us): tile                                   q     x
    //   forward                                                             34.3  33.9
    //   dgrad through W itself                                              45.6  44.9
    //   wgrad + slab reduce                                                 50.7  49.3      whole step 1.398 -> 1.362 ms
    // (x = q with every LDS-DMA piece issued by waves 0..3, one per SIMD: the SIMD partner multiplies while its
    // neighbour's issue slots are taken by the VMEM instructions)
    (void)k_strided;
    if (big < 160) return 0;
    return 6;
}

// Forward / data-gradient form (k-contiguous operands, bf16 out) of a launch with at most 200 tiles of 128 x 128 - a small
// batch of a wide layer, or a narrow layer: 64 x 64 tiles.  One wave per SIMD issues every LDS-DMA piece, fragment read and
// MFMA of its tile itself, so the time per K-tile is the wave's instruction stream (128 x 128: 8 pieces + 16 reads + 32
// MFMAs = 0.75 us per K-tile even with hot weights: 17-22 us for 128 x 1536 x 1536); a quarter of the tile per wave and
// four times the workgroups divide it.  tools/abl/small_tile_sweep.sh, us, 128 x 128 / 64 x 64 tiles: N = K = 1536, M = 256: 21.8 /
// 9.1; 512: 22.0 / 9.8; 1024: 22.5 / 15.7; 2048 (192 tiles): 23.4 / 18.3; 4096 (384 tiles): 27.7 / 35.5.  N = K = 384, M = 1024:
// 9.1 / 4.9; 4096: 9.4 / 5.4; 8192 (192 tiles): 9.9 / 7.5.
static bool small_tile_64(const GemmBf16& g) {
    if (env().no_deep_small || g.c_f32 || g.a_mode != OP_KC || g.b_mode != OP_KC || g.split_k != 1) return false;
    return (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128) <= env().small_tile_max;
}

// Which kernel a (non-loss) launch takes: the ONE place that decides, used by gemm_bf16() itself and by whoever has to know in
// advance (gemm_bf16_takes_relu_bits).  > 0: the pipelined kernel with that cfg (1 / 6: 256 x 192, 7: 128 x 192);
// -64 / -128: the one-barrier kernel on 64 x 64 / 128 x 128 tiles.
static int bf16_path(const GemmBf16& g) {
    const int t = gemm_bf16_tile_big(g.M, g.N, g.split_k, g.b_mode == OP_KS || g.c_f32);
    switch (t) {
        case 3: return 1;               // 256 x 192, 8 waves, phase-pipelined, every wave loads
        case 6: return 6;               // 256 x 192, 8 waves, LDS-DMA issued by one wave per SIMD (default)
        case 7: return 7;               // (forced: 128 x 192 pipelined, forward form only)
        default:                        // small problems: one-barrier double buffer
            if (small_tile_64(g)) return -64;
            // between the 64 x 64 tiles and the 256 x 192 tile, forward / data-gradient form: the pipelined kernel on 128 x 192
            // tiles (tools/bench_gemm_fwd.py, us, 128 x 128 one-barrier / 128 x 192 pipelined: 4096 x 1536 x 1536 28.3 / 21.9,
            // 3000 x 1536 x 1536 26.4 / 19.8, 8192 x 768 x 768 17.9 / 14.6; same bits)
            if (g.a_mode == OP_KC && g.b_mode == OP_KC && !g.c_f32 && g.split_k == 1 && !env().no_deep_small) return 7;
            return -128;
    }
}

// a forward-form (k-contiguous operands, bf16 out, unsplit) launch with this output shape goes to a pipelined kernel (256 x 192 or
// 128 x 192 tiles), whose epilogues write / read the 1-bit ReLU mask
bool gemm_bf16_takes_relu_bits(int M, int N) {
    if (env().gemm_dbg) return false;
    GemmBf16 probe{};
    probe.M = M; probe.N = N; probe.a_mode = OP_KC; probe.b_mode = OP_KC; probe.c_f32 = 0; probe.split_k = 1;
    return bf16_path(probe) > 0;
}

// rows of g.colsum_part the launch gemm_bf16(g) makes will write: one per tile along M of the tile it picks
int gemm_bf16_colsum_rows(const GemmBf16& g) {
    const int t = gemm_bf16_tile_big(g.M, g.N, g.loss.enabled ? 1 : g.split_k, g.b_mode == OP_KS || g.c_f32);
    const int bm = t ? 256 : (small_tile_64(g) ? 64 : 128);
    return (g.M + bm - 1) / bm;
}

int gemm_bf16_loss_parts(const GemmBf16& g) {
    const int t = gemm_bf16_tile_big(g.M, g.N, 1);
    const int sm = small_tile_64(g) ? 64 : 128;
    const int bm = t ? 256 : sm, bn = t ? 192 : sm;
    return ((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn);
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
    CODAE_REQUIRE(g.split_k <= g.K / BK, "gemm_bf16: split_k %d > k tiles %d", g.split_k, g.K / BK);
    CODAE_REQUIRE(!g.c_f32 || (g.relu_src == nullptr && g.colsum_part == nullptr), "gemm_bf16: ReLU mask / column sums need bf16 output");
    if (g.loss.enabled) {
        CODAE_REQUIRE(g.a_mode == OP_KC && g.b_mode == OP_KC && !g.c_f32 && g.split_k == 1 && !g.relu && !g.relu_src,
                      "gemm_bf16: fused loss only on the plain forward form");
        CODAE_REQUIRE(g.loss.data && g.loss.parts && g.loss.io == g.N && g.loss.B <= g.M && (g.N % 8) == 0,
                      "gemm_bf16: fused loss arguments");
        CODAE_REQUIRE(!(g.loss.mask_id || g.loss.mask_to_use) || ((reinterpret_cast<uintptr_t>(g.loss.table) & 7) == 0),
                      "gemm_bf16: mask table must be 8-byte aligned");
        CODAE_REQUIRE((reinterpret_cast<uintptr_t>(g.loss.data) & 15) == 0, "gemm_bf16: dataset must be 16-byte aligned");
    }
    if (env().gemm_dbg) { GemmBf16 g2 = g; g2.dbg = env().gemm_dbg; return gemm_bf16_pipe(g2, 0, s); }
    if (g.loss.enabled) {
        const int t = gemm_bf16_tile_big(g.M, g.N, 1);
        if (t) return gemm_bf16_pipe(g, t >= 6 ? 6 : 1, s);       // 8-wave pipelined kernel, loss from the accumulators
        if (small_tile_64(g)) return launch_cfg<64, 64, 2, 2>(g, s);
        return launch_cfg<128, 128, 2, 2>(g, s);
    }
    const int path = bf16_path(g);
    if (path > 0) return gemm_bf16_pipe(g, path, s);
    if (path == -64) return launch_cfg<64, 64, 2, 2>(g, s);
    return launch_cfg<128, 128, 2, 2>(g, s);
}

#if defined(CODAE_DBG_5D) && (CODAE_DBG_5D & 2)
}  // namespace codae
extern "C" int codae_debug_5d(float* host_out, int n_floats) {
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(codae::g_dbg5d), (size_t)n_floats * sizeof(float)) != hipSuccess) return -1;
    return 0;
}
namespace codae {
#endif
}  // namespace codae
