// Persistent fused chain for NARROW stacks (every width <= 512): the whole per-sample part of a training step -
// batch gather + slot corruption, all forward layers, the MSE loss with its gradient and metric sums, and the whole
// data-gradient chain with the bias-gradient partial sums - in ONE launch (script/train_dae_on_embedding.py:198-210 of
// the reference, minus the weight gradients).  BASELINE config C2 (3 x 128, batch 1024) and the reference's stock batch
// sizes are launch-bound on the per-layer path: ~55 launches of 6-15 us each against 9 GFLOP of arithmetic (round 1:
// 0.31 ms / step, 0.27 ms of it host enqueue time).
//
// Decomposition: rows.  A workgroup (4 waves) owns 16 batch rows and walks them through every layer; nothing is ever
// exchanged between workgroups, so there is no grid barrier and no cross-workgroup visibility protocol.  The 16 x width
// activation panel lives in LDS (two ping-pong buffers), the weights are never staged: each wave owns a quarter of the
// layer's output columns and streams its weight fragments straight from L2 into registers (a narrow stack's bf16
// weights - 2.9 MB at 10 x 384^2 - sit in every XCD's 4 MiB L2), since no other wave of the workgroup would read the
// same fragment.  v_mfma_f32_16x16x32_bf16 with swapped operands, as in the big GEMM kernels: a lane ends up with 4
// consecutive output columns of one row.  Per layer the panel's result goes to the other LDS buffer (input of the
// next layer) and, in whole rows, to HBM: the weight-gradient GEMMs (one grouped launch afterwards) need H_l and dA_l.
// The data-gradient chain is the same loop over the TRANSPOSED weight shadow with the ReLU mask taken from the saved
// activation and the column sums of dA (bias gradient) written as one partial row per workgroup.
//
// What bounds it: every workgroup streams all weights once per direction (C2: 2 x 2.9 MB per workgroup, 64 workgroups),
// i.e. the per-CU L2 -> register rate, not MFMA (72 MFMAs per wave and layer) and not HBM.
#include "codae_common.h"

namespace codae {
namespace {

constexpr int CH_ROWS = 16;          // batch rows per workgroup = one MFMA tile
constexpr int CH_NW = 4;             // waves: each owns width / 4 output columns
constexpr int CH_NT = 64 * CH_NW;
constexpr int CH_MAXW = CODAE_CHAIN_MAX_WIDTH;
constexpr int CH_MAXT = CH_MAXW / (16 * CH_NW);     // 16-column MFMA tiles per wave at the widest layer
constexpr int CH_PITCH = CH_MAXW * 2 + 16;          // bytes per panel row: + 16 so that the 16 rows of an A fragment
                                                    // read (ds_read_b128) fall on different banks

typedef __attribute__((address_space(3))) char lds_char;

// One layer: out[16][N] = in[16][K] . W[N][K]^T (fp32 accumulate).  `in` is the LDS panel; W is k-contiguous per output
// column.  acc[t] = tile t of this wave's T column tiles [n_lo + 16 t, +16); lane: row lane & 15, columns 4 (lane >> 4) .. +3.
// The weight fragments of K-trip i+1 (64 deep: 2 T loads of 1 KiB per wave) are requested before trip i is multiplied:
// the kernel is bound by how many L2 requests a CU keeps in flight (first version, loads and MFMAs of one trip back to
// back: 17 us per layer; every workgroup streams the layer's whole weight matrix).
template <int T>
__device__ __forceinline__ void chain_matmul_t(f32x4 (&acc)[CH_MAXT], const lds_char* in, const bf16_t* __restrict__ W, int K, int n_lo,
                                               int lane) {
    const int r = lane & 15, kc = lane >> 4;
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const lds_char* arow = in + r * CH_PITCH + kc * 16;
    const bf16_t* wrow = W + (int64_t)(n_lo + r) * K + kc * 8;
    const int64_t wtile = (int64_t)16 * K;
    s16x8 bx[2][T], by[2][T];                      // two register sets: trip i in use, trip i + 1 in flight
    auto load = [&](s16x8 (&b0)[T], s16x8 (&b1)[T], int k0) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const bf16_t* p = wrow + t * wtile + k0;
            b0[t] = *reinterpret_cast<const s16x8*>(p);
            b1[t] = *reinterpret_cast<const s16x8*>(p + 32);
        }
    };
    auto mul = [&](const s16x8 (&b0)[T], const s16x8 (&b1)[T], int k0) {
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const __attribute__((address_space(3))) s16x8*>(arow + k0 * 2));
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const __attribute__((address_space(3))) s16x8*>(arow + k0 * 2 + 64));
#pragma unroll
        for (int t = 0; t < T; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b0[t]), a0, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b1[t]), a1, acc[t], 0, 0, 0);
        }
    };
    load(bx[0], bx[1], 0);
    int k0 = 0;
    for (; k0 + 128 <= K; k0 += 128) {             // two trips per iteration: the register sets swap roles statically
        load(by[0], by[1], k0 + 64);
        mul(bx[0], bx[1], k0);
        if (k0 + 128 < K) load(bx[0], bx[1], k0 + 128);
        mul(by[0], by[1], k0 + 64);
    }
    if (k0 < K) mul(bx[0], bx[1], k0);             // odd number of 64-deep trips
}

__device__ __forceinline__ void chain_matmul(f32x4 (&acc)[CH_MAXT], const lds_char* in, const bf16_t* __restrict__ W, int K, int n_lo,
                                             int n_tiles, int lane) {
    switch (n_tiles) {                              // wave-uniform: width / 64
        case 1: chain_matmul_t<1>(acc, in, W, K, n_lo, lane); break;
        case 2: chain_matmul_t<2>(acc, in, W, K, n_lo, lane); break;
        case 3: chain_matmul_t<3>(acc, in, W, K, n_lo, lane); break;
        case 4: chain_matmul_t<4>(acc, in, W, K, n_lo, lane); break;
        case 5: chain_matmul_t<5>(acc, in, W, K, n_lo, lane); break;
        case 6: chain_matmul_t<6>(acc, in, W, K, n_lo, lane); break;
        case 7: chain_matmul_t<7>(acc, in, W, K, n_lo, lane); break;
        default: chain_matmul_t<8>(acc, in, W, K, n_lo, lane); break;
    }
}

// panel rows -> HBM, whole rows, 16 B per lane
__device__ __forceinline__ void panel_to_global(const lds_char* panel, bf16_t* __restrict__ dst, int row0, int rows_total, int width) {
    const int chunks = width / 8;
    for (int q = threadIdx.x; q < CH_ROWS * chunks; q += CH_NT) {
        const int r = q / chunks, c = q - r * chunks;
        if (row0 + r < rows_total) {
            const u32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(panel + r * CH_PITCH + c * 16);
            *reinterpret_cast<uint4*>(dst + (int64_t)(row0 + r) * width + c * 8) = make_uint4(v[0], v[1], v[2], v[3]);
        }
    }
}

__global__ __launch_bounds__(CH_NT, 2) void chain_step_kernel(ChainArgs a) {
    __shared__ __attribute__((aligned(16))) char smem_raw[2 * CH_ROWS * CH_PITCH + CH_ROWS * CH_MAXW * 4 + CH_ROWS * 8 + 64];
    lds_char* smem = (lds_char*)smem_raw;
    lds_char* panel[2] = {smem, smem + CH_ROWS * CH_PITCH};
    float* xf = reinterpret_cast<float*>(smem_raw + 2 * CH_ROWS * CH_PITCH);                  // target rows, fp32 [16][io]
    int* rowinfo = reinterpret_cast<int*>(smem_raw + 2 * CH_ROWS * CH_PITCH + CH_ROWS * CH_MAXW * 4);
    float* wred = reinterpret_cast<float*>(rowinfo + 2 * CH_ROWS);                             // 2 x 4 wave partials

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, g4 = (lane >> 4) * 4;
    const int row0 = blockIdx.x * CH_ROWS;
    const int io = a.width[0];
    const bool masked = a.mask_id != nullptr || a.mask_to_use != nullptr;

    // ---- batch gather + slot corruption (data_tool.py:96-103, embedding_...py:226-239): x (fp32) -> LDS, x * mask (bf16) ->
    //      panel 0 and act[0]; rows past the batch are zero
    if (threadIdx.x < CH_ROWS) {
        const int i = row0 + threadIdx.x;
        int src = -1, id = 0;
        if (i < a.B) {
            src = a.row_idx ? a.row_idx[i] : i;
            if (a.mask_id != nullptr) id = a.mask_id[i];
            else if (a.mask_to_use != nullptr) id = a.mask_to_use[(int64_t)src * a.nb_run + a.run];
        }
        rowinfo[2 * threadIdx.x] = src; rowinfo[2 * threadIdx.x + 1] = id;
    }
    __syncthreads();
    {
        const int chunks = io / 4;
        for (int q = threadIdx.x; q < CH_ROWS * chunks; q += CH_NT) {
            const int r = q / chunks, c = (q - r * chunks) * 4;
            const int src = rowinfo[2 * r], id = rowinfo[2 * r + 1];
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            uint32_t m = 0x01010101u;
            if (src >= 0) {
                x = *reinterpret_cast<const float4*>(a.data + (int64_t)src * io + c);
                if (masked) m = *reinterpret_cast<const uint32_t*>(a.mask_table + (int64_t)id * io + c);
            }
            *reinterpret_cast<float4*>(xf + r * io + c) = x;
            u32x2 o;
            o[0] = pack_bf16x2((m & 0xffu) ? x.x : 0.f, (m & 0xff00u) ? x.y : 0.f);
            o[1] = pack_bf16x2((m & 0xff0000u) ? x.z : 0.f, (m & 0xff000000u) ? x.w : 0.f);
            *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(panel[0] + r * CH_PITCH + c * 2) = o;
        }
    }
    __syncthreads();
    panel_to_global(panel[0], a.act[0], row0, a.rows, io);

    f32x4 acc[CH_MAXT];
    int cur = 0;
    // ---- forward chain (embedding_...py:137-185) ----------------------------------------------------------------------
    for (int l = 0; l < a.L; ++l) {
        const int K = a.width[l], N = a.width[l + 1];
        const int n_tiles = N / (16 * CH_NW), n_lo = w * (N / CH_NW);
        chain_matmul(acc, panel[cur], a.W[l], K, n_lo, n_tiles, lane);
        const bool last = l == a.L - 1;
        lds_char* out = panel[cur ^ 1];
        if (!last) {
            const float floor_v = a.relu[l] ? 0.f : -__builtin_inff();
#pragma unroll
            for (int t = 0; t < CH_MAXT; ++t)
                if (t < n_tiles) {
                    const int j = n_lo + 16 * t + g4;
                    const float4 bj = *reinterpret_cast<const float4*>(a.bias[l] + j);
                    u32x2 o;
                    o[0] = pack_bf16x2(clamp_below(acc[t][0] + bj.x, floor_v), clamp_below(acc[t][1] + bj.y, floor_v));
                    o[1] = pack_bf16x2(clamp_below(acc[t][2] + bj.z, floor_v), clamp_below(acc[t][3] + bj.w, floor_v));
                    *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(out + li * CH_PITCH + j * 2) = o;
                }
            __syncthreads();
            panel_to_global(out, a.act[l + 1], row0, a.rows, N);
        } else {
            // ---- MSE loss, its gradient and the metric sums from the accumulators (train_dae_on_embedding.py:206-223)
            const int src = rowinfo[2 * li], id = rowinfo[2 * li + 1];
            const bool live = src >= 0;
            float sq = 0.f, sqp = 0.f;
#pragma unroll
            for (int t = 0; t < CH_MAXT; ++t)
                if (t < n_tiles) {
                    const int j = n_lo + 16 * t + g4;
                    const float4 bj = *reinterpret_cast<const float4*>(a.bias[l] + j);
                    const float4 x = *reinterpret_cast<const float4*>(xf + li * io + j);
                    uint32_t m = 0x01010101u;
                    if (masked) m = *reinterpret_cast<const uint32_t*>(a.mask_table + (int64_t)id * io + j);
                    const float yv[4] = {acc[t][0] + bj.x, acc[t][1] + bj.y, acc[t][2] + bj.z, acc[t][3] + bj.w};
                    const float xs[4] = {x.x, x.y, x.z, x.w};
                    float gq[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float d = live ? xs[k] - yv[k] : 0.f;
                        const float se = d * d;
                        sq += se;
                        sqp += ((m >> (8 * k)) & 0xffu) == 0 ? se : 0.f;
                        gq[k] = live ? -2.f * d * a.inv_n : 0.f;        // (+0 in the pad rows, like the per-layer kernels)
                    }
                    u32x2 o;
                    o[0] = pack_bf16x2(gq[0], gq[1]);
                    o[1] = pack_bf16x2(gq[2], gq[3]);
                    *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(out + li * CH_PITCH + j * 2) = o;
                    // last bias gradient: column sums of the unrounded dy over the panel's 16 rows
                    float c4[4] = {gq[0], gq[1], gq[2], gq[3]};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        c4[k] += __shfl_xor(c4[k], 1); c4[k] += __shfl_xor(c4[k], 2);
                        c4[k] += __shfl_xor(c4[k], 4); c4[k] += __shfl_xor(c4[k], 8);
                    }
                    if (li == 0) *reinterpret_cast<float4*>(a.colsum_part[l] + (int64_t)blockIdx.x * N + j) = make_float4(c4[0], c4[1], c4[2], c4[3]);
                }
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) { sq += __shfl_xor(sq, o2); sqp += __shfl_xor(sqp, o2); }
            if (lane == 0) { wred[2 * w] = sq; wred[2 * w + 1] = sqp; }
            __syncthreads();
            if (threadIdx.x == 0) {
                float s0 = 0.f, s1 = 0.f;
                for (int ww = 0; ww < CH_NW; ++ww) { s0 += wred[2 * ww]; s1 += wred[2 * ww + 1]; }
                a.loss_parts[2 * blockIdx.x] = (double)s0;
                a.loss_parts[2 * blockIdx.x + 1] = masked ? (double)s1 : 0.0;
            }
            panel_to_global(out, a.dact[l], row0, a.rows, N);
        }
        cur ^= 1;
        // (the next layer reads panel[cur]; its epilogue overwrites panel[cur ^ 1], which every wave has finished reading:
        //  the barrier above sits between this layer's K loop and the next layer's panel writes)
    }
    if (!a.do_backward) return;

    // ---- data-gradient chain (autograd of the above, train_dae_on_embedding.py:210): dA_{l-1} = (dA_l W_l) * [h_l > 0] ---
    for (int l = a.L - 1; l >= 1; --l) {
        const int K = a.width[l + 1], N = a.width[l];            // contraction over layer l's outputs, result per input
        const int n_tiles = N / (16 * CH_NW), n_lo = w * (N / CH_NW);
        chain_matmul(acc, panel[cur], a.Wt[l], K, n_lo, n_tiles, lane);
        lds_char* out = panel[cur ^ 1];
        const bool relu = a.relu[l - 1] != 0;
        const bf16_t* hrow = a.act[l] + (int64_t)(row0 + li) * N;
#pragma unroll
        for (int t = 0; t < CH_MAXT; ++t)
            if (t < n_tiles) {
                const int j = n_lo + 16 * t + g4;
                u32x2 o;
                o[0] = pack_bf16x2(acc[t][0], acc[t][1]);
                o[1] = pack_bf16x2(acc[t][2], acc[t][3]);
                if (relu) {
                    const uint2 h = *reinterpret_cast<const uint2*>(hrow + j);       // saved activation (row0 + li < rows: padded)
                    auto keep = [](uint32_t val, uint32_t hh) -> uint32_t {
                        const uint32_t lo = ((hh & 0x8000u) == 0 && (hh & 0x7fffu) != 0) ? 0x0000ffffu : 0u;
                        const uint32_t hi = ((hh & 0x80000000u) == 0 && (hh & 0x7fff0000u) != 0) ? 0xffff0000u : 0u;
                        return val & (lo | hi);
                    };
                    o[0] = keep(o[0], h.x); o[1] = keep(o[1], h.y);
                }
                *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(out + li * CH_PITCH + j * 2) = o;
                // bias gradient of layer l-1: column sums of the STORED (bf16) values, as the per-layer kernels take them
                float c4[4] = {bf16_to_f32((bf16_t)(o[0] & 0xffff)), bf16_to_f32((bf16_t)(o[0] >> 16)),
                               bf16_to_f32((bf16_t)(o[1] & 0xffff)), bf16_to_f32((bf16_t)(o[1] >> 16))};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    c4[k] += __shfl_xor(c4[k], 1); c4[k] += __shfl_xor(c4[k], 2);
                    c4[k] += __shfl_xor(c4[k], 4); c4[k] += __shfl_xor(c4[k], 8);
                }
                if (li == 0) *reinterpret_cast<float4*>(a.colsum_part[l - 1] + (int64_t)blockIdx.x * N + j) = make_float4(c4[0], c4[1], c4[2], c4[3]);
            }
        __syncthreads();
        panel_to_global(out, a.dact[l - 1], row0, a.rows, N);
        cur ^= 1;
    }
}

}  // namespace

bool chain_supported(int L, const int* in, const int* out) {
    if (L < 1 || L > CODAE_CHAIN_MAX_LAYERS) return false;
    for (int l = 0; l < L; ++l)
        if (in[l] > CH_MAXW || out[l] > CH_MAXW || in[l] % 64 || out[l] % 64) return false;
    return true;
}

int chain_rows_per_workgroup() { return CH_ROWS; }

int launch_chain_step(const ChainArgs& a, hipStream_t s) {
    CODAE_REQUIRE(a.L >= 1 && a.L <= CODAE_CHAIN_MAX_LAYERS && a.rows % CH_ROWS == 0 && a.rows > 0, "chain: bad arguments");
    CODAE_REQUIRE(a.width[0] == a.width[a.L], "chain: the loss needs output width == input width");
    hipLaunchKernelGGL(chain_step_kernel, dim3(a.rows / CH_ROWS), dim3(CH_NT), 0, s, a);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

}  // namespace codae
