// Persistent fused chain for NARROW stacks (every width <= 512): the whole per-sample part of a training step -
// batch gather + slot corruption, all forward layers, the MSE loss with its gradient and metric sums, and the whole
// data-gradient chain with the bias-gradient partial sums - in ONE launch (script/train_dae_on_embedding.py:198-210 of
// the reference, minus the weight gradients).  BASELINE config C2 (3 x 128, batch 1024) and the reference's stock batch
// sizes are launch-bound on the per-layer path: ~55 launches of 6-15 us each against 9 GFLOP of arithmetic (round 1:
// 0.31 ms / step, 0.27 ms of it host enqueue time).
//
// Decomposition: rows.  A workgroup (4 waves) owns 16 batch rows and walks them through every layer; nothing is ever
// exchanged between workgroups, so there is no grid barrier and no cross-workgroup visibility protocol.  The 16 x width
// activation panel lives in LDS (two ping-pong buffers).  Each wave owns a quarter of every layer's output columns; no
// other wave of the workgroup reads the same weights, so each wave streams them through a PRIVATE LDS ring filled by
// LDS-DMA, running ahead of the multiply across layer boundaries (see "the weight stream" below).  v_mfma_f32_16x16x32_bf16
// with swapped operands, as in the big GEMM kernels: a lane ends up with 4 consecutive output columns of one row.  Per
// layer the panel's result goes to the other LDS buffer (input of the next layer) and, in whole rows, to HBM: the
// weight-gradient GEMMs (one grouped launch afterwards) need H_l and dA_l.  The data-gradient chain is the same loop over
// the TRANSPOSED weight shadow; the ReLU mask of each activation is kept from the forward pass as 4 bits per tile and
// thread in LDS (the same thread owns the same (row, columns) of dA_l), the column sums of dA (bias gradient) are
// written as one partial row per workgroup.  Biases are staged in LDS once: inside the layer loops the only vector
// memory operations are the ring's requests and plain stores, so nothing ever has to wait for the ring to drain.
//
// What bounds it (C2, tools/chain_timeline.py on the CHAIN_ABL=9 build): 3.6 us per 384 x 384 matrix and workgroup,
// whatever the ring depth (10, 12, 13 units measured the same) - a CU takes its weights at ~81 GB/s: every byte is
// written to LDS by the DMA and read back once as a fragment (without the reads: 3.1 us, 96 GB/s; without requests at
// all: 1.5 us).  Epilogues 1.3 us per layer, prologue 8 us, 108 us in all; 64 of the 256 CUs are busy.  Not MFMA (72
// per wave and layer), not HBM (the weights stay in L2).  Negative results kept out of the code: an extra wave per
// workgroup that touched the next matrix's lines ahead of the rings (L2 warmer: 133 -> 130 us, within noise once the
// rings existed), and rotating which columns each workgroup starts with so that the workgroups of an XCD do not ask for
// the same lines at the same time (slower: 3.6 -> 3.9 us per matrix; L2 serves the lockstep requests better).
#include "codae_common.h"

#ifndef CHAIN_ABL
#define CHAIN_ABL 0          // timing ablations (tools/abl): 1 no weight requests, 2 no MFMA, 3 no row stores, 9 timeline stamps
#endif

namespace codae {
namespace {

constexpr int CH_ROWS = 16;          // batch rows per workgroup = one MFMA tile
constexpr int CH_NW = 4;             // waves: each owns width / 4 output columns
constexpr int CH_NT = 64 * CH_NW;
#ifdef CH_ABL_MAXW            // (ablation builds: a deeper ring needs the other LDS regions cut to one configuration)
constexpr int CH_MAXW = CH_ABL_MAXW;
#else
constexpr int CH_MAXW = CODAE_CHAIN_MAX_WIDTH;
#endif
constexpr int CH_MAXL = CODAE_CHAIN_MAX_LAYERS;
#ifdef CH_ABL_MROWS
constexpr int CH_MROWS = CH_ABL_MROWS;
#else
constexpr int CH_MROWS = CH_MAXL;
#endif
constexpr int CH_MAXT = CH_MAXW / (16 * CH_NW);     // 16-column MFMA tiles per wave at the widest layer
constexpr int CH_PITCH = CH_MAXW * 2 + 16;          // bytes per panel row: + 16 so that the 16 rows of an A fragment
                                                    // read (ds_read_b128) fall on different banks
constexpr int CH_JJ = (CH_MAXW + CH_NT - 1) / CH_NT;   // bias staging: columns per thread
constexpr int CH_UNIT = 2048;                       // weight unit: one 16-column tile x 64 k = 16 rows x 128 B
#ifdef CH_ABL_RU
constexpr int CH_RU = CH_ABL_RU;
#else
constexpr int CH_RU = 10;
#endif
//                          // units in a wave's ring (CH_RU - 1 requested ahead of the multiply)
#ifdef CH_ABL_BIAS
constexpr int CH_BIAS = CH_ABL_BIAS;
#else
constexpr int CH_BIAS = CODAE_CHAIN_MAX_BIAS;
#endif
//       // floats of bias staged in LDS (sum of the layers' output widths)

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) const void gvoid;

__device__ __forceinline__ void glds16(const void* gsrc, lds_char* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((gvoid*)gsrc, (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// LDS writes of this wave done, then the workgroup barrier.  NOT __syncthreads(): its fence waits vmcnt(0), which would
// drain the weight ring's prefetch queue at every layer boundary.  Nothing a workgroup writes to global memory is read
// back inside the kernel.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Sum over the 16 lanes of a DPP row (the 16 batch rows of a tile column), result in every lane: 4 DPP adds, no LDS
// (__shfl_xor is ds_bpermute_b32: 96 LDS round trips per data-gradient epilogue, 1 us of the 2.5 us it took).
__device__ __forceinline__ float row16_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});     // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});     // row_mirror
    return v;
}

// ---- the weight stream ---------------------------------------------------------------------------------------------
// A wave multiplies its column tiles against EVERY matrix of the chain (forward layers, then the transposed shadows of
// the data-gradient layers), and no other wave of the workgroup reads the same weights.  So each wave runs a private
// ring of CH_RU units in LDS, filled by LDS-DMA (global_load_lds_dwordx4: 1 KiB per instruction, whole 128-byte lines,
// no VGPRs held while in flight) CH_RU - 1 units ahead of the multiply, ACROSS layer boundaries: while a layer's
// epilogue runs, the first units of the next matrix are already landing.  (First version: fragment-shaped global loads
// into registers, one K-trip ahead - 16 B per lane at a 2 K-byte stride, 25 GB/s per CU, 218 us per C2 step.)
// A unit's LDS image is the k-contiguous half-tile image of the GEMM kernels: row r (output column) at r * 128 B, its
// eight 16-B chunks swizzled by (r >> 1) & 7, the swizzle applied on the SOURCE address of the DMA.
// The kernel reads its argument block through the kernarg segment pointer: indexed with a wave-uniform layer number that
// is an s_load; a by-value struct indexed dynamically is copied to scratch first (and every value read from it is then
// per-lane as far as the compiler knows).
typedef const __attribute__((address_space(4))) ChainArgs* KArgs;

struct WCursor {                 // wave-uniform except off0 / off1: the NEXT unit to request
    const char* base;            // the wave's first weight row of the current matrix, k = 0
    int K, T, trips;             // current matrix: contraction length, the wave's column tiles, K / 64
    int trip, t;                 // position in it (trip-major, the order the multiply consumes)
    int step, steps;             // matrix index in the chain / number of matrices
    int slot;                    // ring slot the next unit goes to
    uint32_t off0, off1;         // per-lane source byte offsets of the unit's two DMA instructions (depend on K)
};

__device__ __forceinline__ void cursor_matrix(WCursor& c, KArgs a, int w, int lane) {
    const bf16_t* W; int K, N;
    if (c.step < a->L) { W = a->W[c.step]; K = a->width[c.step]; N = a->width[c.step + 1]; }
    else { const int l = 2 * a->L - 1 - c.step; W = a->Wt[l]; K = a->width[l + 1]; N = a->width[l]; }
    c.base = reinterpret_cast<const char*>(W + (int64_t)w * (N / CH_NW) * K);
    c.K = K; c.T = N / (16 * CH_NW); c.trips = K / 64; c.trip = 0; c.t = 0;
    const int r0 = lane >> 3, r1 = 8 + (lane >> 3);
    c.off0 = (uint32_t)((r0 * K + (((lane & 7) ^ ((r0 >> 1) & 7)) * 8)) * 2);
    c.off1 = (uint32_t)((r1 * K + (((lane & 7) ^ ((r1 >> 1) & 7)) * 8)) * 2);
}

__device__ __forceinline__ void cursor_issue(WCursor& c, KArgs a, lds_char* ring, int w, int lane) {
    const char* src = c.base + ((int64_t)(16 * c.t) * c.K + c.trip * 64) * 2;          // wave-uniform
    lds_char* dst = ring + c.slot * CH_UNIT;
#if CHAIN_ABL != 1
    glds16(src + c.off0, dst);
    glds16(src + c.off1, dst + 1024);
#endif
    c.slot = c.slot + 1 == CH_RU ? 0 : c.slot + 1;
    if (++c.t == c.T) {
        c.t = 0;
        if (++c.trip == c.trips) {
            if (c.step + 1 < c.steps) { ++c.step; cursor_matrix(c, a, w, lane); }
            else { c.trip = c.trips - 1; c.t = c.T - 1; }      // past the end of the chain: re-request the last unit, so
        }                                                      // that the counted waits stay exact to the last multiply
    }
}

// One layer: out[16][N] = in[16][K] . W[N][K]^T (fp32 accumulate).  `in` is the LDS panel.  acc[t] = tile t of this
// wave's T column tiles; lane: row lane & 15, columns 4 (lane >> 4) .. +3.  Unit u's fragments (bc) were read from the
// ring one unit earlier; per unit: wait for unit u + 1 to have landed, read it, multiply unit u, hand unit u's slot to
// the cursor.  At that wait the wave has requested units up to u + CH_RU - 1, i.e. CH_RU - 2 units (2 instructions each)
// are younger than u + 1.  Other vector memory operations in flight only make the wait earlier than needed.
template <int T>
__device__ __forceinline__ void chain_matmul_t(f32x4 (&acc)[CH_MAXT], const lds_char* in, int K, lds_char* ring, int& cslot,
                                               s16x8 (&bc)[2], WCursor& cur, KArgs a, int w, int lane) {
    if constexpr (T > CH_MAXT) return;
    const int r = lane & 15, kc = lane >> 4;
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const lds_char* arow = in + r * CH_PITCH + kc * 16;
    const int roff0 = r * 128 + ((kc ^ (r >> 1)) << 4), roff1 = r * 128 + (((4 + kc) ^ (r >> 1)) << 4);
    for (int k0 = 0; k0 < K; k0 += 64) {
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const __attribute__((address_space(3))) s16x8*>(arow + k0 * 2));
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const __attribute__((address_space(3))) s16x8*>(arow + k0 * 2 + 64));
#pragma unroll
        for (int t = 0; t < T; ++t) {
            wait_vmcnt<2 * (CH_RU - 2)>();
            const int nslot = cslot + 1 == CH_RU ? 0 : cslot + 1;
            const lds_char* nu = ring + nslot * CH_UNIT;
#ifdef CH_ABL_NOREAD
            const s16x8 bn0 = bc[1], bn1 = bc[0]; (void)nu; (void)roff0; (void)roff1;
#else
            const s16x8 bn0 = *reinterpret_cast<const __attribute__((address_space(3))) s16x8*>(nu + roff0);
            const s16x8 bn1 = *reinterpret_cast<const __attribute__((address_space(3))) s16x8*>(nu + roff1);
#endif
#if CHAIN_ABL != 2
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bc[0]), a0, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bc[1]), a1, acc[t], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);       // the request below overwrites the slot bc came from
            cursor_issue(cur, a, ring, w, lane);
            bc[0] = bn0; bc[1] = bn1;
            cslot = nslot;
        }
    }
}

__device__ __forceinline__ void chain_matmul(f32x4 (&acc)[CH_MAXT], const lds_char* in, int K, int n_tiles, lds_char* ring, int& cslot,
                                             s16x8 (&bc)[2], WCursor& cur, KArgs a, int w, int lane) {
    switch (n_tiles) {                              // wave-uniform: width / 64
        case 1: chain_matmul_t<1>(acc, in, K, ring, cslot, bc, cur, a, w, lane); break;
        case 2: chain_matmul_t<2>(acc, in, K, ring, cslot, bc, cur, a, w, lane); break;
        case 3: chain_matmul_t<3>(acc, in, K, ring, cslot, bc, cur, a, w, lane); break;
        case 4: chain_matmul_t<4>(acc, in, K, ring, cslot, bc, cur, a, w, lane); break;
        case 5: chain_matmul_t<5>(acc, in, K, ring, cslot, bc, cur, a, w, lane); break;
        case 6: chain_matmul_t<6>(acc, in, K, ring, cslot, bc, cur, a, w, lane); break;
        case 7: chain_matmul_t<7>(acc, in, K, ring, cslot, bc, cur, a, w, lane); break;
        default: chain_matmul_t<8>(acc, in, K, ring, cslot, bc, cur, a, w, lane); break;
    }
}

// panel rows -> HBM, whole rows, 16 B per lane
__device__ __forceinline__ void panel_to_global(const lds_char* panel, bf16_t* __restrict__ dst, int row0, int rows_total, int width) {
#if CHAIN_ABL == 3
    return;
#endif
    const int r = threadIdx.x >> 4;                      // 16 lanes x 16 B = 256 contiguous bytes of one row per step
    if (row0 + r >= rows_total) return;
    const lds_char* src = panel + r * CH_PITCH;
    bf16_t* out = dst + (int64_t)(row0 + r) * width;
    for (int c = threadIdx.x & 15; c * 8 < width; c += 16) {
        const u32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(src + c * 16);
        *reinterpret_cast<uint4*>(out + c * 8) = make_uint4(v[0], v[1], v[2], v[3]);
    }
}

#if CHAIN_ABL == 9          // timeline build: workgroup 0 / thread 0 stamps the 100 MHz clock into LDS, copied out at the end
#define CH_STAMP() do { if (blockIdx.x == 0 && threadIdx.x == 0) { stamp_s[n_stamp & 63] = wall_clock64(); ++n_stamp; } } while (0)
#else
#define CH_STAMP() do { } while (0)
#endif

__global__ __launch_bounds__(CH_NT, 1) void chain_step_kernel(ChainArgs a_by_value) {
    KArgs a = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int PANELS = 2 * CH_ROWS * CH_PITCH, RING = CH_NW * CH_RU * CH_UNIT;
    __shared__ __attribute__((aligned(1024))) char smem_raw[RING + PANELS + CH_BIAS * 4 + CH_MROWS * CH_NT * 4 + CH_ROWS * 8 + 64];
    lds_char* smem = (lds_char*)smem_raw;
    lds_char* panel[2] = {smem + RING, smem + RING + CH_ROWS * CH_PITCH};
    float* bias_s = reinterpret_cast<float*>(smem_raw + RING + PANELS);                        // all layers' biases
    uint32_t* mbits = reinterpret_cast<uint32_t*>(smem_raw + RING + PANELS + CH_BIAS * 4);    // [activation][thread]: 4 bits per tile
    int* rowinfo = reinterpret_cast<int*>(smem_raw + RING + PANELS + CH_BIAS * 4 + CH_MROWS * CH_NT * 4);
    float* wred = reinterpret_cast<float*>(rowinfo + 2 * CH_ROWS);                             // 2 x 4 wave partials

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if CHAIN_ABL == 9
    unsigned long long* stamp_s = reinterpret_cast<unsigned long long*>(mbits);       // activation 0 has no mask row
    int n_stamp = 0;
#endif
    CH_STAMP();
    const int li = lane & 15, g4 = (lane >> 4) * 4;
    const int row0 = blockIdx.x * CH_ROWS;
    const int io = a->width[0];
    const bool masked = a->mask_id != nullptr || a->mask_to_use != nullptr;
    const uint32_t relu_flags = a->relu_flags;
    if (a->scalars != nullptr && blockIdx.x == 0 && threadIdx.x < CODAE_S_N_SLOTS) {
        // the norm accumulators of this step (the previous step's Adam, the last to read them, ran before this launch):
        // the grouped weight-gradient launch and the bias finish add to them
        a->scalars[CODAE_S_GRAD_SQ_SLOTS + threadIdx.x] = 0.0;
        if (threadIdx.x == 0) a->scalars[CODAE_S_GRAD_SQ] = 0.0;
    }

    // ---- prologue loads, oldest first (vector memory operations retire in order: whatever is requested after the ring's
    //      first units would wait for all of them): the 16 rows' source rows and mask ids, then every layer's bias
    int src = -1, id = 0;
    if (threadIdx.x < CH_ROWS) {
        const int i = row0 + threadIdx.x;
        if (i < a->B) {
            src = a->row_idx ? a->row_idx[i] : i;
            if (a->mask_id != nullptr) id = a->mask_id[i];
            else if (a->mask_to_use != nullptr) id = a->mask_to_use[(int64_t)src * a->nb_run + a->run];
        }
    }
    float bv[CH_JJ][CH_MAXL];
#pragma unroll
    for (int jj = 0; jj < CH_JJ; ++jj)
#pragma unroll
        for (int l = 0; l < CH_MAXL; ++l) {
            bv[jj][l] = 0.f;
            if (l < a->L && jj * CH_NT + (int)threadIdx.x < a->width[l + 1]) bv[jj][l] = a->bias[l][jj * CH_NT + threadIdx.x];
        }
    CH_STAMP();
    // ---- start the weight stream: the first CH_RU units of the chain --------------------------------------------------
    lds_char* ring = smem + w * (CH_RU * CH_UNIT);
    WCursor cur;
    cur.step = 0; cur.steps = a->L + (a->do_backward ? a->L - 1 : 0); cur.slot = 0;
    cursor_matrix(cur, a, w, lane);
#pragma unroll
    for (int u = 0; u < CH_RU; ++u) cursor_issue(cur, a, ring, w, lane);
    CH_STAMP();
    // ---- biases -> LDS (the epilogues then issue no global loads: a load's wait would also wait for the ring's requests)
#pragma unroll
    for (int jj = 0; jj < CH_JJ; ++jj) {
        int off = 0;
#pragma unroll
        for (int l = 0; l < CH_MAXL; ++l)
            if (l < a->L) {
                if (jj * CH_NT + (int)threadIdx.x < a->width[l + 1]) bias_s[off + jj * CH_NT + threadIdx.x] = bv[jj][l];
                off += a->width[l + 1];
            }
    }
    if (threadIdx.x < CH_ROWS) { rowinfo[2 * threadIdx.x] = src; rowinfo[2 * threadIdx.x + 1] = id; }
    CH_STAMP();
    // ---- batch gather + slot corruption (data_tool.py:96-103, embedding_...py:226-239): x * mask (bf16) -> panel 0 and
    //      act[0]; rows past the batch are zero
    lds_barrier();
    {
        const int chunks = io / 4;
        for (int q = threadIdx.x; q < CH_ROWS * chunks; q += CH_NT) {
            const int r = q / chunks, c = (q - r * chunks) * 4;
            const int src = rowinfo[2 * r], id = rowinfo[2 * r + 1];
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            uint32_t m = 0x01010101u;
            if (src >= 0) {
                x = *reinterpret_cast<const float4*>(a->data + (int64_t)src * io + c);
                if (masked) m = *reinterpret_cast<const uint32_t*>(a->mask_table + (int64_t)id * io + c);
            }
            u32x2 o;
            o[0] = pack_bf16x2((m & 0xffu) ? x.x : 0.f, (m & 0xff00u) ? x.y : 0.f);
            o[1] = pack_bf16x2((m & 0xff0000u) ? x.z : 0.f, (m & 0xff000000u) ? x.w : 0.f);
            *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(panel[0] + r * CH_PITCH + c * 2) = o;
        }
    }
    lds_barrier();
    panel_to_global(panel[0], a->act[0], row0, a->rows, io);

    CH_STAMP();
    // unit 0's fragments
    const int fr = lane & 15, fk = lane >> 4;
    s16x8 bc[2];
    int cslot = 0;
    wait_vmcnt<2 * (CH_RU - 1)>();
    bc[0] = *reinterpret_cast<const __attribute__((address_space(3))) s16x8*>(ring + fr * 128 + ((fk ^ (fr >> 1)) << 4));
    bc[1] = *reinterpret_cast<const __attribute__((address_space(3))) s16x8*>(ring + fr * 128 + (((4 + fk) ^ (fr >> 1)) << 4));

    // stored bf16 pair > 0 (as the per-layer data-gradient kernels test the saved activation): 2 bits
    auto pos2 = [](uint32_t hh) -> uint32_t {          // (sign clear and not zero = positive as a signed integer)
        return ((int16_t)(hh & 0xffffu) > 0 ? 1u : 0u) | ((int32_t)hh >= 0x10000 ? 2u : 0u);
    };

    f32x4 acc[CH_MAXT];
    int cur_p = 0, boff = 0;
    // ---- forward chain (embedding_...py:137-185) ----------------------------------------------------------------------
    for (int l = 0; l < a->L; ++l) {
        const int K = a->width[l], N = a->width[l + 1];
        const int n_tiles = N / (16 * CH_NW), n_lo = w * (N / CH_NW);
        CH_STAMP();
        chain_matmul(acc, panel[cur_p], K, n_tiles, ring, cslot, bc, cur, a, w, lane);
        CH_STAMP();
        const bool last = l == a->L - 1;
        lds_char* out = panel[cur_p ^ 1];
        const float* bl = bias_s + boff;
        if (!last) {
            const float floor_v = ((relu_flags >> l) & 1u) ? 0.f : -__builtin_inff();
            uint32_t bits = 0;
#pragma unroll
            for (int t = 0; t < CH_MAXT; ++t)
                if (t < n_tiles) {
                    const int j = n_lo + 16 * t + g4;
                    const float4 bj = *reinterpret_cast<const float4*>(bl + j);
                    u32x2 o;
                    o[0] = pack_bf16x2(clamp_below(acc[t][0] + bj.x, floor_v), clamp_below(acc[t][1] + bj.y, floor_v));
                    o[1] = pack_bf16x2(clamp_below(acc[t][2] + bj.z, floor_v), clamp_below(acc[t][3] + bj.w, floor_v));
                    *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(out + li * CH_PITCH + j * 2) = o;
                    bits |= (pos2(o[0]) | (pos2(o[1]) << 2)) << (4 * t);
                }
            mbits[(l + 1) * CH_NT + threadIdx.x] = bits;       // this thread owns the same (row, columns) of dA_l in the backward
            lds_barrier();
            panel_to_global(out, a->act[l + 1], row0, a->rows, N);
        } else {
            // ---- MSE loss, its gradient and the metric sums from the accumulators (train_dae_on_embedding.py:206-223)
            const int src = rowinfo[2 * li], id = rowinfo[2 * li + 1];
            const bool live = src >= 0;
            const float* xrow = a->data + (int64_t)(live ? src : 0) * io;
            float sq = 0.f, sqp = 0.f;
#pragma unroll
            for (int t = 0; t < CH_MAXT; ++t)
                if (t < n_tiles) {
                    const int j = n_lo + 16 * t + g4;
                    const float4 bj = *reinterpret_cast<const float4*>(bl + j);
                    const float4 x = *reinterpret_cast<const float4*>(xrow + j);
                    uint32_t m = 0x01010101u;
                    if (masked) m = *reinterpret_cast<const uint32_t*>(a->mask_table + (int64_t)id * io + j);
                    const float yv[4] = {acc[t][0] + bj.x, acc[t][1] + bj.y, acc[t][2] + bj.z, acc[t][3] + bj.w};
                    const float xs[4] = {x.x, x.y, x.z, x.w};
                    float gq[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float d = live ? xs[k] - yv[k] : 0.f;
                        const float se = d * d;
                        sq += se;
                        asm volatile("" : "+v"(sq));     // scalar chain, never packed with op_sel half swaps: gemm_bf16.hip, DESIGN.md 5d
                        sqp += ((m >> (8 * k)) & 0xffu) == 0 ? se : 0.f;
                        gq[k] = live ? -2.f * d * a->inv_n : 0.f;        // (+0 in the pad rows, like the per-layer kernels)
                    }
                    u32x2 o;
                    o[0] = pack_bf16x2(gq[0], gq[1]);
                    o[1] = pack_bf16x2(gq[2], gq[3]);
                    *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(out + li * CH_PITCH + j * 2) = o;
                    // last bias gradient: column sums of the unrounded dy over the panel's 16 rows
                    float c4[4] = {gq[0], gq[1], gq[2], gq[3]};
#pragma unroll
                    for (int k = 0; k < 4; ++k) c4[k] = row16_sum(c4[k]);
                    if (li == 0) *reinterpret_cast<float4*>(a->colsum_part[l] + (int64_t)blockIdx.x * N + j) = make_float4(c4[0], c4[1], c4[2], c4[3]);
                }
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) { sq += __shfl_xor(sq, o2); sqp += __shfl_xor(sqp, o2); }
            if (lane == 0) { wred[2 * w] = sq; wred[2 * w + 1] = sqp; }
            lds_barrier();
            if (threadIdx.x == 0) {
                float s0 = 0.f, s1 = 0.f;
                for (int ww = 0; ww < CH_NW; ++ww) { s0 += wred[2 * ww]; s1 += wred[2 * ww + 1]; }
                a->loss_parts[2 * blockIdx.x] = (double)s0;
                a->loss_parts[2 * blockIdx.x + 1] = masked ? (double)s1 : 0.0;
            }
            panel_to_global(out, a->dact[l], row0, a->rows, N);
        }
        cur_p ^= 1;
        boff += N;
        // (the next layer reads panel[cur_p]; its epilogue overwrites panel[cur_p ^ 1], which every wave has finished
        //  reading: the barrier above sits between this layer's K loop and the next layer's panel writes)
    }

    // ---- data-gradient chain (autograd of the above, train_dae_on_embedding.py:210): dA_{l-1} = (dA_l W_l) * [h_l > 0] ---
    if (a->do_backward)
        for (int l = a->L - 1; l >= 1; --l) {
            const int K = a->width[l + 1], N = a->width[l];            // contraction over layer l's outputs, result per input
            const int n_tiles = N / (16 * CH_NW), n_lo = w * (N / CH_NW);
            CH_STAMP();
            chain_matmul(acc, panel[cur_p], K, n_tiles, ring, cslot, bc, cur, a, w, lane);
            CH_STAMP();
            lds_char* out = panel[cur_p ^ 1];
            const uint32_t bits = ((relu_flags >> (l - 1)) & 1u) ? mbits[l * CH_NT + threadIdx.x] : 0xffffffffu;     // [h_l > 0], from the forward
#pragma unroll
            for (int t = 0; t < CH_MAXT; ++t)
                if (t < n_tiles) {
                    const int j = n_lo + 16 * t + g4;
                    const uint32_t b4 = bits >> (4 * t);
                    u32x2 o;
                    o[0] = pack_bf16x2(acc[t][0], acc[t][1]) & (((b4 & 1u) ? 0x0000ffffu : 0u) | ((b4 & 2u) ? 0xffff0000u : 0u));
                    o[1] = pack_bf16x2(acc[t][2], acc[t][3]) & (((b4 & 4u) ? 0x0000ffffu : 0u) | ((b4 & 8u) ? 0xffff0000u : 0u));
                    *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(out + li * CH_PITCH + j * 2) = o;
                    // bias gradient of layer l-1: column sums of the STORED (bf16) values, as the per-layer kernels take them
                    float c4[4] = {bf16_to_f32((bf16_t)(o[0] & 0xffff)), bf16_to_f32((bf16_t)(o[0] >> 16)),
                                   bf16_to_f32((bf16_t)(o[1] & 0xffff)), bf16_to_f32((bf16_t)(o[1] >> 16))};
#pragma unroll
                    for (int k = 0; k < 4; ++k) c4[k] = row16_sum(c4[k]);
                    if (li == 0) *reinterpret_cast<float4*>(a->colsum_part[l - 1] + (int64_t)blockIdx.x * N + j) = make_float4(c4[0], c4[1], c4[2], c4[3]);
                }
            lds_barrier();
            panel_to_global(out, a->dact[l - 1], row0, a->rows, N);
            cur_p ^= 1;
        }
    CH_STAMP();
#if CHAIN_ABL == 9
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a->colsum_part[0]);
        o[0] = n_stamp;
        for (int i = 0; i < n_stamp && i < 64; ++i) o[1 + i] = stamp_s[i];
    }
#endif
    wait_vmcnt<0>();            // the trailing re-requests of the ring land before the workgroup's LDS is released
}

}  // namespace

bool chain_supported(int L, const int* in, const int* out) {
    if (L < 1 || L > CODAE_CHAIN_MAX_LAYERS) return false;
    for (int l = 0; l < L; ++l)
        if (in[l] > CH_MAXW || out[l] > CH_MAXW || in[l] % 64 || out[l] % 64) return false;
    int bias_floats = 0;
    for (int l = 0; l < L; ++l) bias_floats += out[l];
    return bias_floats <= CH_BIAS;
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
