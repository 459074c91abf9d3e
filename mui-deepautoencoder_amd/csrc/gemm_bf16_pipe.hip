// Phase-pipelined bf16 GEMM for the large contractions of the DAE step (same operand modes and
// epilogue as gemm_bf16.hip; see there for the KC / KS storage modes and the swapped-MFMA
// epilogue).  What changes is the K loop.  Every K-tile (64 deep) is cut into four HALF-TILES
// (A0, A1: the two halves of every wave's rows; B0, B1: likewise for its columns) and four PHASES,
// one output quadrant of each wave per phase.  Two things run ahead of the MFMAs:
//
//   * LDS -> register fragment reads run ONE PHASE ahead: the phase that multiplies quadrant q
//     first issues the ds_reads of the operand half that quadrant q+1 newly needs, into a second
//     register set, so the LDS latency is covered by this phase's 24-32 MFMAs (one wave per SIMD:
//     nobody else would cover it);
//   * HBM/L2 -> LDS DMA (global_load_lds) runs SIX PHASES ahead: a half-tile's LDS region is re-filled, with the
//     same half of the tile two K-tiles later, in the SECOND phase after the one that read it (seven phases ahead
//     = re-fill in the very next phase - was measured too: it forces lgkmcnt(0) at every barrier).
//
//      phase   MFMA          fragment read (for)     waits for      LDS-DMA issued (region read two phases ago)
//      P1(t)   A0 x B0       A1(t)    (P2)           A1(t)          A0(t+2)
//      P2(t)   A1 x B0       B1(t)    (P3)           B1(t)          B0(t+2)
//      P3(t)   A1 x B1       A0(t+1)  (P4 .. P1)     A0(t+1)        A1(t+2)
//      P4(t)   A0 x B1       B0(t+1)  (P1, P2)       B0(t+1)        B1(t+2)
//
// DMA is issued in the natural order A0 B0 A1 B1 of tile 0, 1, 2, ... and retires in order, so the
// half a phase needs is complete once at most the 5 half-tiles issued after it are outstanding:
// one COUNTED s_waitcnt per phase - vmcnt(2a + 3b) where an A half is read, vmcnt(3a + 2b) where a B half is
// (a / b = DMA instructions of this wave per A / B half-tile) - never a drain inside the loop; then one raw
// s_barrier per phase, which both publishes every wave's share of the awaited half (RAW: wait -> barrier ->
// ds_read) and proves the region about to be re-filled is no longer read (WAR: its readers ran two phases ago and
// s_waitcnt lgkmcnt(KEEP) in front of the barrier leaves only the LAST phase's reads in flight -> barrier -> DMA
// issue).  Both orders hold by construction, not by timing (MI355X guide: "Read a staged buffer one phase AFTER
// the wait that retires it").  Loads past the last K-tile are issued as dummies, so the counts stay exact.
//
// Configuration: 256 x 192, 8 waves (4 x 2, 64 x 96 per wave, two per SIMD): 8192 x 1536 outputs = 256 workgroups = one per
// CU.  LDS-DMA pieces either on every wave (A: a = 2 per half; B: b = 2 on 6 of the 8 waves) or, the default, on waves
// 0..3 only - one per SIMD - with a = 4, b = 3.  (Round 1 also built a 4-wave form, one wave per
// SIMD with the whole register file: slower, limited by VGPR <-> AGPR moves; pruned.  Round 2 tried an alternating-order
// schedule with single fragment register sets, to hold 64 x 128 accumulator blocks for asymmetric loader / multiplier
// roles and a 256 x 256 tile: the 256 x 192 form ran 34.9 vs 34.1 us, the 256 x 256 form spilled; removed.)
#include "codae_common.h"
#include <type_traits>

namespace codae {
namespace {

#include "gemm_bf16_halftile.h"

// DBG = 8 build only: per-workgroup wall-clock stamps (100 MHz s_memrealtime) at entry, first MFMA phase,
// end of the K loop, epilogue stores issued, stores retired; + the XCC the workgroup ran on
constexpr int TIMELINE_WGS = 4096, TIMELINE_SLOTS = 6;
__device__ unsigned long long g_timeline[TIMELINE_WGS * TIMELINE_SLOTS];

// EPI (bf16 output only): 1 = forward epilogue (bias + ReLU), 2 = data-gradient epilogue (ReLU mask from the saved
// activation + column sums = bias gradient), 3 = last forward layer of a training step with the MSE loss, its gradient
// and the metric sums computed straight from the accumulators (g.loss), 0 = everything decided at run time.  Apart from trimming the forward's
// epilogue this gives the forward and the data-gradient launches distinct kernel symbols in rocprofv3 traces.
// One output tile (or split-K range of one) of one GEMM: the whole kernel body, shared by the plain kernel (one GEMM per
// launch) and the grouped kernel (the weight gradients of every layer in one launch).  wg / nwg: this workgroup's index
// among the nwg workgroups of ITS GEMM; smem_raw: 2 * BUF (+ BM * 8 for EPI 3) bytes of LDS.
template <int BM, int BN, int WM, int WN, int NLB, int A_MODE, int B_MODE, bool C_F32, int DBG = 0, int EPI = 0>
__device__ __forceinline__ void gemm_bf16_pipe_tile(const GemmBf16& g, int tiles_n, int tiles_mn, int kt_total, int wg, int nwg,
                                                    char* smem_raw) {
    constexpr int NW = WM * WN;
    constexpr int SM = BM / WM, SN = BN / WN;        // wave sub-tile
    constexpr int AHR = BM / 2, BHR = BN / 2;        // rows / cols per half-tile
    constexpr int TMH = SM / 32, TNH = SN / 32;      // 16-wide MFMA tiles per wave half
    constexpr int AH = AHR * 128, BH = BHR * 128;    // bytes per half image
    constexpr int BUF = 2 * AH + 2 * BH;
    // LDS-DMA instructions per wave per half-tile; the B half-tile may be loaded by the first NLB
    // waves only (256 x 192 with 8 waves: 12 instructions = 6 waves x 2), the others issue none
    // (likewise the A half-tile by the first NLA waves when its pieces do not divide over all of them.  A 12-wave
    // form - 4 x 3 waves of 64 x 64, three per SIMD, 150 VGPRs - was measured: forward 36.2 vs 36.1 us for 8 waves,
    // wgrad 54.6 vs 50.7: occupancy is not what limits the K loop; it is not instantiated)
    // NLB == 4 with 8 waves: waves 0..3 - one per SIMD - carry ALL LDS-DMA pieces (A and B), their SIMD partners 4..7 none
    constexpr int NLA = (NLB == 4 && NW == 8) ? 4 : (((AHR / 8) % NW == 0) ? NW : 8);
    constexpr int NA = AHR / 8 / NLA, NB = BHR / 8 / NLB;
    static_assert(NLB <= NW && NLA <= NW && (AHR / 8) % NLA == 0 && (BHR / 8) % NLB == 0, "loader waves");
    static_assert(SM % 32 == 0 && SN % 32 == 0, "wave sub-tile must split into 16-wide half tiles");
    lds_char* smem = (lds_char*)smem_raw;

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = w / WN, wc = w % WN;

    // XCD-aware remap (workgroup b of a launch runs on XCD b % 8): consecutive tile ids - the tiles of one A row panel, then of
    // the next - go to ONE XCD's L2.  nwg <= 0: `wg` already is the tile id (the grouped launch remaps over ALL its GEMMs).
    int bid = wg;
    if (nwg > 0) {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int z = bid / tiles_mn;
    const int tmn = bid - z * tiles_mn;
    // Tile ids walk the output in panels of TILE_GROUP_M row tiles: down the panel's rows, then on to the next column.  Any 32
    // consecutive ids - what one XCD runs at a time - are then a 4 x 8 block of tiles sharing 4 A strips and 8 B strips in that
    // XCD's L2, whatever the width.  (Row-major ids gave 1 x 32 at C5's 32 column tiles: every workgroup of an XCD streamed a
    // different strip of the 75 MB weight matrix per row panel.  C3's 8 column tiles: the same 4 x 8 sets as before.)
    constexpr int TILE_GROUP_M = 4;
    const int tiles_m = tiles_mn / tiles_n;
    const int grp = tmn / (TILE_GROUP_M * tiles_n);
    const int tm0 = grp * TILE_GROUP_M;
    const int gsz = tiles_m - tm0 < TILE_GROUP_M ? tiles_m - tm0 : TILE_GROUP_M;
    const int within = tmn - grp * (TILE_GROUP_M * tiles_n);
    const int tn = within / gsz, tm = tm0 + (within - tn * gsz);
    const int i0 = tm * BM, j0 = tn * BN;
    const int kt_begin = (int)((int64_t)kt_total * z / g.split_k);
    const int kt_end = (int)((int64_t)kt_total * (z + 1) / g.split_k);
    const int nkt = kt_end - kt_begin;

    auto a_img = [&](int tile, int h) { return smem + (tile & 1) * BUF + h * AH; };
    auto b_img = [&](int tile, int h) { return smem + (tile & 1) * BUF + 2 * AH + h * BH; };
    // timing-only ablations (compile-time; DBG = 0 in the shipped instantiations)
    constexpr bool dbg_noload = DBG & 1, dbg_nomma = DBG & 2, dbg_nostore = DBG & 4, dbg_time = DBG & 8;
    constexpr bool dbg_no_a = DBG & 16, dbg_no_b = DBG & 32;        // only one operand's LDS-DMA
    auto stamp = [&](int slot) {
        if constexpr (dbg_time) {
            if (threadIdx.x == 0 && blockIdx.x < TIMELINE_WGS) g_timeline[blockIdx.x * TIMELINE_SLOTS + slot] = wall_clock64();
        }
    };
    stamp(0);
    if constexpr (dbg_time) {
        if (threadIdx.x == 0 && blockIdx.x < TIMELINE_WGS) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_timeline[blockIdx.x * TIMELINE_SLOTS + 5] = xcc & 0xf;
        }
    }
    if constexpr (EPI == 3) {
        // {dataset row or -1, mask id} per tile row, fetched once here (two dependent loads that would otherwise sit in
        // front of the epilogue); published to the other waves by the K loop's barriers
        const LossFuse& L = g.loss;
        int* rowinfo = reinterpret_cast<int*>(smem_raw + 2 * BUF);
        for (int r = threadIdx.x; r < BM; r += 64 * NW) {
            const int i = i0 + r;
            int src = -1, id = 0;
            if (i < L.B && i < g.M) {
                src = L.row_idx ? L.row_idx[i] : i;
                if (L.mask_id != nullptr) id = L.mask_id[i];
                else if (L.mask_to_use != nullptr) id = L.mask_to_use[(int64_t)src * L.nb_run + L.run];
            }
            rowinfo[2 * r] = src; rowinfo[2 * r + 1] = id;
        }
    }

    f32x4 acc[2][TMH][2][TNH];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TMH; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < TNH; ++d) acc[a][b][c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // The whole K loop, specialised at COMPILE time on the wave's loader roles (a wave-uniform property: with the
    // 256 x 192 tile only 6 of the 8 waves carry B pieces).  Round 1 tested the roles inside every phase: 2-3 scalar
    // branches per phase, each ending a basic block, so the B-piece issue of P2 / P4 sat in a block of its own in
    // front of the phase's MFMAs instead of between them.  Now there is one branch in front of the loop.
    auto k_loop = [&](auto al_tag, auto bl_tag) {
    constexpr bool a_loader = decltype(al_tag)::value, b_loader = decltype(bl_tag)::value;
    // per-lane LDS-DMA source pointers at k = 0 (loop invariant) and the per-K-tile advance
    uint32_t a_src[2][NA], b_src[2][NB];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int it = 0; it < NA; ++it) a_src[h][it] = half_src<A_MODE, AHR, SM, NLA>(g.lda, i0, g.M, h, it, a_loader ? w : 0, lane);
#pragma unroll
        for (int it = 0; it < NB; ++it) b_src[h][it] = half_src<B_MODE, BHR, SN, NLB>(g.ldb, j0, g.N, h, it, b_loader ? w : 0, lane);
    }
    const int64_t a_step = (A_MODE == OP_KC ? BK : (int64_t)BK * g.lda) * 2;       // bytes per K-tile
    const int64_t b_step = (B_MODE == OP_KC ? BK : (int64_t)BK * g.ldb) * 2;
    const char* a_base = reinterpret_cast<const char*>(g.A) + kt_begin * a_step;
    const char* b_base = reinterpret_cast<const char*>(g.B) + kt_begin * b_step;
    // Loads past the last K-tile are issued anyway, re-reading the last tile into the (dead) region
    // the schedule assigns: every phase stays one straight-line block the scheduler can interleave,
    // and the vmcnt arithmetic is exact to the end (cost: ~2 extra K-tiles of L2-hit DMA per workgroup).
    auto issue_a = [&](int tile, int h) {
        if constexpr (!dbg_noload && !dbg_no_a && a_loader) {
            const char* base = a_base + (int64_t)(tile < nkt ? tile : nkt - 1) * a_step;      // wave-uniform
            lds_char* img = a_img(tile, h);
#pragma unroll
            for (int it = 0; it < NA; ++it) glds16(base + a_src[h][it], img + (it * NLA + w) * 1024);
        }
    };
    auto issue_b = [&](int tile, int h) {
        if constexpr (!dbg_noload && !dbg_no_b && b_loader) {
            const char* base = b_base + (int64_t)(tile < nkt ? tile : nkt - 1) * b_step;
            lds_char* img = b_img(tile, h);
#pragma unroll
            for (int it = 0; it < NB; ++it) glds16(base + b_src[h][it], img + (it * NLB + w) * 1024);
        }
    };

    bf16x8 a0x[TMH][2], a0y[TMH][2], a1[TMH][2], b0[TNH][2], b1[TNH][2];

    auto read_a = [&](bf16x8 (&dst)[TMH][2], int tile, int h) {
#pragma unroll
        for (int mt = 0; mt < TMH; ++mt) dst[mt][0] = read_frag<A_MODE, AHR, 0>(a_img(tile, h), wr * TMH + mt, lane);
#pragma unroll
        for (int mt = 0; mt < TMH; ++mt) dst[mt][1] = read_frag<A_MODE, AHR, 1>(a_img(tile, h), wr * TMH + mt, lane);
    };
    auto read_b = [&](bf16x8 (&dst)[TNH][2], int tile, int h) {
#pragma unroll
        for (int nt = 0; nt < TNH; ++nt) dst[nt][0] = read_frag<B_MODE, BHR, 0>(b_img(tile, h), wc * TNH + nt, lane);
#pragma unroll
        for (int nt = 0; nt < TNH; ++nt) dst[nt][1] = read_frag<B_MODE, BHR, 1>(b_img(tile, h), wc * TNH + nt, lane);
    };
    auto mma = [&](const bf16x8 (&a)[TMH][2], const bf16x8 (&b)[TNH][2], int mh, int nh) {
        if constexpr (dbg_nomma) {
            // keep the fragments alive so the ds_reads stay
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int mt = 0; mt < TMH; ++mt) asm volatile("" ::"v"(a[mt][s]));
#pragma unroll
                for (int nt = 0; nt < TNH; ++nt) asm volatile("" ::"v"(b[nt][s]));
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int mt = 0; mt < TMH; ++mt)
#pragma unroll
                for (int nt = 0; nt < TNH; ++nt)
                    acc[mh][mt][nh][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nt][s], a[mt][s], acc[mh][mt][nh][nt], 0, 0, 0);
    };
    // LDS-DMA runs 6 phases ahead: the half-tile a phase reads has landed once at most the 5 half-tiles issued after
    // it are in flight - 2 A + 3 B in the phases that read an A half (P1, P3), 3 A + 2 B in those that read a B half
    // own LDS-DMA pieces that may still be in flight: CA A-halves + CB B-halves (per loader class of this wave)
    auto wait_dma = [&](auto ca, auto cb) {
        constexpr int CA = decltype(ca)::value, CB = decltype(cb)::value;
        constexpr int N = (a_loader ? CA * NA : 0) + (b_loader ? CB * NB : 0);
        if constexpr (a_loader || b_loader) wait_vmcnt<N>();
    };
    auto wait_for_a = [&]() { wait_dma(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{}); };
    auto wait_for_b = [&]() { wait_dma(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{}); };

    // prologue: both K-tiles of the LDS ring are requested, then A0(0) and B0(0) are read
    issue_a(0, 0); issue_b(0, 0); issue_a(0, 1); issue_b(0, 1);
    issue_a(1, 0); issue_b(1, 0); issue_a(1, 1); issue_b(1, 1);
    wait_dma(std::integral_constant<int, 3>{}, std::integral_constant<int, 4>{});      // A0(0) landed
    phase_barrier();
    read_a(a0x, 0, 0);
    wait_dma(std::integral_constant<int, 3>{}, std::integral_constant<int, 3>{});      // B0(0) landed
    phase_barrier();
    read_b(b0, 0, 0);

    stamp(1);
    // reads that may stay in flight across the barrier after a phase that read an A / a B half: the compiler-visible
    // ds_read_b128 of a k-contiguous operand (it waits for them itself before their first use); the asm-issued
    // transposed reads of a k-strided operand are waited for in full (settle() follows the barrier)
    constexpr int KEEP_A = A_MODE == OP_KC ? 2 * TMH : 0, KEEP_B = B_MODE == OP_KC ? 2 * TNH : 0;
    // one K-tile; `a0` holds A0(t), `a0n` receives A0(t+1).  Reads of a tile past the end fetch the
    // dummy re-load and are never multiplied (see the settle block behind the loop).
    // Issue order inside a phase (k-contiguous operands): an MFMA right behind the barrier, then the fragment reads of the NEXT phase
    // one per MFMA, then this wave's LDS-DMA pieces one per MFMA, then the rest of the 12 MFMAs.  Measured against the
    // compiler's own order with the phases pinned (phase_barrier), whole C3 step on one box, three alternating runs each:
    // 1.190 -> 1.182 ms (forward 35.4 -> 34.9 us, data gradient 38.3 -> 37.7, fused loss 54.7 -> 52.4); DMA pieces first: 1.197;
    // reads and DMA pieces paired behind the first six MFMAs: 1.191; all reads at once behind the first MFMA: 1.206; three more
    // interleavings (reads then DMA back to back; read / DMA alternating over all twelve; two MFMAs up front): within noise of this one.
    // (The k-strided forms' phases laid out by hand with sched_barriers around the asm-issued transposed reads - reads first; two
    // MFMAs, reads, three MFMAs, DMA, rest; two MFMAs, DMA + reads, rest -: grouped weight gradients 33.1-33.6 us against 32.8-33.1
    // for the compiler's own interleaving: left alone.)
    auto phase_order = [&](auto nread_tag, auto ndma_tag) {
        if constexpr (A_MODE == OP_KC && B_MODE == OP_KC && (DBG & 64) == 0) {
            constexpr int NREAD = decltype(nread_tag)::value, NDMA = decltype(ndma_tag)::value;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // one MFMA
                if (i < NREAD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // one LDS read
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < NDMA) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);         // one LDS-DMA piece (a VMEM load)
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
    };
    constexpr bool PINP = (A_MODE == OP_KC && B_MODE == OP_KC) && (DBG & 64) == 0;      // see phase_barrier; DBG 64: GemmBf16::coscheduled
    using RA = std::integral_constant<int, 2 * TMH>;
    using RB = std::integral_constant<int, 2 * TNH>;
    using DA = std::integral_constant<int, a_loader ? NA : 0>;
    using DB = std::integral_constant<int, b_loader ? NB : 0>;
    auto ktile = [&](int t, bf16x8 (&a0)[TMH][2], bf16x8 (&a0n)[TMH][2]) {
        // P1: A0 x B0
        wait_for_a();
        phase_barrier<KEEP_B, PINP>();      // (the previous phase, P4, read a B half)
        if constexpr (A_MODE == OP_KS) settle(a0);
        if constexpr (B_MODE == OP_KS) settle(b0);
        issue_a(t + 2, 0);
        read_a(a1, t, 1);
        mma(a0, b0, 0, 0);
        phase_order(RA{}, DA{});
        // P2: A1 x B0
        wait_for_b();
        phase_barrier<KEEP_A, PINP>();
        if constexpr (A_MODE == OP_KS) settle(a1);
        issue_b(t + 2, 0);
        read_b(b1, t, 1);
        mma(a1, b0, 1, 0);
        phase_order(RB{}, DB{});
        // P3: A1 x B1
        wait_for_a();
        phase_barrier<KEEP_B, PINP>();
        if constexpr (B_MODE == OP_KS) settle(b1);
        issue_a(t + 2, 1);
        read_a(a0n, t + 1, 0);
        mma(a1, b1, 1, 1);
        phase_order(RA{}, DA{});
        // P4: A0 x B1
        wait_for_b();
        phase_barrier<KEEP_A, PINP>();
        issue_b(t + 2, 1);
        read_b(b0, t + 1, 0);
        mma(a0, b1, 0, 1);
        phase_order(RB{}, DB{});
    };
    int t = 0;
    for (; t + 1 < nkt; t += 2) {
        ktile(t, a0x, a0y);
        ktile(t + 1, a0y, a0x);
    }
    if (t < nkt) ktile(t, a0x, a0y);
    if constexpr (A_MODE == OP_KS || B_MODE == OP_KS) {
        // The last K-tile's reads of tile t+1 are dummies nobody multiplies.  For a k-strided operand they are
        // inline-asm reads, so without a use the compiler would recycle their destination registers at once (round 1's
        // code object did: as address temporaries, while the LDS data was still on its way - a write-after-write race
        // the s_waitcnt bookkeeping cannot see; tools/check_isa.py found it).  Waiting for them and naming the
        // registers here keeps them reserved until the data has landed.  (Peeling the last K-tile instead, or skipping
        // the reads under a branch, made the compiler copy fragment registers across the control-flow merge - moves of
        // registers with reads in flight, the same race - and spill.)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (A_MODE == OP_KS) { settle(a0x); settle(a0y); }      // (a1 / b1 are always multiplied)
        if constexpr (B_MODE == OP_KS) settle(b0);
    }
    };   // k_loop
    {
        const bool al = (NLA == NW) || (w < NLA), bl = (NLB == NW) || (w < NLB);
        if constexpr (NLA == NW && NLB == NW) k_loop(std::true_type{}, std::true_type{});
        else if constexpr (NLA == NW) { if (bl) k_loop(std::true_type{}, std::true_type{}); else k_loop(std::true_type{}, std::false_type{}); }
        else {
            if (al && bl) k_loop(std::true_type{}, std::true_type{});
            else if (al) k_loop(std::true_type{}, std::false_type{});
            else if (bl) k_loop(std::false_type{}, std::true_type{});
            else k_loop(std::false_type{}, std::false_type{});
        }
    }
    wait_vmcnt<0>();        // the trailing dummy loads must not outlive the kernel's use of LDS
    stamp(2);

    // ---- epilogue: lane holds C[i][j .. j+3] of each 16 x 16 tile (swapped MFMA operands)
    const int li = lane & 15, g4 = (lane >> 4) * 4;
    // the next launch's weights: this workgroup's share of the matrix, one 4-B load per 128-B line, issued now and consumed
    // (by a store that never happens) after the epilogue's own stores - its whole latency lies under the epilogue
    uint32_t pf_val = 0;
    if (g.prefetch != nullptr) {
        const int64_t lines = (g.prefetch_bytes + 127) >> 7;
        const int64_t per_wg = (lines + nwg - 1) / nwg;
        for (int64_t i = threadIdx.x; i < per_wg; i += 64 * NW) {
            const int64_t mine = (int64_t)wg * per_wg + i;
            if (mine < lines) pf_val ^= *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(g.prefetch) + (mine << 7));
        }
    }
    // (consumed at the kernel's exits by an empty asm that names the register: the compiler keeps the loads and waits for
    // them only there, after the epilogue's own stores have been issued)
    auto consume_prefetch = [&]() { asm volatile("" ::"v"(pf_val)); };
    phase_barrier();          // every wave is past its last fragment read and every DMA has landed: LDS is free
    if constexpr (EPI == 3) {
        // Last forward layer of a training step (train_dae_on_embedding.py:206-223): y = acc + bias stays in registers.
        // Each lane holds C[i][j .. j+3] of its 16 x 16 tiles, so it fetches the 16 B of the target row x (gathered
        // dataset row, fp32) and the 4 mask bytes that face them - unconditional loads from clamped addresses, one row
        // half of the wave at a time - computes dy = 2 (y - x) / n, the metric sums and the column sums of dy, and only
        // the bf16 dy tile goes through LDS (one pass, whole rows out, 16 B per lane).  Round 1 staged the fp32 y tile
        // through LDS in two halves and re-read it row by row: 19 us of LDS / VALU work on top of the HBM time.
        const LossFuse& L = g.loss;
        const int* rowinfo = reinterpret_cast<const int*>(smem_raw + 2 * BUF);
        const bool masked = (L.mask_id != nullptr) || (L.mask_to_use != nullptr);
        const uint8_t* tb = masked ? L.table : reinterpret_cast<const uint8_t*>(L.data);     // (unmasked: any valid bytes)
        constexpr int NT = 64 * NW;
        constexpr int PITCH = BN * 2 + 16;
        static_assert(BM * PITCH <= 2 * BUF, "output tile must fit in the staging buffers");
        float sq = 0.f, sqp = 0.f;
        float cs[2][TNH][4];
        int jcl[2][TNH];                // clamped global column of the lane's 4-column group; -1 flag kept in jok
        bool jok[2][TNH];
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int nt = 0; nt < TNH; ++nt) {
                const int j = j0 + nh * BHR + wc * (SN / 2) + 16 * nt + g4;
                jok[nh][nt] = j < g.N;
                jcl[nh][nt] = jok[nh][nt] ? j : 0;
                const float4 bj = load_bias4(g.bias, g.A, j, g.N);          // y = acc + bias, in place (frees 24 registers)
#pragma unroll
                for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                    for (int mt = 0; mt < TMH; ++mt) {
                        acc[mh][mt][nh][nt][0] += bj.x; acc[mh][mt][nh][nt][1] += bj.y;
                        acc[mh][mt][nh][nt][2] += bj.z; acc[mh][mt][nh][nt][3] += bj.w;
                    }
#pragma unroll
                for (int k = 0; k < 4; ++k) cs[nh][nt][k] = 0.f;
            }
        // a tile that lies wholly inside the N columns (the usual case) addresses the lane's 6 column groups as ONE
        // pointer + compile-time offsets (instruction immediates); per-group clamped addresses cost 48 more registers
        const int jbase = j0 + wc * (SN / 2) + g4;
        auto loss_rows = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                float4 xv[TMH][2][TNH];
                uint32_t mk[TMH][2][TNH];
                bool live[TMH];
#pragma unroll
                for (int mt = 0; mt < TMH; ++mt) {
                    const int il = mh * AHR + wr * (SM / 2) + 16 * mt + li;
                    const int src = rowinfo[2 * il], id = rowinfo[2 * il + 1];
                    live[mt] = src >= 0;
                    const float* xrow = L.data + (int64_t)(src >= 0 ? src : 0) * L.io;
                    const uint8_t* mrow = tb + (int64_t)id * L.io;
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                        for (int nt = 0; nt < TNH; ++nt) {
                            const int off = FULL ? jbase + (nh * BHR + 16 * nt) : jcl[nh][nt];
                            xv[mt][nh][nt] = *reinterpret_cast<const float4*>(xrow + off);
                            const uint32_t mv = *reinterpret_cast<const uint32_t*>(mrow + off);
                            mk[mt][nh][nt] = masked ? mv : 0x01010101u;
                        }
                }
#pragma unroll
                for (int mt = 0; mt < TMH; ++mt) {
                    const int il = mh * AHR + wr * (SM / 2) + 16 * mt + li;
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                        for (int nt = 0; nt < TNH; ++nt) {
                            const int jl = nh * BHR + wc * (SN / 2) + 16 * nt + g4;
                            const f32x4 a = acc[mh][mt][nh][nt];
                            const float4 x = xv[mt][nh][nt];
                            const float yv[4] = {a[0], a[1], a[2], a[3]};
                            const float xs[4] = {x.x, x.y, x.z, x.w};
                            const bool on = live[mt] && (FULL || jok[nh][nt]);
                            float gq[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float d = on ? xs[k] - yv[k] : 0.f;
                                const float se = d * d;
                                sq += se;
                                asm volatile("" : "+v"(sq));     // scalar chain, never packed with op_sel half swaps: gemm_bf16.hip, DESIGN.md 5d
                                sqp += ((mk[mt][nh][nt] >> (8 * k)) & 0xffu) == 0 ? se : 0.f;
                                gq[k] = on ? -2.f * d * L.inv_n : 0.f;       // (+0 in pad rows / columns)
                                cs[nh][nt][k] += gq[k];
                            }
                            u32x2 o;
                            o[0] = pack_bf16x2(gq[0], gq[1]);
                            o[1] = pack_bf16x2(gq[2], gq[3]);
                            *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(smem + il * PITCH + jl * 2) = o;
                        }
                }
                // (keeps the second half's gather registers from being live beside the first half's: 256-register budget)
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (j0 + BN <= g.N) loss_rows(std::true_type{}); else loss_rows(std::false_type{});
        stamp(3);                 // (DBG 8: loss arithmetic done, dy tile staged)
        phase_barrier();
        {
            constexpr int CH = BN / 8, RL = NT / CH;
            const int c = threadIdx.x % CH, rl = threadIdx.x / CH;
            const int j = j0 + c * 8;
            bf16_t* Cb = reinterpret_cast<bf16_t*>(g.C);
            constexpr int ITER = (BM + RL - 1) / RL;          // fully unrolled: every LDS read is issued before the first store
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int r = rl + it * RL;
                const bool ok = rl < RL && j < g.N && r < BM && i0 + r < g.M;
                const u32x4 lv = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(smem + (ok ? r : 0) * PITCH + c * 16);
                if (ok) *reinterpret_cast<uint4*>(Cb + (int64_t)(i0 + r) * g.ldc + j) = make_uint4(lv[0], lv[1], lv[2], lv[3]);
            }
        }
        if constexpr (dbg_time) { wait_vmcnt<0>(); stamp(4); }
        // metric sums: lanes -> wave -> workgroup -> this workgroup's row of L.parts
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) { sq += __shfl_xor(sq, o2); sqp += __shfl_xor(sqp, o2); }
        // column sums of dy (the last bias gradient): the 16 row-lanes of a column group, then the WM wave rows
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int nt = 0; nt < TNH; ++nt)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v = cs[nh][nt][k];
                    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                    cs[nh][nt][k] = v;
                }
        phase_barrier();          // the staged dy tile has been read by everyone: LDS is scratch again
        float* red = reinterpret_cast<float*>(smem_raw);          // [WM][BN] column partials, then 2 * NW block partials
        static_assert((WM * BN + 2 * NW) * 4 <= 2 * BUF, "reduction scratch must fit");
        if (li == 0) {
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int nt = 0; nt < TNH; ++nt) {
                    const int jl = nh * BHR + wc * (SN / 2) + 16 * nt + g4;
#pragma unroll
                    for (int k = 0; k < 4; ++k) red[wr * BN + jl + k] = cs[nh][nt][k];
                }
        }
        float* wsum = red + WM * BN;
        if (lane == 0) { wsum[2 * w] = sq; wsum[2 * w + 1] = sqp; }
        phase_barrier();
        if (threadIdx.x == 0) {
            float a = 0.f, b2 = 0.f;
            for (int ww = 0; ww < NW; ++ww) { a += wsum[2 * ww]; b2 += wsum[2 * ww + 1]; }
            L.parts[2 * bid] = (double)a;                       // (bid: tile index after the XCD remap, < gridDim.x)
            L.parts[2 * bid + 1] = masked ? (double)b2 : 0.0;
        }
        if (g.colsum_part != nullptr) {
            for (int col = threadIdx.x; col < BN; col += NT) {
                float sum = 0.f;
#pragma unroll
                for (int r = 0; r < WM; ++r) sum += red[r * BN + col];
                if (j0 + col < g.N) g.colsum_part[(int64_t)tm * g.N + j0 + col] = sum;
            }
        }
        consume_prefetch();
        return;
    }
    if constexpr (!C_F32 && !dbg_nostore) {
        // bf16 output through LDS, whole rows, 16 B per lane (see gemm_bf16.hip); mask + bias-grad sums ride along
        constexpr int NT = 64 * NW;
        constexpr int PITCH = BN * 2 + 16;
        static_assert(BM * PITCH <= 2 * BUF, "output tile must fit in the staging buffers");
        const float floor_v = g.relu ? 0.f : -__builtin_inff();
        constexpr int CH = BN / 8, RL = NT / CH;
        constexpr int ITER = (BM + RL - 1) / RL;
        const int c = threadIdx.x % CH, rl = threadIdx.x / CH;
        const int j = j0 + c * 8;
        const int jc = j < g.N ? j : 0;
        // ReLU mask of the data gradient: one BIT per element when the forward launch left them (g.relu_bits: 1 byte per lane
        // and row instead of 16 - the saved activation is 25 MB per launch at C3 and comes from HBM), else the activation.
        // The byte loads depend on nothing but indices: issued HERE, in front of the staging pass, their latency lies under it
        // (behind the barrier that ends the staging they were the first thing the write-out waited for).
        const bool bit_mask = EPI != 1 && g.relu_bits != nullptr;
        uint8_t hb[ITER];
        if constexpr (EPI != 1) {
            const uint8_t* bsrc = bit_mask ? g.relu_bits : reinterpret_cast<const uint8_t*>(g.A);
            const int64_t bld = bit_mask ? g.ld_bits : 0;
            if (bit_mask) {
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    const int r = rl + it * RL;
                    const bool ok = rl < RL && j < g.N && r < BM && i0 + r < g.M;
                    hb[it] = bsrc[(int64_t)(ok ? i0 + r : 0) * bld + (ok ? (jc >> 3) : 0)];
                }
            }
        }
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int nt = 0; nt < TNH; ++nt) {
                const int jl = nh * BHR + wc * (SN / 2) + 16 * nt + g4;
                const float4 bj = load_bias4(g.bias, g.A, j0 + jl, g.N);
#pragma unroll
                for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                    for (int mt = 0; mt < TMH; ++mt) {
                        const int il = mh * AHR + wr * (SM / 2) + 16 * mt + li;
                        const f32x4 a = acc[mh][mt][nh][nt];
                        const float v0 = clamp_below(a[0] + bj.x, floor_v), v1 = clamp_below(a[1] + bj.y, floor_v);
                        const float v2 = clamp_below(a[2] + bj.z, floor_v), v3 = clamp_below(a[3] + bj.w, floor_v);
                        u32x2 o;
                        o[0] = pack_bf16x2(v0, v1);
                        o[1] = pack_bf16x2(v2, v3);
                        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(smem + il * PITCH + jl * 2) = o;
                    }
            }
        __syncthreads();
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        {
            // Fully unrolled with predicated stores and UNCONDITIONAL loads from clamped addresses: the ReLU-mask loads
            // (the saved activation: 25 MB per launch, from HBM) of all of a thread's 12-13 rows are in flight together.
            // As a loop with one row per trip every trip waited for its own load: the data-gradient launch was 10 us
            // longer than the forward one.
            bf16_t* Cb = reinterpret_cast<bf16_t*>(g.C);
            const bool relu_mask = EPI != 1 && !bit_mask && g.relu_src != nullptr;
            const bf16_t* hsrc = relu_mask ? g.relu_src : g.A;                 // (no mask: any valid 16-B aligned bytes)
            const int64_t hld = relu_mask ? g.ld_relu : 0;
            uint4 hv[ITER];
            if constexpr (EPI != 1) {
                if (!bit_mask) {
#pragma unroll
                    for (int it = 0; it < ITER; ++it) {
                        const int r = rl + it * RL;
                        const bool ok = rl < RL && j < g.N && r < BM && i0 + r < g.M;
                        hv[it] = *reinterpret_cast<const uint4*>(hsrc + (int64_t)(ok ? i0 + r : 0) * hld + (relu_mask ? jc : 0));
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int r = rl + it * RL;
                const bool ok = rl < RL && j < g.N && r < BM && i0 + r < g.M;
                const u32x4 lv = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(smem + (ok ? r : 0) * PITCH + c * 16);
                uint4 v = make_uint4(lv[0], lv[1], lv[2], lv[3]);
                if constexpr (EPI != 1) {
                    if (bit_mask) {
                        const uint32_t hbv = hb[it];
                        auto keepb = [](uint32_t val, uint32_t b2) -> uint32_t {        // b2: the element pair's two mask bits
                            return val & (((b2 & 1u) ? 0x0000ffffu : 0u) | ((b2 & 2u) ? 0xffff0000u : 0u));
                        };
                        v.x = keepb(v.x, hbv); v.y = keepb(v.y, hbv >> 2); v.z = keepb(v.z, hbv >> 4); v.w = keepb(v.w, hbv >> 6);
                    } else if (relu_mask) {
                        const uint4 h = hv[it];
                        auto keep = [](uint32_t val, uint32_t hh) -> uint32_t {
                            const uint32_t lo = ((hh & 0x8000u) == 0 && (hh & 0x7fffu) != 0) ? 0x0000ffffu : 0u;
                            const uint32_t hi = ((hh & 0x80000000u) == 0 && (hh & 0x7fff0000u) != 0) ? 0xffff0000u : 0u;
                            return val & (lo | hi);
                        };
                        v.x = keep(v.x, h.x); v.y = keep(v.y, h.y); v.z = keep(v.z, h.z); v.w = keep(v.w, h.w);
                    }
                }
                if constexpr (EPI == 1) {
                    if (g.relu_bits_out != nullptr) {
                        // 1 bit per stored element: > 0 for a bf16 = sign clear and not zero - what the data gradient's `keep` tests on
                        // the activation.  A lane's 8 elements are one byte; the 4 lanes of a quad hold 4 consecutive bytes of one
                        // row (CH and the wave size are multiples of 4), gathered by DPP quad rotations into ONE dword store by the
                        // quad's first lane (byte stores, 24 per row: + 1.6 us per launch; dwords: see DESIGN.md 5e).
                        // (a ReLU output is never negative, so "> 0" = "magnitude not zero": adding 0x7fff to a 15-bit magnitude carries
                        //  into bit 15 exactly when it is not zero - both halves of the pair at once, no compares)
                        auto pos = [](uint32_t w2) -> uint32_t {
                            const uint32_t t = (w2 & 0x7fff7fffu) + 0x7fff7fffu;
                            return ((t >> 15) & 1u) | ((t >> 30) & 2u);
                        };
                        const uint32_t b0 = ok ? (pos(v.x) | (pos(v.y) << 2) | (pos(v.z) << 4) | (pos(v.w) << 6)) : 0u;
                        const uint32_t b1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b0, 0x39, 0xf, 0xf, true);      // quad_perm [1,2,3,0]
                        const uint32_t b2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b0, 0x4e, 0xf, 0xf, true);      // [2,3,0,1]
                        const uint32_t b3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b0, 0x93, 0xf, 0xf, true);      // [3,0,1,2]
                        if (ok && (c & 3) == 0)
                            *reinterpret_cast<uint32_t*>(g.relu_bits_out + (int64_t)(i0 + r) * g.ld_bits + (j >> 3)) = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
                    }
                }
                if (ok) {
                    *reinterpret_cast<uint4*>(Cb + (int64_t)(i0 + r) * g.ldc + j) = v;
                    if (EPI != 1 && g.colsum_part != nullptr) {
                        cs[0] += bf16_to_f32((bf16_t)(v.x & 0xffff)); cs[1] += bf16_to_f32((bf16_t)(v.x >> 16));
                        cs[2] += bf16_to_f32((bf16_t)(v.y & 0xffff)); cs[3] += bf16_to_f32((bf16_t)(v.y >> 16));
                        cs[4] += bf16_to_f32((bf16_t)(v.z & 0xffff)); cs[5] += bf16_to_f32((bf16_t)(v.z >> 16));
                        cs[6] += bf16_to_f32((bf16_t)(v.w & 0xffff)); cs[7] += bf16_to_f32((bf16_t)(v.w >> 16));
                    }
                }
            }
        }
        stamp(3);
        if constexpr (dbg_time) { wait_vmcnt<0>(); stamp(4); }
        if (EPI != 1 && g.colsum_part != nullptr) {
            __syncthreads();
            float* red = reinterpret_cast<float*>(smem_raw);
            static_assert(RL * BN * 4 <= 2 * BUF, "reduction scratch must fit");
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
        consume_prefetch();
        return;
    }
    // fp32 output (split-K slabs of the weight gradient, fp32 y): staged through LDS like the bf16 path, in the two
    // row halves of the phase structure (the fp32 tile is twice the staging space); whole 768-B row segments,
    // 16 B per lane.  (Straight from the accumulators a wave-store wrote 64-B pieces: the slab launch moved 60 MB
    // for 47 MB of slabs.)
    if constexpr (C_F32) {
        constexpr int NT = 64 * NW;
        constexpr int HR = AHR;
        constexpr int PITCH = BN * 4 + 16;
        static_assert(HR * PITCH <= 2 * BUF, "fp32 half tile must fit in the staging buffers");
        constexpr int CH = BN / 4, RL = NT / CH;          // float4 chunks per row, rows per pass
        float* Cf = reinterpret_cast<float*>(g.C);
        if (g.split_k > 1) Cf += (int64_t)z * g.M * g.ldc;
        const float floor_v = g.relu ? 0.f : -__builtin_inff();
        const int c = threadIdx.x % CH, rl = threadIdx.x / CH;
        const int j = j0 + c * 4;
        float sq = 0.f;                                  // sum of the stored values' squares (g.sumsq_slots)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int nt = 0; nt < TNH; ++nt) {
                    const int jl = nh * BHR + wc * (SN / 2) + 16 * nt + g4;
                    const float4 bj = load_bias4(g.bias, g.A, j0 + jl, g.N);
#pragma unroll
                    for (int mt = 0; mt < TMH; ++mt) {
                        const int il = wr * (SM / 2) + 16 * mt + li;          // row inside this half
                        const f32x4 a = acc[hh][mt][nh][nt];
                        f32x4 v;
                        v[0] = clamp_below(a[0] + bj.x, floor_v); v[1] = clamp_below(a[1] + bj.y, floor_v);
                        v[2] = clamp_below(a[2] + bj.z, floor_v); v[3] = clamp_below(a[3] + bj.w, floor_v);
                        *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(smem + il * PITCH + jl * 4) = v;
                    }
                }
            __syncthreads();
            constexpr int ITER = (HR + RL - 1) / RL;
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int r = rl + it * RL;
                const bool ok = rl < RL && j < g.N && r < HR && i0 + hh * HR + r < g.M;
                const f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(smem + (ok ? r : 0) * PITCH + c * 16);
                if (ok) {
                    *reinterpret_cast<float4*>(Cf + (int64_t)(i0 + hh * HR + r) * g.ldc + j) = make_float4(v[0], v[1], v[2], v[3]);
                    sq += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                }
            }
            __syncthreads();
        }
        if (g.sumsq_slots != nullptr) {      // an unsplit weight gradient: clip_grad_norm_'s sum g^2 without a pass of its own
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
            float* red = reinterpret_cast<float*>(smem_raw);
            if (lane == 0) red[w] = sq;
            __syncthreads();
            if (threadIdx.x == 0) {
                float t = 0.f;
                for (int ww = 0; ww < NW; ++ww) t += red[ww];
                atomicAdd(g.sumsq_slots + (wg & (CODAE_S_N_SLOTS - 1)), (double)t);
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int NLB, int A_MODE, int B_MODE, bool C_F32, int DBG = 0, int EPI = 0>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN + 3) / 4)
void gemm_bf16_pipe_kernel(GemmBf16 g, int tiles_n, int tiles_mn, int kt_total) {
    // (EPI = 3: + {dataset row, mask id} of the tile's rows; ONE array: a second __shared__ object beside an LDS-DMA
    // staging array makes the compiler drain vmcnt in front of LDS reads)
    __shared__ __attribute__((aligned(16))) char smem_raw[2 * (BM + BN) * 128 + (EPI == 3 ? BM * 8 : 0)];
    gemm_bf16_pipe_tile<BM, BN, WM, WN, NLB, A_MODE, B_MODE, C_F32, DBG, EPI>(g, tiles_n, tiles_mn, kt_total, blockIdx.x, gridDim.x, smem_raw);
}

// Every layer's weight gradient dW_l = dA_l^T H_l (both operands k-strided, fp32 straight into the gradient vector, K = the whole
// batch: NO split-K slabs, no reduce pass, sum g^2 from the epilogue) in ONE launch of 256 x 192 tiles: workgroup -> (GEMM, tile)
// by the prefix sums of the descriptor block.  C3: 10 x 48 tiles of 128 K-tiles each on 256 CUs; the long K loop runs at the
// rate DESIGN.md section 5 measured for K = 24576 (fixed costs amortised), where the per-layer launches (K = 1638 per workgroup
// after a 5-way split) pay pipeline fill, epilogue and a 47 MB slab round trip per layer.
__global__ __launch_bounds__(512, 2) void gemm_bf16_pipe_grouped_kernel(GemmBf16Group grp) {
    constexpr int BM = 256, BN = 192;
    __shared__ __attribute__((aligned(16))) char smem_raw[2 * (BM + BN) * 128];
    // Workgroup b runs on XCD b % 8, the s-th of that XCD's workgroups (s = b / 8).  Each XCD takes a CONTIGUOUS eighth of the
    // launch's tile list (GEMM after GEMM, row panel after row panel), so that the 32 workgroups an XCD runs at a time are 4
    // consecutive row panels x all column tiles of ONE weight gradient: they share 4 dA strips and the layer's 8 H strips in that
    // XCD's L2.  (Remapped per GEMM, as a plain launch does, an XCD held 6 tiles of each of 5 layers at a time and every tile
    // streamed its own H strip from HBM: 2.25 GB read per launch for 0.5 GB of operands, the launch ran at the HBM roofline.)
    const int total = grp.wg_begin[grp.n];
    int t = (int)blockIdx.x;
    {
        const int q = total >> 3, r = total & 7, xcd = t & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
    }
    int j = 0;
    while (j + 1 < grp.n && t >= grp.wg_begin[j + 1]) ++j;
    const GemmBf16& g = grp.g[j];
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    gemm_bf16_pipe_tile<BM, BN, 4, 2, 4, OP_KS, OP_KS, true, 0, 0>(g, tiles_n, tiles_m * tiles_n, g.K / BK, t - grp.wg_begin[j], 0, smem_raw);
}

template <int BM, int BN, int WM, int WN, int NLB>
int launch_pipe(const GemmBf16& g, hipStream_t s) {
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const int kt_total = g.K / BK;
    const int64_t nwg = (int64_t)tiles_m * tiles_n * g.split_k;
    CODAE_REQUIRE(nwg < (1 << 30), "gemm_bf16: grid too large");
    // LDS-DMA sources are addressed as {scalar base, 32-bit lane offset}
    CODAE_REQUIRE((int64_t)(g.a_mode == OP_KC ? g.M : g.K) * g.lda * 2 < (int64_t)1 << 32 &&
                      (int64_t)(g.b_mode == OP_KC ? g.N : g.K) * g.ldb * 2 < (int64_t)1 << 32,
                  "gemm_bf16: operand larger than 4 GiB");
    dim3 grid((unsigned)nwg), block(64 * WM * WN);
#define LAUNCH(AM, BMODE, CF, EP) \
    hipLaunchKernelGGL((gemm_bf16_pipe_kernel<BM, BN, WM, WN, NLB, AM, BMODE, CF, 0, EP>), grid, block, 0, s, g, tiles_n, tiles_m * tiles_n, kt_total)
#define LAUNCH_BF16(AM, BMODE) do { if (bwd_epi) LAUNCH(AM, BMODE, false, 2); else LAUNCH(AM, BMODE, false, 1); } while (0)
    const bool bwd_epi = g.relu_src != nullptr || g.colsum_part != nullptr;
    // (the data gradient beside another stream's weight gradients: the same kernel with the compiler's schedule, GemmBf16::coscheduled)
    if (g.coscheduled && bwd_epi && !g.loss.enabled && g.a_mode == OP_KC && g.b_mode == OP_KC && !g.c_f32) {
        hipLaunchKernelGGL((gemm_bf16_pipe_kernel<BM, BN, WM, WN, NLB, OP_KC, OP_KC, false, 64, 2>), grid, block, 0, s, g, tiles_n, tiles_m * tiles_n, kt_total);
        CODAE_LAUNCH_CHECK();
        return CODAE_OK;
    }
    if (g.loss.enabled) {
        if constexpr (WM * WN == 8) LAUNCH(OP_KC, OP_KC, false, 3);
        else { set_error("gemm_bf16: fused loss is built for the 8-wave pipelined tile only"); return CODAE_E_UNSUPPORTED; }
    } else if (g.a_mode == OP_KC && g.b_mode == OP_KC) { if (g.c_f32) LAUNCH(OP_KC, OP_KC, true, 0); else LAUNCH_BF16(OP_KC, OP_KC); }
    else if (g.a_mode == OP_KC && g.b_mode == OP_KS) { if (g.c_f32) LAUNCH(OP_KC, OP_KS, true, 0); else LAUNCH_BF16(OP_KC, OP_KS); }
    else if (g.a_mode == OP_KS && g.b_mode == OP_KS) { if (g.c_f32) LAUNCH(OP_KS, OP_KS, true, 0); else LAUNCH_BF16(OP_KS, OP_KS); }
    else { set_error("gemm_bf16: operand mode combination not built"); return CODAE_E_UNSUPPORTED; }
#undef LAUNCH_BF16
#undef LAUNCH
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

}  // namespace

// timing-only ablation builds of the forward form (KC x KC, bf16 out), 256 x 192
template <int DBG>
int launch_dbg(const GemmBf16& g, hipStream_t s) {
    constexpr int BM = 256, BN = 192;
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_bf16_pipe_kernel<BM, BN, 4, 2, 4, OP_KC, OP_KC, false, DBG>), dim3(tiles_m * tiles_n), dim3(512), 0, s, g,
                       tiles_n, tiles_m * tiles_n, g.K / BK);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

// 128 x 192, 8 waves of 32 x 96 (forward / data-gradient form only: the k-strided half images need 96 or 128 columns):
// for launches between the 64 x 64 tiles and the 256 x 192 tile - 4096 x 1536 is 256 of these, one per CU
int launch_pipe_mid(const GemmBf16& g, hipStream_t s) {
    constexpr int BM = 128, BN = 192;
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    CODAE_REQUIRE(g.a_mode == OP_KC && g.b_mode == OP_KC && !g.c_f32 && !g.loss.enabled && g.split_k == 1, "gemm_bf16: 128 x 192 tile is forward-form only");
    CODAE_REQUIRE((int64_t)g.M * g.lda * 2 < (int64_t)1 << 32 && (int64_t)g.N * g.ldb * 2 < (int64_t)1 << 32, "gemm_bf16: operand larger than 4 GiB");
    dim3 grid((unsigned)(tiles_m * tiles_n)), block(512);
    if (g.relu_src != nullptr || g.colsum_part != nullptr)
        hipLaunchKernelGGL((gemm_bf16_pipe_kernel<BM, BN, 4, 2, 4, OP_KC, OP_KC, false, 0, 2>), grid, block, 0, s, g, tiles_n, tiles_m * tiles_n, g.K / BK);
    else
        hipLaunchKernelGGL((gemm_bf16_pipe_kernel<BM, BN, 4, 2, 4, OP_KC, OP_KC, false, 0, 1>), grid, block, 0, s, g, tiles_n, tiles_m * tiles_n, g.K / BK);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

// all weight gradients of a step in one launch (see gemm_bf16_pipe_grouped_kernel); fills grp.wg_begin
int gemm_bf16_pipe_grouped(GemmBf16Group& grp, hipStream_t s) {
    CODAE_REQUIRE(grp.n >= 1 && grp.n <= CODAE_GROUP_MAX, "gemm_bf16_pipe_grouped: %d GEMMs", grp.n);
    int total = 0;
    for (int j = 0; j < grp.n; ++j) {
        const GemmBf16& g = grp.g[j];
        CODAE_REQUIRE(g.a_mode == OP_KS && g.b_mode == OP_KS && g.c_f32 && g.split_k == 1 && g.K % BK == 0 && g.K >= BK && g.M % 8 == 0 &&
                          g.N % 8 == 0 && g.M >= 8 && g.N >= 8 && !g.loss.enabled && g.relu_src == nullptr && g.colsum_part == nullptr &&
                          g.bias == nullptr && !g.relu && (g.lda % 8) == 0 && (g.ldb % 8) == 0 && (g.ldc % 4) == 0,
                      "gemm_bf16_pipe_grouped: GEMM %d is not a plain unsplit weight-gradient form", j);
        CODAE_REQUIRE((int64_t)g.K * g.lda * 2 < (int64_t)1 << 32 && (int64_t)g.K * g.ldb * 2 < (int64_t)1 << 32,
                      "gemm_bf16_pipe_grouped: operand larger than 4 GiB");
        CODAE_REQUIRE((reinterpret_cast<uintptr_t>(g.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(g.B) & 15) == 0 &&
                          (reinterpret_cast<uintptr_t>(g.C) & 15) == 0, "gemm_bf16_pipe_grouped: operands must be 16-byte aligned");
        grp.wg_begin[j] = total;
        total += ((g.M + 255) / 256) * ((g.N + 191) / 192);
    }
    grp.wg_begin[grp.n] = total;
    hipLaunchKernelGGL(gemm_bf16_pipe_grouped_kernel, dim3(total), dim3(512), 0, s, grp);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

// cfg 0: 256 x 192 with 4 waves; cfg 1: 256 x 192 with 8 waves
int gemm_bf16_pipe(const GemmBf16& g, int cfg, hipStream_t s) {
    if (cfg == 7) return launch_pipe_mid(g, s);
    if (g.dbg == 8 && g.loss.enabled) {          // stamped build of the fused-loss kernel (tools/timeline_loss.py)
        const int tiles_m = (g.M + 255) / 256, tiles_n = (g.N + 191) / 192;
        hipLaunchKernelGGL((gemm_bf16_pipe_kernel<256, 192, 4, 2, 6, OP_KC, OP_KC, false, 8, 3>), dim3(tiles_m * tiles_n), dim3(512), 0, s, g,
                           tiles_n, tiles_m * tiles_n, g.K / BK);
        CODAE_LAUNCH_CHECK();
        return CODAE_OK;
    }
    if (g.dbg && !g.loss.enabled && g.a_mode == OP_KC && g.b_mode == OP_KC && !g.c_f32 && g.split_k == 1) {
        switch (g.dbg) {
            case 1: return launch_dbg<1>(g, s);
            case 2: return launch_dbg<2>(g, s);
            case 3: return launch_dbg<3>(g, s);
            case 4: return launch_dbg<4>(g, s);
            case 5: return launch_dbg<5>(g, s);
            case 6: return launch_dbg<6>(g, s);
            case 7: return launch_dbg<7>(g, s);
            case 8: return launch_dbg<8>(g, s);
            case 9: return launch_dbg<9>(g, s);      // stamps + no LDS-DMA
            case 10: return launch_dbg<10>(g, s);    // stamps + no MFMA
            case 16: return launch_dbg<16>(g, s);    // no LDS-DMA of the A operand
            case 32: return launch_dbg<32>(g, s);    // no LDS-DMA of the B operand
            default: break;
        }
    }
    // CODAE_GEMM_TILE=x: all LDS-DMA pieces on waves 0..3 (one per SIMD), their SIMD partners 4..7 only multiply
    if (cfg == 6) return launch_pipe<256, 192, 4, 2, 4>(g, s);
    return launch_pipe<256, 192, 4, 2, 6>(g, s);                 // 8 waves (64 x 96 per wave), B halves by 6 loader waves
}

// copies the DBG = 8 stamps of the first n_wg workgroups (TIMELINE_SLOTS words each) to the host
int gemm_bf16_timeline(unsigned long long* host_out, int n_wg) {
    CODAE_REQUIRE(host_out && n_wg > 0 && n_wg <= TIMELINE_WGS, "timeline: bad args");
    CODAE_HIP_CHECK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_timeline), (size_t)n_wg * TIMELINE_SLOTS * sizeof(unsigned long long)));
    return CODAE_OK;
}

}  // namespace codae
