// fp32 GEMM on the bf16 matrix cores: every fp32 operand value is cut into THREE bf16 planes while its tile is staged into
// LDS, and the product is the sum of the six plane products whose weight reaches fp32's last bit.
//
//   a ~ a0 + a1 + a2: a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1), round-to-nearest-even each time; the two
//   differences are exact in fp32, |a1| <= 2^-8 |a|, |a2| <= 2^-16 |a|, and what the three planes leave out of a is below
//   2^-25 |a| (bf16 has fp32's exponent range: no scaling, no overflow; only a value below 2^-110 loses planes to underflow).
//   a * b = the sum of nine plane products; a1 b2 + a2 b1 + a2 b2 <= 2^-23 |a b| are dropped.  The six that stay are each
//   EXACT in fp32 (8 x 8 significant bits) and are added by v_mfma_f32_16x16x32_bf16 into fp32 accumulators - the accumulation
//   the fp32 MFMA of gemm_f32.hip performs, on products that are within 1.5 * 2^-23 (relative, either sign) of the exact ones:
//   about what a separately rounded fp32 multiply would lose - and 32 of them share ONE fp32 rounding (the MFMA's k-depth) where an
//   fp32 fma chain rounds after every product.  Measured against float64: kernel level 10-15 % closer than gemm_f32.hip in every
//   form (tests/test_gpu_kernels.py); step level all first-step weight gradients of a 10-layer stack within 3.7e-7 where the fp32
//   fma-chain arithmetic (numpy, gemm_f32.hip) is 2.6e-4 off (tests/test_gpu_parity.py); integer-valued operands below 2^24 give
//   bit-exact results.  Non-finite
//   inputs come out as NaN (inf - inf in the split), which the native kernel would have passed on as inf.
//
// Why: the bf16 MFMA rate is 16 x the fp32 MFMA rate (2.5 PFLOP/s against 157 TFLOP/s dense), so six bf16 products per fp32
// product is a 2.7 x higher roofline for the SAME arithmetic - the parity-mode step of the engine, which is the only mode that
// meets the reference's rtol 1e-3 / atol 1e-5 bar, lost to the vendor's fp32 GEMM before.
//
// Same interface and epilogue as gemm_f32.hip (GemmF32: strided operands, bias, ReLU, ReLU mask, per-64-row column sums,
// split-K slabs, device-side row count).  Tile 128 x 128 x 32 per 512-thread workgroup, each of the 2 x 4 waves a 64 x 32 sub-tile of
// 4 x 2 MFMA tiles (48 MFMAs per K-tile).  Global loads run two K-tiles ahead in two register sets; the split (3 v_cvt_pk_bf16_f32,
// 4 shifts / masks, 4 subtractions per pair of values) and the LDS stores of tile t+1 are issued among the first half of tile t's
// MFMAs (the matrix pipe runs them while the wave issues on), into the other of two LDS buffers; then the one barrier of the
// K-tile, the fragment reads of tile t+1 (a second fragment register set) and the second half of tile t's MFMAs over them.  96 KB of
// LDS: one workgroup = two waves per SIMD per CU (with 4 waves of 64 x 64, one per SIMD, nothing covered a wave's LDS stores
// and round trips: 275 us for the 8192 x 1536 x 1536 forward form, of which the stores alone were 107).
#include "codae_common.h"
#include <type_traits>

namespace codae {
namespace {

constexpr int XM = 128, XN = 128, XK = 32, XT = 512;
// timing-only ablation builds (make EXTRA=-DX3_DBG=n, bits): 1 no MFMAs, 2 no split arithmetic (planes = raw halves), 4 no LDS stores
// in the loop, 8 no global loads in the loop, 64 the compiler's own issue order in the first half, 128 no workgroup barrier, 256 no
// fragment reads in the loop.  Wrong results, right instruction mix: what DESIGN.md 5f's breakdown was measured with.  0 in the
// library that ships.
#ifndef X3_DBG
#define X3_DBG 0
#endif
constexpr int PLANE = 128 * 64;        // bytes of one plane image: [128 rows][32 k] bf16
constexpr int OPND = 3 * PLANE;        // an operand's three planes
constexpr int XBUF = 2 * OPND;         // A + B

typedef __attribute__((address_space(3))) char lds_c;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) u32x2 lds_u32x2;

// Plane image: row r holds its 32 k-values as four 16-B chunks; chunk c is stored at slot c ^ T[(r >> 2) & 3], T = {0, 3, 2, 1}.
// A fragment read (ds_read_b128: lane -> row 16 t + (lane & 15), chunk lane >> 4) is serviced in the four 16-lane groups
// {0-3, 12-15, 20-27}, ... of MI355X_MICROARCH.md "LDS": rows r, r + 12 with chunk g and r + 4, r + 8 with chunk g ^ 1 share
// a 64-B half of the bank row, and T puts the four of them on its four different 16-B slots: conflict-free.
__device__ __forceinline__ int chunk_slot(int row, int chunk) { return chunk ^ ((0 - (row >> 2)) & 3); }

// Plane image of a ROW-contiguous operand (!KC: element(r, k) = P[k * ks + r], the weight matrix in the data gradient, both operands
// of the weight gradient): k-major, [32 k-rows][128 rows] bf16, 256 B per k-row, read with the transposing ds_read_b64_tr_b16 (the
// layout and swizzle of the bf16 kernels' k-strided half images, gemm_bf16_halftile.h): the 16-row (32-B) blocks of k-row kr are
// stored at block ^ key(kr), key = (kr & 3) | ((kr >> 3) & 1) << 2, so the 8 k-rows a 32-lane half of the transposed read
// touches fall on 8 different 32-B slots; a store (8 B = 4 rows of one k-row per lane, 16 lanes = 128 contiguous bytes) stays
// inside one 128-B window.  Storing this operand row-major like the other kind cost 12 ds_write_b32 per thread with 64-B global
// segments: the 1536 x 1536 x 8192 weight gradient ran 400 us, the forward form of the same size 186.
__device__ __forceinline__ int ks_block(int block, int kr) { return block ^ ((kr & 3) | (((kr >> 3) & 1) << 2)); }

// two fp32 values -> their three bf16 planes, each as one packed pair (low half = x), round-to-nearest-even at every cut
// (v_cvt_pk_bf16_f32); the two subtractions are exact in fp32
__device__ __forceinline__ void split_pair(float x, float y, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    if constexpr ((X3_DBG & 2) != 0) {
        p0 = __float_as_uint(x); p1 = __float_as_uint(y); p2 = p0 ^ p1;
        return;
    }
    p0 = pack_bf16x2(x, y);
    const float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xffff0000u);
    p1 = pack_bf16x2(rx, ry);
    const float sx = rx - __uint_as_float(p1 << 16), sy = ry - __uint_as_float(p1 & 0xffff0000u);
    p2 = pack_bf16x2(sx, sy);
}

// Load a 128 (rows) x 32 (k) tile into 8 registers per thread: unconditional 16-B loads (the launcher only takes shapes where
// they are aligned and K is whole K-tiles).  Rows past the operand's end are CLAMPED to its last rows, not zeroed: they only
// feed output rows / columns the epilogue does not store.  No branch anywhere: with a branch around a load the compiler
// loses count of the loads in flight and drains them all (s_waitcnt vmcnt(0)) at every join - the prefetch would be gone.
//   KC : element(r, k) = P[r * rs + k]   thread -> row t >> 2, k 8 (t & 3) .. + 7: reg[c]
//   !KC: element(r, k) = P[k * ks + r]   thread -> k 2 (t >> 5) + kk, rows 4 (t & 31) .. + 3   (kk = 0, 1): reg[4 kk + c]
//        (32 consecutive lanes read the 512 contiguous bytes of one k-row)
template <bool KC>
__device__ __forceinline__ void x3_load(float (&reg)[8], const float* __restrict__ P, int64_t rs, int64_t ks, int r0, int k0,
                                        int R, int t) {
    if constexpr (KC) {
        int r = r0 + (t >> 2);
        r = r < R ? r : R - 1;
        const float* src = P + (int64_t)r * rs + k0 + 8 * (t & 3);
        const float4 v = *reinterpret_cast<const float4*>(src), u = *reinterpret_cast<const float4*>(src + 4);
        reg[0] = v.x; reg[1] = v.y; reg[2] = v.z; reg[3] = v.w;
        reg[4] = u.x; reg[5] = u.y; reg[6] = u.z; reg[7] = u.w;
    } else {
        int r = r0 + 4 * (t & 31);
        r = r + 4 <= R ? r : R - 4;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int k = k0 + 2 * (t >> 5) + kk;
            const float4 v = *reinterpret_cast<const float4*>(P + (int64_t)k * ks + r);
            reg[4 * kk + 0] = v.x; reg[4 * kk + 1] = v.y; reg[4 * kk + 2] = v.z; reg[4 * kk + 3] = v.w;
        }
    }
}

// A thread's 8 values, split and stored into the operand's three plane images.
//   KC : the 8 k-values of row t >> 2: one 16-B chunk per plane (ds_write_b128; 8 consecutive lanes fill 128 consecutive bytes)
//   !KC: k-rows 2 (t >> 5), + 1, rows 4 (t & 31) .. + 3 of each: 8 B per plane and k-row into the k-major image (ds_write_b64)
template <bool KC>
__device__ __forceinline__ void x3_store(lds_c* opnd, const float (&reg)[8], int t) {
    if constexpr (KC) {
        const int row = t >> 2;
        lds_c* dst = opnd + row * 64 + (chunk_slot(row, t & 3) << 4);
        u32x4 p0, p1, p2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t a, b, c;
            split_pair(reg[2 * q], reg[2 * q + 1], a, b, c);
            p0[q] = a; p1[q] = b; p2[q] = c;
        }
        *reinterpret_cast<lds_u32x4*>(dst) = p0;
        *reinterpret_cast<lds_u32x4*>(dst + PLANE) = p1;
        *reinterpret_cast<lds_u32x4*>(dst + 2 * PLANE) = p2;
    } else {
        const int rg = t & 31;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int kr = 2 * (t >> 5) + kk;
            lds_c* dst = opnd + kr * 256 + (ks_block(rg >> 2, kr) << 5) + (rg & 3) * 8;
            u32x2 p0, p1, p2;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                uint32_t a, b, c;
                split_pair(reg[4 * kk + 2 * q], reg[4 * kk + 2 * q + 1], a, b, c);
                p0[q] = a; p1[q] = b; p2[q] = c;
            }
            *reinterpret_cast<lds_u32x2*>(dst) = p0;
            *reinterpret_cast<lds_u32x2*>(dst + PLANE) = p1;
            *reinterpret_cast<lds_u32x2*>(dst + 2 * PLANE) = p2;
        }
    }
}

// the 8 k-values lane `lane` feeds the MFMA with for 16-row tile `tile16` of a plane image: k = 8 (lane >> 4) .. + 7 of row
// 16 tile16 + (lane & 15).  Row-major image: one ds_read_b128.  k-major image: two transposing reads (k-rows 8 g + q and + 4 of the
// lane's 4-row piece; the hardware hands lane (r, g) the four k-values of row r from each)
template <bool KC>
__device__ __forceinline__ bf16x8 x3_frag(const lds_c* plane, int tile16, int lane) {
    if constexpr (KC) {
        const int r = lane & 15, g = lane >> 4;
        const int off = (16 * tile16 + r) * 64 + ((g ^ ((0 - (r >> 2)) & 3)) << 4);
        const u32x4 v = *reinterpret_cast<const lds_u32x4*>(plane + off);
        return __builtin_bit_cast(bf16x8, v);
    } else {
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const int kr = 8 * g + q;
        lds_c* src = const_cast<lds_c*>(plane) + kr * 256 + (ks_block(tile16, kr) << 5) + 8 * p;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>(src));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>(src + 4 * 256));
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    }
}

// tile id -> (tile_m, tile_n): ids walk the output in panels of 4 row tiles (down the panel, then the next column), as gemm_f32.hip
__device__ __forceinline__ void x3_tile_of(int id, int tiles_m, int tiles_n, int& tile_m, int& tile_n) {
    constexpr int GROUP_M = 4;
    const int grp = id / (GROUP_M * tiles_n);
    const int tm0 = grp * GROUP_M;
    const int gsz = tiles_m - tm0 < GROUP_M ? tiles_m - tm0 : GROUP_M;
    const int within = id - grp * (GROUP_M * tiles_n);
    tile_n = within / gsz;
    tile_m = tm0 + (within - tile_n * gsz);
}
// workgroup b of a launch runs on XCD b % 8: give each XCD a contiguous eighth of the ids
__device__ __forceinline__ int x3_xcd_remap(int id, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}

// one output tile (split-K range z of it) of one GEMM: the whole kernel body, shared by the plain kernel and the grouped one
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void x3_tile(GemmF32 g, int tile_m, int tile_n, int z, char* smem_raw) {
    lds_c* smem = (lds_c*)smem_raw;

    const int t = threadIdx.x;
    const int lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 2, wc = w & 3;          // 2 x 4 waves, 64 x 32 outputs each
    const int i0 = tile_m * XM, j0 = tile_n * XN;
    const int m_alloc = g.M;            // loads are clamped to the rows that EXIST; a device-side row count only trims the stores
    if (g.m_dev != nullptr) {
        const int m = *g.m_dev;
        if (m < g.M) g.M = m;
    }
    if (i0 >= g.M) return;

    f32x4 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int k_begin = 0, k_end = g.K;
    if (g.split_k > 1) {
        const int kt = (g.K + XK - 1) / XK;
        k_begin = (int)((int64_t)kt * z / g.split_k) * XK;
        k_end = (int)((int64_t)kt * (z + 1) / g.split_k) * XK;
        if (k_end > g.K) k_end = g.K;
        g.C += (int64_t)z * g.M * g.ldc;
    }
    const int nk = (k_end - k_begin + XK - 1) / XK;

    float ra[2][8], rb[2][8];
    // (a tile index past the end re-loads the last tile: the loop body stays one straight-line block, see x3_load)
    auto load = [&](auto set_tag, int tile) {
        constexpr int SET = decltype(set_tag)::value;
        const int k0 = k_begin + (tile < nk ? tile : nk - 1) * XK;
        x3_load<A_KC>(ra[SET], g.A, g.a_rs, g.a_ks, i0, k0, m_alloc, t);
        x3_load<B_KC>(rb[SET], g.B, g.b_rs, g.b_ks, j0, k0, g.N, t);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // (the fences keep the prologue's loads in the order the loop issues them: the compiler's count of loads in flight at the loop
    //  header is the worst case over both ways in, and a shuffled prologue made it drain everything in every other K-tile)
    load(S0{}, 0);
    asm volatile("" ::: "memory");
    load(S1{}, 1);
    asm volatile("" ::: "memory");
    x3_store<A_KC>(smem, ra[0], t);
    x3_store<B_KC>(smem + OPND, rb[0], t);
    __syncthreads();
    load(S0{}, 2);

    // Fragment registers of TWO K-tiles: the reads of tile t + 1 are issued in the middle of tile t's MFMAs, as soon as the barrier
    // has published its planes, so that their LDS round trip (18 ds_read_b128 per wave, all 8 waves at once) lies under the
    // second half of tile t's MFMAs.  (Read at the top of each K-tile, behind the barrier, the matrix pipe sat idle for the
    // whole read phase: MFMAs + fragment reads alone ran at 56 % of the MFMA rate.)
    bf16x8 af[2][4][3], bf[2][2][3];
    auto read_frags = [&](auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        const lds_c* cb = smem + SET * XBUF;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) af[SET][mt][p] = x3_frag<A_KC>(cb + p * PLANE, 4 * wr + mt, lane);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[SET][nt][p] = x3_frag<B_KC>(cb + OPND + p * PLANE, 2 * wc + nt, lane);
    };
    read_frags(S0{});
    if constexpr ((X3_DBG & 256) != 0) read_frags(S1{});

    // K-tile `tile` (fragments already in register set CUR): the first half of its MFMAs with the split + LDS stores of tile + 1
    // (register set CUR ^ 1 -> the other LDS buffer) and the global loads of tile + 3 issued between them; the barrier; the
    // fragment reads of tile + 1; the second half of the MFMAs.  One barrier per K-tile: a wave that stores tile + 2 into buffer CUR (next
    // K-tile) has passed this one's barrier, which every wave reaches only after its last use of the fragments it read from CUR.
    auto ktile = [&](int tile, auto cur_tag) {
        constexpr int CUR = decltype(cur_tag)::value, NXT = CUR ^ 1;
        lds_c* nb = smem + NXT * XBUF;
        // half h of the K-tile's MFMAs: three of the six plane products (smallest first: 2^-16, 2^-16, 2^-16 | 2^-8, 2^-8, 1
        // relative to a0 b0) for ALL eight accumulator tiles, product by product - eight independent accumulators between two
        // MFMAs of one chain.  (Split by column block instead - four accumulators x six products each - the launch took the same
        // time; with every load, store, split, fragment read and the barrier compiled out (X3_DBG 398) the MFMAs + prologue +
        // epilogue of the 8192 x 1536 x 1536 forward form take 137-143 us either way: 92 us of pure pipe time at the nominal rate.)
        auto mma_block = [&](auto h_tag) {
            constexpr int h = decltype(h_tag)::value;
            constexpr int PB[6] = {0, 1, 2, 0, 1, 0}, PA[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
            for (int q = 3 * h; q < 3 * h + 3; ++q) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        if constexpr ((X3_DBG & 1) != 0) {
                            asm volatile("" ::"v"(bf[CUR][nt][PB[q]]), "v"(af[CUR][mt][PA[q]]));
                            continue;
                        }
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[CUR][nt][PB[q]], af[CUR][mt][PA[q]], acc[mt][nt], 0, 0, 0);
                    }
                }
            }
        };
        mma_block(std::integral_constant<int, 0>{});
        // (after the last tile this stages a dummy nobody reads)
        if constexpr ((X3_DBG & 4) == 0) {
            x3_store<A_KC>(nb, ra[NXT], t);
            x3_store<B_KC>(nb + OPND, rb[NXT], t);
        }
        // the register set just staged is free: re-load it with tile + 3 at once (a K-tile and a half ahead of its use; issued behind
        // the second column block instead, one K-tile ahead, the wait for these loads was 17 % of the launch)
        if constexpr ((X3_DBG & 8) == 0) load(std::integral_constant<int, NXT>{}, tile + 3);
        if constexpr ((X3_DBG & 64) == 0) {
            // issue order of this half: the split and the LDS stores spread over the first 12 MFMAs (so that the stores have landed
            // when the barrier asks), the global loads over the next four.  (Left to itself the scheduler bunches them: 22 MFMAs in
            // a row, then the stores right in front of the barrier's lgkmcnt(0).)
            constexpr int N_DS = (A_KC ? 3 : 6) + (B_KC ? 3 : 6);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);      // eight VALU
                if ((i * N_DS) / 12 != ((i + 1) * N_DS) / 12) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // one LDS store
            }
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // one global load
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // (MFMAs of this half must not sink behind the barrier, in front of the fragment reads)
        if constexpr ((X3_DBG & 128) == 0) __syncthreads();
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // nothing moves across (in particular not the next K-tile's split, plain VALU work on the register set whose loads were
        // issued last: the wait for those loads would come up here with it)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((X3_DBG & 256) == 0) read_frags(std::integral_constant<int, NXT>{});
        __builtin_amdgcn_sched_barrier(0);      // (the scheduler sinks the reads below the MFMAs otherwise: fewer live registers)
        mma_block(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
    };
    int tile = 0;
    for (; tile + 1 < nk; tile += 2) {
        ktile(tile, S0{});
        ktile(tile + 1, S1{});
    }
    if (tile < nk) ktile(tile, S0{});

    // epilogue: swapped MFMA operands -> lane holds C[i][j .. j + 3], i = tile row (lane & 15), j = 4 (lane >> 4)
    const int li = lane & 15, jq = 4 * (lane >> 4);
    const bool vec_c = (g.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0);
    const bool vec_r = g.relu_src != nullptr && (g.ld_relu % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.relu_src) & 15) == 0);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int j = j0 + 32 * wc + 16 * nt + jq;
        float bj[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bj[r] = (g.bias != nullptr && j + r < g.N) ? g.bias[j + r] : 0.f;
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int i = i0 + 64 * wr + 16 * mt + li;
            if (i < g.M && j < g.N) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[mt][nt][r] + bj[r];
                    if (g.relu) v[r] = fmaxf(v[r], 0.f);
                }
                if (g.relu_src != nullptr) {
                    const float* m = g.relu_src + (int64_t)i * g.ld_relu + j;
                    if (vec_r && j + 3 < g.N) {
                        const float4 mv = *reinterpret_cast<const float4*>(m);
                        v[0] = mv.x > 0.f ? v[0] : 0.f; v[1] = mv.y > 0.f ? v[1] : 0.f;
                        v[2] = mv.z > 0.f ? v[2] : 0.f; v[3] = mv.w > 0.f ? v[3] : 0.f;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (j + r < g.N) v[r] = m[r] > 0.f ? v[r] : 0.f;
                    }
                }
                float* dst = g.C + (int64_t)i * g.ldc + j;
                if (vec_c && j + 3 < g.N) {
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) cs[r] += v[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (j + r < g.N) { dst[r] = v[r]; cs[r] += v[r]; }
                }
            }
        }
        if (g.colsum_part != nullptr) {
            // one partial row per 64-row wave block (as gemm_f32.hip): add over the 16 lanes that hold the block's rows
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = cs[r];
                s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
                cs[r] = s;
            }
            if (li == 0 && i0 + 64 * wr < g.M) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (j + r < g.N) g.colsum_part[(int64_t)(2 * tile_m + wr) * g.N + j + r] = cs[r];
            }
        }
    }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(XT) void gemm_f32x3_kernel(GemmF32 g) {
    __shared__ __attribute__((aligned(16))) char smem_raw[2 * XBUF];
    const int tiles_n = gridDim.x, tiles_m = gridDim.y;
    int tile_m, tile_n;
    x3_tile_of(x3_xcd_remap(blockIdx.y * tiles_n + blockIdx.x, tiles_n * tiles_m), tiles_m, tiles_n, tile_m, tile_n);
    x3_tile<A_KC, B_KC>(g, tile_m, tile_n, blockIdx.z, smem_raw);
}

// several GEMMs of the row-contiguous x row-contiguous form (the weight gradients of every layer of the stack: dW_l = dA_l^T act_l) in
// ONE launch, each with its whole K (the batch) unsplit: no slabs, no reduce pass, one prologue / epilogue per 256 K-tiles
// instead of per 37.  The XCD remap runs over the whole launch, so an XCD's workgroups work on neighbouring tiles of one GEMM.
__global__ __launch_bounds__(XT) void gemm_f32x3_grouped_kernel(GemmF32Group grp) {
    __shared__ __attribute__((aligned(16))) char smem_raw[2 * XBUF];
    const int id = x3_xcd_remap(blockIdx.x, gridDim.x);
    int j = 0;
    while (j + 1 < grp.n && id >= grp.wg_begin[j + 1]) ++j;
    const GemmF32& g = grp.g[j];
    const int tiles_m = (g.M + XM - 1) / XM, tiles_n = (g.N + XN - 1) / XN;
    int tile_m, tile_n;
    x3_tile_of(id - grp.wg_begin[j], tiles_m, tiles_n, tile_m, tile_n);
    x3_tile<false, false>(g, tile_m, tile_n, 0, smem_raw);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

bool gemm_f32x3_takes(const GemmF32& g) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.K % XK != 0) return false;
    const bool a_kc = (g.a_ks == 1), b_kc = (g.b_ks == 1);
    if (!(a_kc || g.a_rs == 1) || !(b_kc || g.b_rs == 1)) return false;
    // every load is an unconditional 16-B load: aligned bases and strides; a row-contiguous operand in whole groups of 4 rows
    if (!aligned16(g.A) || !aligned16(g.B) || (a_kc ? g.a_rs : g.a_ks) % 4 != 0 || (b_kc ? g.b_rs : g.b_ks) % 4 != 0) return false;
    if ((!a_kc && g.M % 4 != 0) || (!b_kc && g.N % 4 != 0)) return false;
    return true;
}

int gemm_f32x3(const GemmF32& g, hipStream_t s) {
    CODAE_REQUIRE(gemm_f32x3_takes(g), "gemm_f32x3: M=%d N=%d K=%d: needs K in whole 32-deep tiles and 16-byte aligned operand rows", g.M, g.N, g.K);
    const bool a_kc = (g.a_ks == 1);
    const bool b_kc = (g.b_ks == 1);
    const int split = g.split_k > 1 ? g.split_k : 1;
    CODAE_REQUIRE(split == 1 || (g.bias == nullptr && !g.relu && g.relu_src == nullptr && g.colsum_part == nullptr && g.m_dev == nullptr &&
                                 split <= g.K / XK),
                  "gemm_f32x3: split-K writes plain partial products (no epilogue terms), at most one range per K-tile");
    dim3 grid((g.N + XN - 1) / XN, (g.M + XM - 1) / XM, split);
    CODAE_REQUIRE(grid.y <= 65535, "gemm_f32x3: M=%d too large", g.M);
    if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32x3_kernel<true, true>), grid, dim3(XT), 0, s, g);
    else if (a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_f32x3_kernel<true, false>), grid, dim3(XT), 0, s, g);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32x3_kernel<false, true>), grid, dim3(XT), 0, s, g);
    else
        hipLaunchKernelGGL((gemm_f32x3_kernel<false, false>), grid, dim3(XT), 0, s, g);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int gemm_f32x3_grouped(GemmF32Group& grp, hipStream_t s) {
    CODAE_REQUIRE(grp.n >= 1 && grp.n <= CODAE_GROUP_MAX, "gemm_f32x3_grouped: %d GEMMs", grp.n);
    int total = 0;
    for (int j = 0; j < grp.n; ++j) {
        const GemmF32& g = grp.g[j];
        CODAE_REQUIRE(gemm_f32x3_takes(g) && g.a_rs == 1 && g.b_rs == 1 && g.split_k <= 1 && g.m_dev == nullptr && g.bias == nullptr && !g.relu &&
                          g.relu_src == nullptr && g.colsum_part == nullptr,
                      "gemm_f32x3_grouped: GEMM %d (M=%d N=%d K=%d) is not a plain row-contiguous x row-contiguous product", j, g.M, g.N, g.K);
        grp.wg_begin[j] = total;
        total += ((g.M + XM - 1) / XM) * ((g.N + XN - 1) / XN);
    }
    grp.wg_begin[grp.n] = total;
    hipLaunchKernelGGL(gemm_f32x3_grouped_kernel, dim3(total), dim3(XT), 0, s, grp);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

}  // namespace codae
