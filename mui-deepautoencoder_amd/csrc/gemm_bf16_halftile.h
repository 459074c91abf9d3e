// Half-tile LDS images of the phase-pipelined bf16 GEMM kernels (gemm_bf16_pipe.hip, gemm_bf16_snake.hip): LDS-DMA
// staging with the bank swizzle on the SOURCE address, fragment reads (ds_read_b128 for k-contiguous operands, asm-issued
// ds_read_b64_tr_b16 for k-strided ones), counted waits and the raw phase barrier.  Included inside each kernel file's
// anonymous namespace.
#pragma once

constexpr int BK = 64;

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) const void gvoid;

__device__ __forceinline__ void glds16(const void* gsrc, lds_char* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((gvoid*)gsrc, (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// All but the wave's KEEP youngest LDS reads have returned; then the workgroup barrier.  The LDS-DMA issued after
// the barrier overwrites a region whose last readers ran TWO phases ago, so the reads of the phase just finished
// (KEEP of them, compiler-visible ds_read_b128 only) may stay in flight across it.
template <int KEEP = 0, bool PIN = false>
__device__ __forceinline__ void phase_barrier() {
    // PIN (the k-contiguous forms: forward, data gradient, fused loss): nothing is scheduled across the phase boundary.  Left free, the
    // compiler moved MFMAs across the barriers until the phases held 3 to 22 of them instead of 12 each, and all eight waves starve
    // the matrix pipe in the same thin phases (whole C3 step 1.197 -> 1.190 ms with the phases pinned, -> 1.182 with the issue order
    // inside them fixed as well: gemm_bf16_pipe.hip).  The k-strided forms stay free: pinned, the grouped weight gradients lose
    // 0.3 us each and the split-K weight gradient of the data-parallel backward, which shares the chip with the data-gradient
    // chain, 10 % (one-rank rehearsal 1.44 -> 1.55 ms/step, tools/abl/ab_dp.sh).
    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(KEEP) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
}

// ---- half-tile images ------------------------------------------------------------------------
// Half-tile h is the contiguous range [h * RH, (h+1) * RH) of the tile's rows (columns); inside it wave row
// (column) w owns [w * S/2, (w+1) * S/2).  So a wave's two halves are RH apart in the output, and a k-strided
// half image fetches ONE contiguous 2*RH-byte segment per k-row (interleaving the waves' halves instead fetched
// two half-used segments and cost 20-35 % on the dgrad / wgrad forms).
template <int S>
__device__ __forceinline__ int half_to_tile(int l, int h, int RH) {
    return h * RH + l;
}

// KS image of a half: [64 k-rows][RH columns], 32-B blocks swizzled per k-row so that the 8 rows
// one 32-lane half of ds_read_b64_tr_b16 touches fall on 8 different 32-B slots:
//   RH = 128 (256-B rows): block' = block ^ ((kr & 3) | (((kr >> 3) & 1) << 2))
//   RH =  96 (192-B rows, slot = (6 kr + block') mod 8): block' = (block + ((kr >> 3) & 1)) mod 6
template <int RH>
__device__ __forceinline__ int ks_to_lds_block(int block, int kr) {
    if constexpr (RH == 96) {
        const int b = block + ((kr >> 3) & 1);
        return b >= 6 ? b - 6 : b;
    } else {
        static_assert(RH == 128, "KS half image: 96 or 128 columns");
        return block ^ ((kr & 3) | (((kr >> 3) & 1) << 2));
    }
}
template <int RH>
__device__ __forceinline__ int ks_from_lds_block(int lds_block, int kr) {
    if constexpr (RH == 96) {
        const int b = lds_block - ((kr >> 3) & 1);
        return b < 0 ? b + 6 : b;
    } else {
        return lds_block ^ ((kr & 3) | (((kr >> 3) & 1) << 2));
    }
}

// Global source (at k = 0), as a BYTE OFFSET from the operand's base, of the 16 bytes lane `lane` of loader wave `w`
// places with its `it`-th LDS-DMA instruction of half-tile h (instruction j = it * NW + w writes img + j * 1024 +
// lane * 16).  Everything here is loop invariant; the K advance is wave-uniform and goes into the scalar base, so the
// DMA instruction takes {SGPR base, 32-bit VGPR offset}: half the address registers of per-lane 64-bit pointers and no
// 64-bit VALU add per issue (operands are < 4 GiB: checked at launch).
template <int MODE, int RH, int S, int NW>
__device__ __forceinline__ uint32_t half_src(int64_t ld, int r0, int rmax, int h, int it, int w, int lane) {
    const int j = it * NW + w;
    if constexpr (MODE == OP_KC) {
        const int lr = j * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((lr >> 1) & 7);
        int grow = r0 + half_to_tile<S>(lr, h, RH);
        grow = grow < rmax ? grow : rmax - 1;
        return (uint32_t)(((int64_t)grow * ld + c * 8) * 2);
    } else {
        constexpr int CPR = RH / 8;
        const int q = j * 64 + lane;
        const int kr = q / CPR;
        const int cp = q - kr * CPR;
        const int lc = (ks_from_lds_block<RH>(cp >> 1, kr) * 2 + (cp & 1)) * 8;
        int col = r0 + half_to_tile<S>(lc, h, RH);
        col = col + 8 <= rmax ? col : rmax - 8;
        return (uint32_t)(((int64_t)kr * ld + col) * 2);
    }
}

// ds_read_b64_tr_b16 issued behind the compiler's back.  A transposed LDS read the compiler knows about
// gets an `s_waitcnt vmcnt(0)` in front of it whenever LDS-DMA loads are in flight (it cannot tell the
// images apart), which drains the whole 7-phase prefetch queue in every phase.  The kernel's own
// protocol already orders these reads: every fragment is consumed only after the next phase_barrier()
// (s_waitcnt lgkmcnt(0) + s_barrier), where settle() hands the registers back to the compiler.
template <int OFF>
__device__ __forceinline__ s16x4 lds_read_tr16(const lds_char* p) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"((uint32_t)(uintptr_t)p), "n"(OFF));
    return v;
}
// after the wait that covers their reads: MFMAs consuming `f` cannot be scheduled above this point
template <int N>
__device__ __forceinline__ void settle(bf16x8 (&f)[N][2]) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        asm volatile("" : "+v"(f[i][0]));
        asm volatile("" : "+v"(f[i][1]));
    }
}

// 8-element MFMA fragment: 16-wide tile `t` (local to the half image), k-step S.
template <int MODE, int RH, int S>
__device__ __forceinline__ bf16x8 read_frag(const lds_char* img, int t, int lane) {
    if constexpr (MODE == OP_KC) {
        const int r = lane & 15, g = lane >> 4;
        const int off = (16 * t + r) * 128 + (((4 * S + g) ^ (r >> 1)) << 4);
        const s16x8 v = *reinterpret_cast<const __attribute__((address_space(3))) s16x8*>(img + off);
        return __builtin_bit_cast(bf16x8, v);
    } else {
        // the swizzle key depends on (kr & 3) and ((kr >> 3) & 1) only, i.e. not on the k-step: both k-steps read from
        // ONE address register, the second through the instruction's offset field (half the address VGPRs)
        const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const int kr = 8 * g + q;
        const int off = kr * (2 * RH) + (ks_to_lds_block<RH>(t, kr) << 5) + 8 * p;
        const s16x4 lo = lds_read_tr16<S * 32 * 2 * RH>(img + off);
        const s16x4 hi = lds_read_tr16<S * 32 * 2 * RH + 4 * 2 * RH>(img + off);
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    }
}

