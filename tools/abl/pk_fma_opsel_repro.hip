// Stand-alone probe for DESIGN.md section 5d: do packed fp32 VALU ops whose operand halves are re-routed by op_sel /
// op_sel_hi give the documented result in every lane, every time - also while the matrix pipe is busy?
//
// Each lane checks a packed result against scalar v_fma_f32 / v_add_f32 / v_mul_f32 of the same operands and counts
// mismatches per lane and half.  The operands arrive the way they do in the fused-loss epilogue of gemm_bf16.hip: A from
// an LDS read turned into a difference in place, C from a packed multiply just before.
//   FORM 0  v_pk_fma_f32 D, A, A, C op_sel:[0,0,1] op_sel_hi:[1,1,0]   (src2 halves swapped)   D.lo = A.lo^2 + C.hi, D.hi = A.hi^2 + C.lo
//   FORM 1  v_pk_fma_f32 D, A, A, C                                    (no re-routing)
//   FORM 2  v_pk_add_f32 D, C, A    op_sel:[1,0]   op_sel_hi:[0,1]     (src0 halves swapped)   D.lo = C.hi + A.lo, D.hi = C.lo + A.hi
//   FORM 3  v_pk_mul_f32 D, C, A    op_sel:[1,0]   op_sel_hi:[0,1]     (src0 halves swapped)
//   FORM 4  v_pk_fma_f32 D, A, A, C op_sel:[0,0,1] op_sel_hi:[1,1,1]   (lo result takes C.hi, hi result takes C.hi)
//   FORM 5  v_pk_fma_f32 D, A, C, A op_sel:[0,1,0] op_sel_hi:[1,0,1]   (src1 halves swapped)   D.lo = A.lo * C.hi + A.lo ...
//   FORM 6  v_pk_fma_f32 D, C, A, A op_sel:[1,0,0] op_sel_hi:[0,1,1]   (src0 halves swapped)
//   FORM 7  v_pk_add_f32 D, A, C    op_sel:[0,1]   op_sel_hi:[1,0]     (src1 halves swapped)
//   FORM 8  v_pk_mul_f32 D, A, C    op_sel:[0,1]   op_sel_hi:[1,0]     (src1 halves swapped)
//   FORM 9  v_pk_fma_f32 D, A, A, C op_sel_hi:[1,1,0]                  (both results take C.lo: the broadcast form)
// MFMA placement: 0 none; 1 the SAME wave alternates 8 MFMAs / one packed op per iteration; 2 waves 0-1 of every workgroup
// only issue MFMAs, waves 2-3 only packed ops (other waves of the SIMD keep the matrix pipe busy); 3 = 1 with an s_nop 7
// x 2 between the last MFMA and the packed op of the next iteration.
//   hipcc --offload-arch=gfx950 -O3 -o pk_fma_opsel_repro pk_fma_opsel_repro.hip && ./pk_fma_opsel_repro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Sample { float a0, a1, c0, c1, got0, got1, exp0, exp1; int lane, it; };

template <int FORM, int MFMA>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ x, unsigned* __restrict__ bad, int iters, float* __restrict__ sink,
                                             Sample* __restrict__ samples, unsigned* __restrict__ n_samples) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 68];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 68; i += 256) lds[i] = x[(blockIdx.x * 977 + i) & 0xffff] * 0.25f;
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    bf16x8 fa, fb;
    for (int k = 0; k < 8; ++k) { fa[k] = (__bf16)(0.01f * (lane + k)); fb[k] = (__bf16)(0.02f * (lane - k)); }
    float keep = 0.f;
    unsigned n_lo = 0, n_hi = 0;
    const float4* xg = reinterpret_cast<const float4*>(x);
    for (int it = 0; it < iters; ++it) {
        const bool mfma_turn = (MFMA == 1 || MFMA == 3) ? ((w + it) & 1) : (MFMA == 2 ? (w < 2) : false);
        if (mfma_turn) {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
            if (MFMA == 3) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7");
            continue;
        }
        const float4 xv = xg[(blockIdx.x * 131 + it * 64 + lane + w * 17) & 0x3fff];
        const int r = (it * 5 + w * 16 + (lane >> 2)) & 63, c = (lane & 3) * 8 + ((it >> 3) & 1) * 32;
        const f32x4 ya = *reinterpret_cast<const f32x4*>(&lds[r * 68 + c]);
        f32x2 a = {xv.x - ya[0], xv.y - ya[1]};
        f32x2 cc = a * a + (f32x2){xv.z, xv.w};
        f32x2 p;
        float e0, e1;
        if constexpr (FORM == 0) {
            asm volatile("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(p) : "v"(a), "v"(cc));
            asm volatile("v_fma_f32 %0, %1, %1, %2" : "=v"(e0) : "v"(a[0]), "v"(cc[1]));
            asm volatile("v_fma_f32 %0, %1, %1, %2" : "=v"(e1) : "v"(a[1]), "v"(cc[0]));
        } else if constexpr (FORM == 1) {
            asm volatile("v_pk_fma_f32 %0, %1, %1, %2" : "=v"(p) : "v"(a), "v"(cc));
            asm volatile("v_fma_f32 %0, %1, %1, %2" : "=v"(e0) : "v"(a[0]), "v"(cc[0]));
            asm volatile("v_fma_f32 %0, %1, %1, %2" : "=v"(e1) : "v"(a[1]), "v"(cc[1]));
        } else if constexpr (FORM == 2) {
            asm volatile("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(p) : "v"(a), "v"(cc));
            asm volatile("v_add_f32 %0, %2, %1" : "=v"(e0) : "v"(a[0]), "v"(cc[1]));
            asm volatile("v_add_f32 %0, %2, %1" : "=v"(e1) : "v"(a[1]), "v"(cc[0]));
        } else if constexpr (FORM == 3) {
            asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(p) : "v"(a), "v"(cc));
            asm volatile("v_mul_f32 %0, %2, %1" : "=v"(e0) : "v"(a[0]), "v"(cc[1]));
            asm volatile("v_mul_f32 %0, %2, %1" : "=v"(e1) : "v"(a[1]), "v"(cc[0]));
        } else if constexpr (FORM == 4) {
            asm volatile("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "=v"(p) : "v"(a), "v"(cc));
            asm volatile("v_fma_f32 %0, %1, %1, %2" : "=v"(e0) : "v"(a[0]), "v"(cc[1]));
            asm volatile("v_fma_f32 %0, %1, %1, %2" : "=v"(e1) : "v"(a[1]), "v"(cc[1]));
        } else if constexpr (FORM == 5) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(p) : "v"(a), "v"(cc));
            asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(e0) : "v"(a[0]), "v"(cc[1]));
            asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(e1) : "v"(a[1]), "v"(cc[0]));
        } else if constexpr (FORM == 6) {
            asm volatile("v_pk_fma_f32 %0, %2, %1, %1 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(p) : "v"(a), "v"(cc));
            asm volatile("v_fma_f32 %0, %2, %1, %1" : "=v"(e0) : "v"(a[0]), "v"(cc[1]));
            asm volatile("v_fma_f32 %0, %2, %1, %1" : "=v"(e1) : "v"(a[1]), "v"(cc[0]));
        } else if constexpr (FORM == 7) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "v"(cc));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(a[0]), "v"(cc[1]));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(a[1]), "v"(cc[0]));
        } else if constexpr (FORM == 8) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "v"(cc));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(a[0]), "v"(cc[1]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(a[1]), "v"(cc[0]));
        } else {
            // FORM 9: hi result reads the LO half of src2 only (op_sel_hi bit cleared, op_sel all zero): the broadcast form
            asm volatile("v_pk_fma_f32 %0, %1, %1, %2 op_sel_hi:[1,1,0]" : "=v"(p) : "v"(a), "v"(cc));
            asm volatile("v_fma_f32 %0, %1, %1, %2" : "=v"(e0) : "v"(a[0]), "v"(cc[0]));
            asm volatile("v_fma_f32 %0, %1, %1, %2" : "=v"(e1) : "v"(a[1]), "v"(cc[0]));
        }
        const bool b0 = p[0] != e0, b1 = p[1] != e1;
        n_lo += b0; n_hi += b1;
        if (b0 || b1) {
            const unsigned k = atomicAdd(n_samples, 1u);
            if (k < 16) samples[k] = Sample{a[0], a[1], cc[0], cc[1], p[0], p[1], e0, e1, lane, it};
        }
        keep += xv.z;
    }
    if (n_lo) atomicAdd(&bad[lane], n_lo);
    if (n_hi) atomicAdd(&bad[64 + lane], n_hi);
    if (keep == 123.456f || acc[0] == 7.f) sink[threadIdx.x] = keep + acc[1];
}

static const char* FORMS[] = {"pk_fma src2 swapped", "pk_fma plain", "pk_add src0 swapped", "pk_mul src0 swapped", "pk_fma src2 = hi for both halves",
                              "pk_fma src1 swapped", "pk_fma src0 swapped", "pk_add src1 swapped", "pk_mul src1 swapped",
                              "pk_fma src2.lo for both halves"};
static const char* MFMAS[] = {"no MFMA", "same wave alternates MFMA / packed op", "other waves of the workgroup issue MFMAs", "same wave alternates, 32 idle cycles after the MFMAs"};

template <int FORM, int MFMA>
static void run(const float* dx, unsigned* dbad, float* dsink, Sample* dsamp, unsigned* dns, int launches, int iters) {
    (void)hipMemset(dbad, 0, 128 * sizeof(unsigned));
    (void)hipMemset(dns, 0, sizeof(unsigned));
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((probe<FORM, MFMA>), dim3(2048), dim3(256), 0, 0, dx, dbad, iters, dsink, dsamp, dns);
    (void)hipDeviceSynchronize();
    unsigned bad[128], ns = 0;
    (void)hipMemcpy(bad, dbad, sizeof(bad), hipMemcpyDeviceToHost);
    (void)hipMemcpy(&ns, dns, sizeof(ns), hipMemcpyDeviceToHost);
    unsigned long long tlo = 0, thi = 0, q[4] = {0, 0, 0, 0};
    for (int i = 0; i < 64; ++i) { tlo += bad[i]; thi += bad[64 + i]; q[i >> 4] += bad[i] + bad[64 + i]; }
    printf("%-34s | %-52s | wrong lo %8llu hi %8llu | lanes 0-15 %llu, 16-31 %llu, 32-47 %llu, 48-63 %llu\n", FORMS[FORM], MFMAS[MFMA], tlo, thi,
           q[0], q[1], q[2], q[3]);
    if (ns) {
        Sample s[16];
        (void)hipMemcpy(s, dsamp, sizeof(s), hipMemcpyDeviceToHost);
        for (unsigned i = 0; i < (ns < 3 ? ns : 3); ++i)
            printf("    lane %2d it %3d: A = (%.9g, %.9g) C = (%.9g, %.9g) got (%.9g, %.9g) expected (%.9g, %.9g); A.lo^2 + C.lo = %.9g\n", s[i].lane, s[i].it,
                   s[i].a0, s[i].a1, s[i].c0, s[i].c1, s[i].got0, s[i].got1, s[i].exp0, s[i].exp1, fmaf(s[i].a0, s[i].a0, s[i].c0));
    }
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 20, iters = argc > 2 ? atoi(argv[2]) : 512;
    std::vector<float> hx(1 << 16);
    unsigned s = 12345u;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / (float)(1 << 24); }
    float *dx, *dsink; unsigned *dbad, *dns; Sample* dsamp;
    (void)hipMalloc(&dx, hx.size() * 4); (void)hipMalloc(&dsink, 4096); (void)hipMalloc(&dbad, 128 * 4); (void)hipMalloc(&dns, 4);
    (void)hipMalloc(&dsamp, 16 * sizeof(Sample));
    (void)hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    printf("%d launches x 2048 workgroups x 256 threads x %d iterations\n", launches, iters);
#define RUN(F) run<F, 0>(dx, dbad, dsink, dsamp, dns, launches, iters); run<F, 1>(dx, dbad, dsink, dsamp, dns, launches, iters); \
               run<F, 2>(dx, dbad, dsink, dsamp, dns, launches, iters); run<F, 3>(dx, dbad, dsink, dsamp, dns, launches, iters);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9)
    return 0;
}
