// Which bf16 MFMA shape sustains the higher rate on this part when NOTHING else runs: v_mfma_f32_16x16x32_bf16 (what the GEMM kernels
// use) or v_mfma_f32_32x32x16_bf16 (same nominal flops per cycle, half the operand-register reads per flop)?  Pure register-operand
// loops, 8 waves per CU (2 per SIMD) like the 256 x 192 tile, independent accumulators as in the kernel (6 per wave and k-step).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape_rate mfma_shape_rate.hip && ./mfma_shape_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(512) void spin(float* out, int iters) {
    bf16x8 a[4], b[6];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 8; ++k) a[i][k] = (__bf16)(0.001f * (threadIdx.x + i + k));
    for (int i = 0; i < 6; ++i) for (int k = 0; k < 8; ++k) b[i][k] = (__bf16)(0.002f * (threadIdx.x - i + k));
    float keep = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[4][6];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) keep += acc[i][j][0];
    } else {
        f32x16 acc[2][3];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j + 3 * s], a[i + 2 * s], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) keep += acc[i][j][0];
    }
    if (keep == 12345.f) out[threadIdx.x] = keep;
}

template <int SHAPE>
static void run(float* d, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(spin<SHAPE>, dim3(256), dim3(512), 0, 0, d, iters / 10);
    (void)hipDeviceSynchronize();
    float best = 1e9f, sum = 0.f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(spin<SHAPE>, dim3(256), dim3(512), 0, 0, d, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        sum += ms; if (ms < best) best = ms;
    }
    // per iteration and wave: 24 MFMAs of 16x16x32 (16384 flops) or 12 of 32x32x16 (32768 flops) = 393216 flops; 2048 waves
    const double flops = 393216.0 * 2048.0 * iters;
    printf("v_mfma_f32_%s_bf16: %d iterations: mean %.1f us, best %.1f us = %.0f TFLOP/s mean, %.0f best (nominal peak 2500)\n",
           SHAPE == 16 ? "16x16x32" : "32x32x16", iters, sum / 5 * 1e3, best * 1e3, flops / (sum / 5 * 1e-3) / 1e12, flops / (best * 1e-3) / 1e12);
}

int main() {
    float* d; (void)hipMalloc(&d, 4096);
    for (int rep = 0; rep < 2; ++rep) {
        run<16>(d, 4000); run<32>(d, 4000);      // ~100-200 us: the length of a GEMM launch
        run<16>(d, 40000); run<32>(d, 40000);    // ~1.5 ms: sustained
    }
    return 0;
}
