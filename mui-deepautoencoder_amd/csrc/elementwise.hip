// HBM-bound pieces of the DAE step: batch gather + slot corruption, mask expansion,
// fused MSE loss forward/backward with the reference's per-step metrics, gradient norm,
// clip + Adam with bf16 shadow refresh.  All are single-pass, 16 B per lane where the
// row width allows, grid capped at 2048 blocks with a grid-stride loop.
#include "codae_common.h"

namespace codae {
namespace {

constexpr int NT = 256;

inline int grid_for(int64_t work_items) {
    int64_t b = (work_items + NT - 1) / NT;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// sum over the 256-thread block; result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* red /*[4]*/) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return r;
}

__device__ __forceinline__ uint2 pack_bf16x4(float a, float b, float c, float d) {
    uint2 o;
    o.x = (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
    o.y = (uint32_t)f32_to_bf16(c) | ((uint32_t)f32_to_bf16(d) << 16);
    return o;
}

// The training step's form of the kernel below (bf16 out, io % 8 == 0): 8 elements per thread = two 16-B loads, one
// 8-B mask load, one 16-B store (the 4-element form stores 8 B per lane, half-width store instructions).
__global__ __launch_bounds__(NT) void gather_corrupt_bf16x8_kernel(const float* __restrict__ data,
                                                                   const int32_t* __restrict__ row_idx,
                                                                   const int32_t* __restrict__ mask_id,
                                                                   const uint8_t* __restrict__ table, int B, int io,
                                                                   bf16_t* __restrict__ out,
                                                                   const int32_t* __restrict__ mask_to_use, int nb_run,
                                                                   int run, int64_t out_ld) {
    const bool masked = (mask_id != nullptr) || (mask_to_use != nullptr);
    const int cols = io / 8;
    const int64_t total = (int64_t)B * cols;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total; e += (int64_t)gridDim.x * NT) {
        const int b = (int)(e / cols);
        const int c = (int)(e - (int64_t)b * cols) * 8;
        const int64_t src_row = row_idx ? row_idx[b] : b;
        const float* src = data + src_row * io + c;
        const float4 x0 = *reinterpret_cast<const float4*>(src);
        const float4 x1 = *reinterpret_cast<const float4*>(src + 4);
        float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        if (masked) {
            const int id = mask_id ? mask_id[b] : mask_to_use[src_row * nb_run + run];
            const uint2 m = *reinterpret_cast<const uint2*>(table + (int64_t)id * io + c);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (((k < 4 ? m.x : m.y) >> (8 * (k & 3))) & 0xff) ? v[k] : 0.f;
        }
        uint4 o;
        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
        o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4*>(out + (int64_t)b * out_ld + c) = o;
    }
}

// ---- a2 + a10: out[b][:] = data[row_idx[b]][:] * mask_table[mask_id[b]][:] -----------------
// (collate_embedding data_tool.py:96-103 + corrupt embedding_...py:226-239 fused; the [B,io]
// fp32 mask of Corrupter.get_masks is never materialised.)
template <bool VEC, bool OUT_BF16>
__global__ __launch_bounds__(NT) void gather_corrupt_kernel(const float* __restrict__ data,
                                                            const int32_t* __restrict__ row_idx,
                                                            const int32_t* __restrict__ mask_id,
                                                            const uint8_t* __restrict__ table, int B, int io,
                                                            void* __restrict__ out,
                                                            const int32_t* __restrict__ mask_to_use, int nb_run,
                                                            int run, int64_t out_ld) {
    constexpr int W = VEC ? 4 : 1;
    const bool masked = (mask_id != nullptr) || (mask_to_use != nullptr);
    const int cols = io / W;
    const int64_t total = (int64_t)B * cols;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total; e += (int64_t)gridDim.x * NT) {
        const int b = (int)(e / cols);
        const int c = (int)(e - (int64_t)b * cols) * W;
        const int64_t src_row = row_idx ? row_idx[b] : b;
        const float* src = data + src_row * io + c;
        float v[4];
        if constexpr (VEC) {
            const float4 x = *reinterpret_cast<const float4*>(src);
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
            if (masked) {
                const int id = mask_id ? mask_id[b] : mask_to_use[src_row * nb_run + run];
                const uint32_t m = *reinterpret_cast<const uint32_t*>(table + (int64_t)id * io + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = ((m >> (8 * k)) & 0xff) ? v[k] : 0.f;
            }
        } else {
            v[0] = src[0];
            if (masked) {
                const int id = mask_id ? mask_id[b] : mask_to_use[src_row * nb_run + run];
                v[0] = table[(int64_t)id * io + c] ? v[0] : 0.f;
            }
        }
        const int64_t o = (int64_t)b * out_ld + c;
        if constexpr (OUT_BF16) {
            bf16_t* op = reinterpret_cast<bf16_t*>(out) + o;
            if constexpr (VEC) *reinterpret_cast<uint2*>(op) = pack_bf16x4(v[0], v[1], v[2], v[3]);
            else op[0] = f32_to_bf16(v[0]);
        } else {
            float* op = reinterpret_cast<float*>(out) + o;
            if constexpr (VEC) *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
            else op[0] = v[0];
        }
    }
}

__global__ __launch_bounds__(NT) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                       int64_t n) {
    const int64_t n4 = n / 4;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n4; e += (int64_t)gridDim.x * NT) {
        const float4 x = reinterpret_cast<const float4*>(src)[e];
        reinterpret_cast<uint2*>(dst)[e] = pack_bf16x4(x.x, x.y, x.z, x.w);
    }
    for (int64_t e = n4 * 4 + (int64_t)blockIdx.x * NT + threadIdx.x; e < n; e += (int64_t)gridDim.x * NT)
        dst[e] = f32_to_bf16(src[e]);
}

__global__ __launch_bounds__(NT) void cast_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst,
                                                      int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n; e += (int64_t)gridDim.x * NT)
        dst[e] = bf16_to_f32(src[e]);
}

// model.corrupt on dense tensors: out = x * mask
__global__ __launch_bounds__(NT) void corrupt_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                     float* __restrict__ out, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n; e += (int64_t)gridDim.x * NT)
        out[e] = x[e] * m[e];
}

// Corrupter.get_masks (data_tool.py:252-262)
__global__ __launch_bounds__(NT) void expand_masks_kernel(const int32_t* __restrict__ mask_id,
                                                          const uint8_t* __restrict__ table,
                                                          const int32_t* __restrict__ k_of_mask, int B, int io,
                                                          int k_max, float* __restrict__ masks,
                                                          float* __restrict__ fmask) {
    const int64_t total = (int64_t)B * io;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total; e += (int64_t)gridDim.x * NT) {
        const int b = (int)(e / io);
        const int c = (int)(e - (int64_t)b * io);
        const int id = mask_id[b];
        const float v = table[(int64_t)id * io + c] ? 1.f : 0.f;
        const int k = k_of_mask[id] - 1;
        for (int kk = 0; kk < k_max; ++kk) masks[(int64_t)kk * total + e] = (kk == k) ? v : 0.f;
        fmask[e] = v;
    }
}

// ---- a4 + a11: MSELoss(mean) forward/backward + the per-step metric sums --------------------
// One block owns ROWS consecutive batch rows and sweeps all columns; the bias gradient of the last
// Linear (column sums of dy) leaves as one partial-sum row per block (plain stores, added up in block
// order by bias_finish_kernel: deterministic).
//   dy = 2 (y - x) * inv_n                          (autograd of train_dae_on_embedding.py:206)
//   SQ_FULL    += sum (x-y)^2                        (:218-220)
//   SQ_PARTIAL += sum (1-fmask)(x-y)^2               (:223)
constexpr int LOSS_ROWS = 32;   // rows per block = rows per partial column-sum row
constexpr int LOSS_UNROLL = 8;  // rows in flight per thread
template <bool VEC, bool DY_BF16>
__global__ __launch_bounds__(NT) void mse_loss_kernel(const float* __restrict__ data,
                                                      const int32_t* __restrict__ row_idx,
                                                      const int32_t* __restrict__ mask_id,
                                                      const uint8_t* __restrict__ table, int B, int io,
                                                      const float* __restrict__ y, void* __restrict__ dy,
                                                      float inv_n, float* __restrict__ colsum_part,
                                                      double* __restrict__ loss_parts, int want_grad,
                                                      const int32_t* __restrict__ mask_to_use, int nb_run, int run, int64_t dy_ld) {
    __shared__ float red[4];
    const bool masked = (mask_id != nullptr) || (mask_to_use != nullptr);
    constexpr int W = VEC ? 4 : 1;
    const int cols = io / W;
    const int r_begin = blockIdx.x * LOSS_ROWS;
    float sq = 0.f, sqp = 0.f;
    for (int cv = threadIdx.x; cv < cols; cv += NT) {
        const int c = cv * W;
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
        // rows are independent: unrolled with clamped (always valid) addresses so that the loads of
        // all LOSS_ROWS rows are in flight together; rows past the batch contribute nothing
        for (int r0 = 0; r0 < LOSS_ROWS; r0 += LOSS_UNROLL)
#pragma unroll
        for (int ru = 0; ru < LOSS_UNROLL; ++ru) {
            const int rr = r0 + ru;
            const bool live = r_begin + rr < B;
            const int b = live ? r_begin + rr : B - 1;
            const int64_t src_row = row_idx ? row_idx[b] : b;
            float xv[4], yv[4];
            uint32_t m = 0x01010101u;
            const int id = !masked ? 0 : (mask_id ? mask_id[b] : mask_to_use[src_row * nb_run + run]);
            if constexpr (VEC) {
                const float4 x4 = *reinterpret_cast<const float4*>(data + src_row * io + c);
                const float4 y4 = *reinterpret_cast<const float4*>(y + (int64_t)b * io + c);
                xv[0] = x4.x; xv[1] = x4.y; xv[2] = x4.z; xv[3] = x4.w;
                yv[0] = y4.x; yv[1] = y4.y; yv[2] = y4.z; yv[3] = y4.w;
                if (masked) m = *reinterpret_cast<const uint32_t*>(table + (int64_t)id * io + c);
            } else {
                xv[0] = data[src_row * io + c];
                yv[0] = y[(int64_t)b * io + c];
                if (masked) m = table[(int64_t)id * io + c];
            }
            float g[4];
#pragma unroll
            for (int k = 0; k < W; ++k) {
                const float d = live ? xv[k] - yv[k] : 0.f;
                const float se = d * d;
                sq += se;
                if (((m >> (8 * k)) & 0xff) == 0) sqp += se;
                g[k] = -2.f * d * inv_n;
                cs[k] += g[k];
            }
            if (want_grad && live) {
                const int64_t o = (int64_t)b * dy_ld + c;
                if constexpr (DY_BF16) {
                    bf16_t* op = reinterpret_cast<bf16_t*>(dy) + o;
                    if constexpr (VEC) *reinterpret_cast<uint2*>(op) = pack_bf16x4(g[0], g[1], g[2], g[3]);
                    else op[0] = f32_to_bf16(g[0]);
                } else {
                    float* op = reinterpret_cast<float*>(dy) + o;
                    if constexpr (VEC) *reinterpret_cast<float4*>(op) = make_float4(g[0], g[1], g[2], g[3]);
                    else op[0] = g[0];
                }
            }
        }
        if (want_grad && colsum_part) {
#pragma unroll
            for (int k = 0; k < W; ++k) colsum_part[(int64_t)blockIdx.x * io + c + k] = cs[k];
        }
    }
    const float bsq = block_sum(sq, red);
    const float bsqp = block_sum(sqp, red);
    if (threadIdx.x == 0) {
        loss_parts[2 * blockIdx.x] = (double)bsq;
        loss_parts[2 * blockIdx.x + 1] = masked ? (double)bsqp : 0.0;
    }
}

// dense variant for the drop-in path (x, y, fmask already materialised)
__global__ __launch_bounds__(NT) void mse_dense_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       const float* __restrict__ fmask, float* __restrict__ dy,
                                                       int64_t n, float inv_n, double* __restrict__ scalars) {
    __shared__ float red[4];
    float sq = 0.f, sqp = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n; e += (int64_t)gridDim.x * NT) {
        const float d = x[e] - y[e];
        const float se = d * d;
        sq += se;
        if (fmask) sqp += (1.f - fmask[e]) * se;
        if (dy) dy[e] = -2.f * d * inv_n;
    }
    const float bsq = block_sum(sq, red);
    const float bsqp = block_sum(sqp, red);
    if (threadIdx.x == 0) {
        atomicAdd(&scalars[CODAE_S_SQ_FULL], (double)bsq);
        atomicAdd(&scalars[CODAE_S_STEP_SQ], (double)bsq);
        if (fmask) atomicAdd(&scalars[CODAE_S_SQ_PARTIAL], (double)bsqp);
    }
}

// Per-workgroup metric sums -> the epoch accumulators (added in index order: deterministic); LAST_LOSS = step sum * inv_n;
// reset the per-step accumulators.  One block.
__global__ __launch_bounds__(NT) void finish_loss_kernel(double* scalars, double inv_n, const double* __restrict__ parts,
                                                         int n_parts) {
    __shared__ double red[2][NT];
    if (parts != nullptr) {
        double a = 0.0, b = 0.0;
        for (int i = threadIdx.x; i < n_parts; i += NT) { a += parts[2 * i]; b += parts[2 * i + 1]; }
        red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
        __syncthreads();
        for (int o = NT / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        double step = scalars[CODAE_S_STEP_SQ];
        if (parts != nullptr) {
            step = red[0][0];
            scalars[CODAE_S_SQ_FULL] += red[0][0];
            scalars[CODAE_S_SQ_PARTIAL] += red[1][0];
        }
        scalars[CODAE_S_LAST_LOSS] = step * inv_n;
        scalars[CODAE_S_STEP_SQ] = 0.0;
        scalars[CODAE_S_GRAD_SQ] = 0.0;
    }
    if (threadIdx.x < CODAE_S_N_SLOTS) scalars[CODAE_S_GRAD_SQ_SLOTS + threadIdx.x] = 0.0;
}

// ---- a7: sum g^2 (clip_grad_norm_, train_dae_on_embedding.py:213) ---------------------------
__global__ __launch_bounds__(NT) void sumsq_kernel(const float* __restrict__ g, int64_t n, double* out) {
    __shared__ float red[4];
    float s = 0.f;
    const int64_t n4 = n / 4;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n4; e += (int64_t)gridDim.x * NT) {
        const float4 v = reinterpret_cast<const float4*>(g)[e];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (int64_t e = n4 * 4 + (int64_t)blockIdx.x * NT + threadIdx.x; e < n; e += (int64_t)gridDim.x * NT)
        s += g[e] * g[e];
    const float b = block_sum(s, red);
    // `out` points at scalars[GRAD_SQ]; scatter over the slot array that follows the named scalars
    if (threadIdx.x == 0)
        atomicAdd(out + (CODAE_S_GRAD_SQ_SLOTS - CODAE_S_GRAD_SQ) + (blockIdx.x & (CODAE_S_N_SLOTS - 1)), (double)b);
}

// *acc += sum g^2 with at most 64 blocks (a rank's parameter shard in the sharded update): one address, few adders
__global__ __launch_bounds__(NT) void sumsq_to_kernel(const float* __restrict__ g, int64_t n, double* acc) {
    __shared__ float red[4];
    float s = 0.f;
    const int64_t n4 = n / 4;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n4; e += (int64_t)gridDim.x * NT) {
        const float4 v = reinterpret_cast<const float4*>(g)[e];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (int64_t e = n4 * 4 + (int64_t)blockIdx.x * NT + threadIdx.x; e < n; e += (int64_t)gridDim.x * NT) s += g[e] * g[e];
    const float b = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(acc, (double)b);
}

// clip_grad_norm_'s coefficient from a total sum g^2 (train_dae_on_embedding.py:213): min(1, max_norm / (norm + 1e-6))
__global__ void clip_coef_kernel(const double* total_sq, float max_norm, double* coef_out) {
    const float total = sqrtf((float)*total_sq);
    *coef_out = (double)fminf(1.f, max_norm / (total + 1e-6f));
}

// ---- a7 + a8: clip scale folded into Adam (torch.optim.Adam, amsgrad off, L2 decay) ----------
struct AdamConst {
    float lr_over_bc1, inv_sqrt_bc2, beta1, beta2, eps, wd, max_norm;
    float lr;                // graph replay: the bias corrections are rebuilt on the device from *step_dev
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float coef, const AdamConst& c) {
    g = g * coef + c.wd * p;
    m = c.beta1 * m + (1.f - c.beta1) * g;
    v = c.beta2 * v + (1.f - c.beta2) * g * g;
    const float denom = sqrtf(v) * c.inv_sqrt_bc2 + c.eps;
    p = p - c.lr_over_bc1 * (m / denom);
}

__global__ __launch_bounds__(NT) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                       AdamConst c, const double* __restrict__ grad_sq,
                                                       bf16_t* __restrict__ shadow, const double* __restrict__ coef_in,
                                                       const double* __restrict__ step_dev) {
    if (step_dev != nullptr) {
        // replayed from a hipGraph: the step count lives in device memory (kernel arguments are frozen at capture)
        const double t = *step_dev;
        c.lr_over_bc1 = (float)((double)c.lr / (1.0 - pow((double)c.beta1, t)));
        c.inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)c.beta2, t)));
    }
    float coef = 1.f;
    if (coef_in != nullptr) {
        coef = (float)(*coef_in);
    } else if (c.max_norm > 0.f) {
        // sum g^2 = scalars[GRAD_SQ] + its 64 partial slots (one wave adds them up, LDS broadcasts)
        __shared__ double total_sq;
        if (threadIdx.x < 64) {
            double v = grad_sq[(CODAE_S_GRAD_SQ_SLOTS - CODAE_S_GRAD_SQ) + threadIdx.x];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (threadIdx.x == 0) total_sq = v + grad_sq[0];
        }
        __syncthreads();
        const float total = sqrtf((float)total_sq);
        coef = fminf(1.f, c.max_norm / (total + 1e-6f));
    }
    const int64_t n4 = n / 4;  // n is padded to a multiple of 64 by the engine; tail handled below anyway
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n4; e += (int64_t)gridDim.x * NT) {
        float4 pp = reinterpret_cast<float4*>(p)[e];
        const float4 gg = reinterpret_cast<const float4*>(g)[e];
        float4 mm = reinterpret_cast<float4*>(m)[e];
        float4 vv = reinterpret_cast<float4*>(v)[e];
        adam_one(pp.x, gg.x, mm.x, vv.x, coef, c);
        adam_one(pp.y, gg.y, mm.y, vv.y, coef, c);
        adam_one(pp.z, gg.z, mm.z, vv.z, coef, c);
        adam_one(pp.w, gg.w, mm.w, vv.w, coef, c);
        reinterpret_cast<float4*>(p)[e] = pp;
        reinterpret_cast<float4*>(m)[e] = mm;
        reinterpret_cast<float4*>(v)[e] = vv;
        if (shadow) reinterpret_cast<uint2*>(shadow)[e] = pack_bf16x4(pp.x, pp.y, pp.z, pp.w);
    }
    for (int64_t e = n4 * 4 + (int64_t)blockIdx.x * NT + threadIdx.x; e < n; e += (int64_t)gridDim.x * NT) {
        float pp = p[e], mm = m[e], vv = v[e];
        adam_one(pp, g[e], mm, vv, coef, c);
        p[e] = pp; m[e] = mm; v[e] = vv;
        if (shadow) shadow[e] = f32_to_bf16(pp);
    }
}

// out[i] = sum_s slabs[s][i]  (split-K partials of the weight-gradient GEMM); optionally also
// sumsq += sum out[i]^2 so that clip_grad_norm_ needs no second pass over the gradient
__global__ __launch_bounds__(NT) void reduce_slabs_kernel(const float* __restrict__ slabs, int n_slabs,
                                                          int64_t stride, float* __restrict__ out, int64_t n,
                                                          double* __restrict__ sumsq) {
    __shared__ float red[4];
    float sq = 0.f;
    const int64_t n4 = n / 4;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n4; e += (int64_t)gridDim.x * NT) {
        float4 a = reinterpret_cast<const float4*>(slabs)[e];
        for (int s = 1; s < n_slabs; ++s) {
            const float4 b = reinterpret_cast<const float4*>(slabs + s * stride)[e];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        reinterpret_cast<float4*>(out)[e] = a;
        sq += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    }
    for (int64_t e = n4 * 4 + (int64_t)blockIdx.x * NT + threadIdx.x; e < n; e += (int64_t)gridDim.x * NT) {
        float a = slabs[e];
        for (int s = 1; s < n_slabs; ++s) a += slabs[s * stride + e];
        out[e] = a;
        sq += a * a;
    }
    if (sumsq != nullptr) {
        const float bsq = block_sum(sq, red);
        if (threadIdx.x == 0)
            atomicAdd(sumsq + (CODAE_S_GRAD_SQ_SLOTS - CODAE_S_GRAD_SQ) + (blockIdx.x & (CODAE_S_N_SLOTS - 1)), (double)bsq);
    }
}


// Split-K fp32 GEMM, second stage WITH the GEMM's epilogue (exact-fp32 forward / data-gradient launches of a small batch:
// 128 rows of a 1536-wide layer are 12 workgroups walking 48 K-tiles each - 124 / 178 us per launch; split over K they
// fill the chip and this kernel finishes them): C[i][j] = epi(sum_z slabs[z][i][j]), epi = + bias, ReLU, * [relu_src > 0];
// colsum_part[i / 64][j] = sum of the stored values over the block's 64 rows (the next bias gradient's partial rows, as
// gemm_f32 writes them).  One workgroup = 64 rows x 64 columns: thread -> 4 columns x 4 rows (16 apart).
__global__ __launch_bounds__(NT) void reduce_slabs_epi_kernel(const float* __restrict__ slabs, int n_slabs, int64_t stride, int M, int N,
                                                              float* __restrict__ C, int64_t ldc, const float* __restrict__ bias, int relu,
                                                              const float* __restrict__ relu_src, int64_t ld_relu,
                                                              float* __restrict__ colsum_part) {
    __shared__ float red[16][64 + 4];
    const int cg = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int j = blockIdx.x * 64 + cg * 4;
    const int i0 = blockIdx.y * 64;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    if (j < N) {                                               // (N % 4 == 0: checked by the launcher)
        float4 bj = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias != nullptr) bj = *reinterpret_cast<const float4*>(bias + j);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + rg + 16 * q;
            if (i >= M) break;
            float4 a = *reinterpret_cast<const float4*>(slabs + (int64_t)i * N + j);
            for (int z = 1; z < n_slabs; ++z) {
                const float4 b = *reinterpret_cast<const float4*>(slabs + z * stride + (int64_t)i * N + j);
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            a.x += bj.x; a.y += bj.y; a.z += bj.z; a.w += bj.w;
            if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
            if (relu_src != nullptr) {
                const float4 h = *reinterpret_cast<const float4*>(relu_src + (int64_t)i * ld_relu + j);
                a.x = h.x > 0.f ? a.x : 0.f; a.y = h.y > 0.f ? a.y : 0.f; a.z = h.z > 0.f ? a.z : 0.f; a.w = h.w > 0.f ? a.w : 0.f;
            }
            *reinterpret_cast<float4*>(C + (int64_t)i * ldc + j) = a;
            cs[0] += a.x; cs[1] += a.y; cs[2] += a.z; cs[3] += a.w;
        }
    }
    if (colsum_part != nullptr) {
        red[rg][cg * 4 + 0] = cs[0]; red[rg][cg * 4 + 1] = cs[1]; red[rg][cg * 4 + 2] = cs[2]; red[rg][cg * 4 + 3] = cs[3];
        __syncthreads();
        if (threadIdx.x < 64 && blockIdx.x * 64 + (int)threadIdx.x < N) {
            float t = 0.f;
            for (int r = 0; r < 16; ++r) t += red[r][threadIdx.x];
            colsum_part[(int64_t)blockIdx.y * N + blockIdx.x * 64 + threadIdx.x] = t;
        }
    }
}

// parts[block][c] = sum over the block's 64 rows of src[r][c]   (bias gradient of a dense dy, first stage)
// grid (row blocks, column chunks of NT): 8 row loads in flight per thread, added in row order (the same bits as a plain loop:
// that form - 128 blocks walking 64 dependent loads each - took 94 us for the 50 MB dy of the drop-in backward at C3)
__global__ __launch_bounds__(NT) void colsum_parts_f32_kernel(const float* __restrict__ src, int M, int N,
                                                              float* __restrict__ parts) {
    const int r_begin = blockIdx.x * 64;
    const int r_end = min(M, r_begin + 64);
    const int c = blockIdx.y * NT + threadIdx.x;
    if (c >= N) return;
    float s = 0.f;
    for (int r0 = r_begin; r0 < r_end; r0 += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[(int64_t)min(r0 + k, r_end - 1) * N + c];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += (r0 + k < r_end) ? v[k] : 0.f;
    }
    parts[(int64_t)blockIdx.x * N + c] = s;
}

// out[c] = sum_r src[r][c], rows in index order, one thread per column (stand-alone primitive: small problems)
__global__ __launch_bounds__(NT) void colsum_f32_kernel(const float* __restrict__ src, int M, int N,
                                                        float* __restrict__ out) {
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= N) return;
    float s = 0.f;
    for (int r = 0; r < M; ++r) s += src[(int64_t)r * N + c];
    out[c] = s;
}

// Second stage of every bias gradient (and the deterministic replacement of round 1's float atomics): job j's
// out[c] = parts[0][c] + parts[1][c] + ... in row order; optionally sum out^2 for clip_grad_norm_.
__global__ __launch_bounds__(NT) void bias_finish_kernel(BiasFinishJobs jobs, double* __restrict__ sumsq, LossFinish lf) {
    __shared__ float red[4];
    if (lf.scalars != nullptr && blockIdx.x == gridDim.x - 1) {          // the extra block: metric sums of the step
        __shared__ double lred[2][NT];
        double a = 0.0, b = 0.0;
        for (int i = threadIdx.x; i < lf.n_parts; i += NT) { a += lf.parts[2 * i]; b += lf.parts[2 * i + 1]; }
        lred[0][threadIdx.x] = a; lred[1][threadIdx.x] = b;
        __syncthreads();
        for (int o = NT / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) { lred[0][threadIdx.x] += lred[0][threadIdx.x + o]; lred[1][threadIdx.x] += lred[1][threadIdx.x + o]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            lf.scalars[CODAE_S_SQ_FULL] += lred[0][0];
            lf.scalars[CODAE_S_SQ_PARTIAL] += lred[1][0];
            lf.scalars[CODAE_S_LAST_LOSS] = lred[0][0] * lf.inv_n;
            lf.scalars[CODAE_S_STEP_SQ] = 0.0;
        }
        return;
    }
    const int gc = blockIdx.x * NT + threadIdx.x;            // column in the concatenation of all jobs
    float sq = 0.f;
    if (gc < jobs.col_begin[jobs.n]) {
        int j = 0;
        while (j + 1 < jobs.n && gc >= jobs.col_begin[j + 1]) ++j;
        const int c = gc - jobs.col_begin[j];
        const int N = jobs.cols[j], R = jobs.rows[j];
        const float* p = jobs.parts[j] + c;
        float s = 0.f;
        int r = 0;
        for (; r + 32 <= R; r += 32) {              // 32 independent loads in flight (C3: all of a column's rows in ONE trip to
            float v[32];                             // memory; 8 at a time was four dependent trips, 9.5 us), added in row order
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = p[(int64_t)(r + u) * N];
#pragma unroll
            for (int u = 0; u < 32; ++u) s += v[u];
        }
        for (; r + 8 <= R; r += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(r + u) * N];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; r < R; ++r) s += p[(int64_t)r * N];
        jobs.out[j][c] = s;
        sq = s * s;
    }
    if (sumsq != nullptr) {
        const float bsq = block_sum(sq, red);
        if (threadIdx.x == 0)
            atomicAdd(sumsq + (CODAE_S_GRAD_SQ_SLOTS - CODAE_S_GRAD_SQ) + (blockIdx.x & (CODAE_S_N_SLOTS - 1)), (double)bsq);
    }
}

// dst[c][r] = src[r][c] for up to 64 bf16 matrices in one launch (the transposed weight shadows);
// 64 x 64 tiles through LDS, 16-byte global accesses on both sides
struct TransposeJobs {
    int n;
    int64_t off[64];        // element offset of the matrix in src and dst
    int rows[64], cols[64]; // src is [rows][cols]
    int tile_begin[65];     // prefix sum of tiles per matrix
};
__global__ __launch_bounds__(NT) void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                            TransposeJobs jobs) {
    __shared__ bf16_t tile[64][72];          // 72: rows stay 16-byte aligned, banks staggered
    int m = 0;
    while (m + 1 < jobs.n && (int)blockIdx.x >= jobs.tile_begin[m + 1]) ++m;
    const int t = blockIdx.x - jobs.tile_begin[m];
    const int R = jobs.rows[m], C = jobs.cols[m];
    const int tiles_c = (C + 63) / 64;
    const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    const bf16_t* S = src + jobs.off[m];
    bf16_t* D = dst + jobs.off[m];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int q = threadIdx.x + p * NT;      // 512 chunks of 8 elements
        const int r = q >> 3, c8 = (q & 7) * 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (r0 + r < R && c0 + c8 < C) v = *reinterpret_cast<const uint4*>(S + (int64_t)(r0 + r) * C + c0 + c8);
        *reinterpret_cast<uint4*>(&tile[r][c8]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int q = threadIdx.x + p * NT;
        const int c = q >> 3, r8 = (q & 7) * 8;  // output row c (a source column), 8 source rows r8..r8+7
        if (c0 + c < C && r0 + r8 < R) {
            bf16_t e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = tile[r8 + k][c];
            uint4 v;
            v.x = (uint32_t)e[0] | ((uint32_t)e[1] << 16); v.y = (uint32_t)e[2] | ((uint32_t)e[3] << 16);
            v.z = (uint32_t)e[4] | ((uint32_t)e[5] << 16); v.w = (uint32_t)e[6] | ((uint32_t)e[7] << 16);
            *reinterpret_cast<uint4*>(D + (int64_t)(c0 + c) * R + r0 + r8) = v;
        }
    }
}

inline bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int launch_gather_corrupt(const codae_batch* b, void* out, int out_bf16, hipStream_t s, int64_t out_ld) {
    if (out_ld <= 0) out_ld = b ? b->io : 0;
    CODAE_REQUIRE(b && b->data && out && b->B > 0 && b->io > 0, "gather_corrupt: bad batch");
    const bool masked = b->mask_id || b->mask_to_use;
    CODAE_REQUIRE(!masked || b->mask_table, "gather_corrupt: mask ids without mask_table");
    CODAE_REQUIRE(!b->mask_to_use || b->mask_id || (b->nb_run > 0 && b->run >= 0 && b->run < b->nb_run),
                  "gather_corrupt: run %d outside [0, %d)", b->run, b->nb_run);
    const bool vec = (b->io % 4 == 0) && (out_ld % 4 == 0) && a16(b->data) && a16(out) && (!masked || (reinterpret_cast<uintptr_t>(b->mask_table) & 3) == 0);
    const int64_t items = (int64_t)b->B * (vec ? b->io / 4 : b->io);
    const int grid = grid_for(items);
    if (vec && out_bf16 && b->io % 8 == 0 && (!masked || (reinterpret_cast<uintptr_t>(b->mask_table) & 7) == 0)) {
        hipLaunchKernelGGL(gather_corrupt_bf16x8_kernel, dim3(grid_for(items / 2)), dim3(NT), 0, s, b->data, b->row_idx, b->mask_id,
                           b->mask_table, b->B, b->io, reinterpret_cast<bf16_t*>(out), b->mask_to_use, b->nb_run, b->run, out_ld);
        CODAE_LAUNCH_CHECK();
        return CODAE_OK;
    }
#define GC(V, O) hipLaunchKernelGGL((gather_corrupt_kernel<V, O>), dim3(grid), dim3(NT), 0, s, b->data, b->row_idx, \
                                    b->mask_id, b->mask_table, b->B, b->io, out, b->mask_to_use, b->nb_run, b->run, out_ld)
    if (vec && out_bf16) GC(true, true);
    else if (vec) GC(true, false);
    else if (out_bf16) GC(false, true);
    else GC(false, false);
#undef GC
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int launch_cast_bf16(const float* src, bf16_t* dst, int64_t n, hipStream_t s) {
    CODAE_REQUIRE(src && dst && n > 0 && a16(src) && (reinterpret_cast<uintptr_t>(dst) & 7) == 0, "cast: bad args");
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n / 4 + 1)), dim3(NT), 0, s, src, dst, n);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int launch_cast_f32(const bf16_t* src, float* dst, int64_t n, hipStream_t s) {
    CODAE_REQUIRE(src && dst && n > 0, "cast: bad args");
    hipLaunchKernelGGL(cast_f32_kernel, dim3(grid_for(n)), dim3(NT), 0, s, src, dst, n);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int launch_corrupt(const float* x, const float* mask, float* out, int64_t n, hipStream_t s) {
    CODAE_REQUIRE(x && mask && out && n > 0, "corrupt: bad args");
    hipLaunchKernelGGL(corrupt_kernel, dim3(grid_for(n)), dim3(NT), 0, s, x, mask, out, n);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int launch_expand_masks(const int32_t* mask_id, const uint8_t* table, const int32_t* k_of_mask, int B, int io,
                        int k_max, float* masks_out, float* fmask_out, hipStream_t s) {
    CODAE_REQUIRE(mask_id && table && k_of_mask && masks_out && fmask_out && B > 0 && io > 0 && k_max > 0,
                  "expand_masks: bad args");
    hipLaunchKernelGGL(expand_masks_kernel, dim3(grid_for((int64_t)B * io)), dim3(NT), 0, s, mask_id, table,
                       k_of_mask, B, io, k_max, masks_out, fmask_out);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int mse_loss_colsum_rows(int B) { return (B + LOSS_ROWS - 1) / LOSS_ROWS; }

int launch_mse_loss(const codae_batch* b, const float* y, void* dy, int dy_bf16, float inv_n, float* colsum_part,
                    double* loss_parts, int want_grad, hipStream_t s, int64_t dy_ld) {
    if (dy_ld <= 0) dy_ld = b ? b->io : 0;
    CODAE_REQUIRE(b && b->data && y && loss_parts && b->B > 0 && b->io > 0, "mse_loss: bad args");
    CODAE_REQUIRE(!want_grad || dy, "mse_loss: gradient requested without dy");
    const bool masked = b->mask_id || b->mask_to_use;
    CODAE_REQUIRE(!masked || b->mask_table, "mse_loss: mask ids without mask_table");
    const bool vec = (b->io % 4 == 0) && (dy_ld % 4 == 0) && a16(b->data) && a16(y) && (!dy || a16(dy)) &&
                     (!masked || (reinterpret_cast<uintptr_t>(b->mask_table) & 3) == 0);
    const int grid = (b->B + LOSS_ROWS - 1) / LOSS_ROWS;
#define ML(V, O) hipLaunchKernelGGL((mse_loss_kernel<V, O>), dim3(grid), dim3(NT), 0, s, b->data, b->row_idx, \
                                    b->mask_id, b->mask_table, b->B, b->io, y, dy, inv_n, colsum_part, loss_parts, want_grad, \
                                    b->mask_to_use, b->nb_run, b->run, dy_ld)
    if (vec && dy_bf16) ML(true, true);
    else if (vec) ML(true, false);
    else if (dy_bf16) ML(false, true);
    else ML(false, false);
#undef ML
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int launch_mse_dense(const float* x, const float* y, const float* fmask, float* dy, int64_t n, float inv_n,
                     double* scalars, hipStream_t s) {
    CODAE_REQUIRE(x && y && scalars && n > 0, "mse_dense: bad args");
    hipLaunchKernelGGL(mse_dense_kernel, dim3(grid_for(n)), dim3(NT), 0, s, x, y, fmask, dy, n, inv_n, scalars);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int launch_finish_loss(double* scalars, double inv_n, hipStream_t s, const double* parts, int n_parts) {
    hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(NT), 0, s, scalars, inv_n, parts, parts ? n_parts : 0);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int launch_sumsq(const float* g, int64_t n, double* out, hipStream_t s) {
    CODAE_REQUIRE(g && out && n > 0 && a16(g), "sumsq: bad args");
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n / 4 + 1)), dim3(NT), 0, s, g, n, out);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

// Tiled form of clip_adam for the bf16 engine: the weight matrices are walked in 64 x 128 tiles so that the same
// pass can also write the TRANSPOSED bf16 shadow (Wt_l [in][out], what the data-gradient GEMM reads) through LDS;
// the blocks past the last tile do the flat bias block.  Replaces clip_adam_kernel + transpose_bf16_kernel: 85 MB
// less traffic per step at C3 and no cross-stream hand-off for the transposes.
struct AdamTiles {
    int n_layers;
    int64_t off[64];          // element offset of W_l in the flat vectors (same in the shadows)
    int rows[64], cols[64];   // W_l is [rows = out][cols = in]
    int tile_begin[65];       // prefix sum of 64 x 128 tiles per layer
    int transposed_from;      // layers >= this also get the transposed shadow
    int64_t bias_off, bias_n; // flat tail
};
constexpr int AT_R = 64, AT_C = 128;

__global__ __launch_bounds__(NT) void clip_adam_tiled_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                             float* __restrict__ m, float* __restrict__ v, AdamConst c,
                                                             const double* __restrict__ grad_sq, bf16_t* __restrict__ shadow,
                                                             bf16_t* __restrict__ shadow_t, AdamTiles jobs,
                                                             const double* __restrict__ step_dev) {
    __shared__ bf16_t tt[AT_C][AT_R + 8];          // transposed tile: [col][row], rows stay 16-byte aligned
    __shared__ double total_sq;
    if (step_dev != nullptr) {
        const double t = *step_dev;
        c.lr_over_bc1 = (float)((double)c.lr / (1.0 - pow((double)c.beta1, t)));
        c.inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)c.beta2, t)));
    }
    float coef = 1.f;
    if (c.max_norm > 0.f) {
        if (threadIdx.x < 64) {
            double sv = grad_sq[(CODAE_S_GRAD_SQ_SLOTS - CODAE_S_GRAD_SQ) + threadIdx.x];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sv += __shfl_xor(sv, o);
            if (threadIdx.x == 0) total_sq = sv + grad_sq[0];
        }
        __syncthreads();
        coef = fminf(1.f, c.max_norm / (sqrtf((float)total_sq) + 1e-6f));
    }
    const int n_tiles = jobs.tile_begin[jobs.n_layers];
    if ((int)blockIdx.x >= n_tiles) {
        // flat bias block, grid-strided over the remaining blocks
        const int64_t nb = (int64_t)gridDim.x - n_tiles;
        for (int64_t e = ((int64_t)blockIdx.x - n_tiles) * NT + threadIdx.x; e < jobs.bias_n; e += nb * NT) {
            const int64_t i = jobs.bias_off + e;
            float pp = p[i], mm = m[i], vv = v[i];
            adam_one(pp, g[i], mm, vv, coef, c);
            p[i] = pp; m[i] = mm; v[i] = vv;
            shadow[i] = f32_to_bf16(pp);
        }
        return;
    }
    int l = 0;
    while (l + 1 < jobs.n_layers && (int)blockIdx.x >= jobs.tile_begin[l + 1]) ++l;
    const int t = blockIdx.x - jobs.tile_begin[l];
    const int R = jobs.rows[l], C = jobs.cols[l];
    const int tiles_c = (C + AT_C - 1) / AT_C;
    const int r0 = (t / tiles_c) * AT_R, c0 = (t % tiles_c) * AT_C;
    const int64_t base = jobs.off[l];
    const bool want_t = shadow_t != nullptr && l >= jobs.transposed_from;
#pragma unroll
    for (int it = 0; it < AT_R * AT_C / 4 / NT; ++it) {
        const int q = threadIdx.x + it * NT;          // float4 index inside the tile
        const int r = q / (AT_C / 4), cc = (q % (AT_C / 4)) * 4;
        if (r0 + r < R && c0 + cc < C) {
            const int64_t e = base + (int64_t)(r0 + r) * C + c0 + cc;
            float4 pp = *reinterpret_cast<float4*>(p + e);
            const float4 gg = *reinterpret_cast<const float4*>(g + e);
            float4 mm = *reinterpret_cast<float4*>(m + e);
            float4 vv = *reinterpret_cast<float4*>(v + e);
            adam_one(pp.x, gg.x, mm.x, vv.x, coef, c);
            adam_one(pp.y, gg.y, mm.y, vv.y, coef, c);
            adam_one(pp.z, gg.z, mm.z, vv.z, coef, c);
            adam_one(pp.w, gg.w, mm.w, vv.w, coef, c);
            *reinterpret_cast<float4*>(p + e) = pp;
            *reinterpret_cast<float4*>(m + e) = mm;
            *reinterpret_cast<float4*>(v + e) = vv;
            const uint2 sh = pack_bf16x4(pp.x, pp.y, pp.z, pp.w);
            *reinterpret_cast<uint2*>(shadow + e) = sh;
            if (want_t) {
                tt[cc + 0][r] = (bf16_t)(sh.x & 0xffff); tt[cc + 1][r] = (bf16_t)(sh.x >> 16);
                tt[cc + 2][r] = (bf16_t)(sh.y & 0xffff); tt[cc + 3][r] = (bf16_t)(sh.y >> 16);
            }
        }
    }
    if (!want_t) return;
    __syncthreads();
    // Wt[c0 + c][r0 .. r0 + 63]: 128-B row segments, 16 B per lane
#pragma unroll
    for (int it = 0; it < AT_C * (AT_R / 8) / NT; ++it) {
        const int q = threadIdx.x + it * NT;
        const int cidx = q / (AT_R / 8), r8 = (q % (AT_R / 8)) * 8;
        if (c0 + cidx < C && r0 + r8 < R)
            *reinterpret_cast<uint4*>(shadow_t + base + (int64_t)(c0 + cidx) * R + r0 + r8) =
                *reinterpret_cast<const uint4*>(&tt[cidx][r8]);
    }
}

__global__ void set_scalar_kernel(double* dst, double value) { *dst = value; }

int launch_set_scalar(double* dst, double value, hipStream_t s) {
    CODAE_REQUIRE(dst != nullptr, "set_scalar: null destination");
    hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, s, dst, value);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int launch_clip_adam(float* p, float* g, float* m, float* v, int64_t n, const codae_hyper* hp,
                     const double* grad_sq, bf16_t* shadow, const double* coef_in, hipStream_t s,
                     const double* step_dev) {
    CODAE_REQUIRE(p && g && m && v && hp && n > 0, "clip_adam: bad args");
    CODAE_REQUIRE(a16(p) && a16(g) && a16(m) && a16(v), "clip_adam: buffers must be 16-byte aligned");
    CODAE_REQUIRE(hp->step >= 1, "clip_adam: step must be >= 1");
    CODAE_REQUIRE(hp->max_grad_norm <= 0.f || grad_sq || coef_in, "clip_adam: clipping needs the grad_sq scalar");
    AdamConst c;
    const double bc1 = 1.0 - pow((double)hp->beta1, (double)hp->step);
    const double bc2 = 1.0 - pow((double)hp->beta2, (double)hp->step);
    c.lr_over_bc1 = (float)((double)hp->lr / bc1);
    c.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    c.beta1 = hp->beta1; c.beta2 = hp->beta2; c.eps = hp->eps; c.wd = hp->weight_decay;
    c.max_norm = hp->max_grad_norm; c.lr = hp->lr;
    hipLaunchKernelGGL(clip_adam_kernel, dim3(grid_for(n / 4 + 1)), dim3(NT), 0, s, p, g, m, v, n, c, grad_sq, shadow, coef_in,
                       step_dev);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int launch_clip_adam_tiled(float* p, float* g, float* m, float* v, const codae_hyper* hp, const double* grad_sq,
                           bf16_t* shadow, bf16_t* shadow_t, int n_layers, const int64_t* w_off, const int* rows,
                           const int* cols, int transposed_from, int64_t bias_off, int64_t bias_n, hipStream_t s,
                           const double* step_dev) {
    CODAE_REQUIRE(p && g && m && v && hp && shadow && n_layers > 0 && n_layers <= 64, "clip_adam_tiled: bad args");
    CODAE_REQUIRE(a16(p) && a16(g) && a16(m) && a16(v), "clip_adam_tiled: buffers must be 16-byte aligned");
    CODAE_REQUIRE(hp->step >= 1, "clip_adam_tiled: step must be >= 1");
    CODAE_REQUIRE(hp->max_grad_norm <= 0.f || grad_sq, "clip_adam_tiled: clipping needs the grad_sq scalar");
    AdamConst c;
    const double bc1 = 1.0 - pow((double)hp->beta1, (double)hp->step);
    const double bc2 = 1.0 - pow((double)hp->beta2, (double)hp->step);
    c.lr_over_bc1 = (float)((double)hp->lr / bc1);
    c.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    c.beta1 = hp->beta1; c.beta2 = hp->beta2; c.eps = hp->eps; c.wd = hp->weight_decay;
    c.max_norm = hp->max_grad_norm; c.lr = hp->lr;
    AdamTiles jobs;
    jobs.n_layers = n_layers; jobs.transposed_from = transposed_from; jobs.bias_off = bias_off; jobs.bias_n = bias_n;
    int total = 0;
    for (int l = 0; l < n_layers; ++l) {
        CODAE_REQUIRE(rows[l] % 8 == 0 && cols[l] % 4 == 0 && w_off[l] % 8 == 0, "clip_adam_tiled: layer %d shape %d x %d", l, rows[l], cols[l]);
        jobs.off[l] = w_off[l]; jobs.rows[l] = rows[l]; jobs.cols[l] = cols[l];
        jobs.tile_begin[l] = total;
        total += ((rows[l] + AT_R - 1) / AT_R) * ((cols[l] + AT_C - 1) / AT_C);
    }
    jobs.tile_begin[n_layers] = total;
    const int bias_blocks = (int)((bias_n + NT * 8 - 1) / (NT * 8)) > 0 ? (int)((bias_n + NT * 8 - 1) / (NT * 8)) : 1;
    hipLaunchKernelGGL(clip_adam_tiled_kernel, dim3(total + bias_blocks), dim3(NT), 0, s, p, g, m, v, c, grad_sq, shadow, shadow_t,
                       jobs, step_dev);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int launch_sumsq_to(const float* g, int64_t n, double* acc, hipStream_t s) {
    CODAE_REQUIRE(g && acc && n > 0, "sumsq_to: bad args");
    int grid = grid_for(n / 4 + 1);
    if (grid > 64) grid = 64;
    hipLaunchKernelGGL(sumsq_to_kernel, dim3(grid), dim3(NT), 0, s, g, n, acc);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int launch_clip_coef(const double* total_sq, float max_norm, double* coef_out, hipStream_t s) {
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, s, total_sq, max_norm, coef_out);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int launch_colsum_f32(const float* src, int M, int N, float* out, hipStream_t s) {
    CODAE_REQUIRE(src && out && M > 0 && N > 0, "colsum: bad args");
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((N + NT - 1) / NT), dim3(NT), 0, s, src, M, N, out);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int launch_colsum_parts_f32(const float* src, int M, int N, float* parts, hipStream_t s) {
    CODAE_REQUIRE(src && parts && M > 0 && N > 0, "colsum_parts: bad args");
    hipLaunchKernelGGL(colsum_parts_f32_kernel, dim3((M + 63) / 64, (N + NT - 1) / NT), dim3(NT), 0, s, src, M, N, parts);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int launch_bias_finish(const BiasFinishJobs& jobs, double* sumsq, hipStream_t s, const LossFinish* loss) {
    CODAE_REQUIRE(jobs.n > 0 && jobs.n <= 64, "bias_finish: %d jobs", jobs.n);
    const int total = jobs.col_begin[jobs.n];
    CODAE_REQUIRE(total > 0, "bias_finish: no columns");
    LossFinish lf{};
    if (loss != nullptr) { lf = *loss; CODAE_REQUIRE(lf.scalars && lf.parts && lf.n_parts > 0, "bias_finish: bad loss block"); }
    hipLaunchKernelGGL(bias_finish_kernel, dim3((total + NT - 1) / NT + (loss != nullptr ? 1 : 0)), dim3(NT), 0, s, jobs, sumsq, lf);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int launch_transpose_bf16(const bf16_t* src, bf16_t* dst, int n, const int64_t* off, const int* rows, const int* cols,
                          hipStream_t s) {
    CODAE_REQUIRE(src && dst && n > 0 && n <= 64, "transpose: bad args");
    TransposeJobs jobs;
    jobs.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        CODAE_REQUIRE(rows[i] % 8 == 0 && cols[i] % 8 == 0, "transpose: matrix %d is %d x %d (need multiples of 8)", i, rows[i], cols[i]);
        jobs.off[i] = off[i]; jobs.rows[i] = rows[i]; jobs.cols[i] = cols[i];
        jobs.tile_begin[i] = total;
        total += ((rows[i] + 63) / 64) * ((cols[i] + 63) / 64);
    }
    jobs.tile_begin[n] = total;
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(total), dim3(NT), 0, s, src, dst, jobs);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int launch_reduce_slabs(const float* slabs, int n_slabs, int64_t stride, float* out, int64_t n, double* sumsq,
                        hipStream_t s) {
    CODAE_REQUIRE(slabs && out && n_slabs >= 1 && n > 0, "reduce_slabs: bad args");
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid_for(n / 4 + 1)), dim3(NT), 0, s, slabs, n_slabs, stride, out, n, sumsq);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int launch_reduce_slabs_epi(const float* slabs, int n_slabs, int64_t stride, int M, int N, float* C, int64_t ldc, const float* bias,
                            int relu, const float* relu_src, int64_t ld_relu, float* colsum_part, hipStream_t s) {
    CODAE_REQUIRE(slabs && C && n_slabs >= 1 && M > 0 && N > 0 && N % 4 == 0 && ldc % 4 == 0 && (relu_src == nullptr || ld_relu % 4 == 0) &&
                      a16(slabs) && a16(C) && (bias == nullptr || a16(bias)) && (relu_src == nullptr || a16(relu_src)),
                  "reduce_slabs_epi: bad args");
    hipLaunchKernelGGL(reduce_slabs_epi_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(NT), 0, s, slabs, n_slabs, stride, M, N, C, ldc, bias,
                       relu, relu_src, ld_relu, colsum_part);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

}  // namespace codae
