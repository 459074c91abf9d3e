// "Next" rows of the scope table: the abalone training loss
// (CombinedCriterion, codae/tool/metering.py:82-180 of the reference) and the validation rank
// metric (RankingLoss, metering.py:29-79) as HIP kernels.  Small, HBM/latency-bound work: one pass
// per kernel, no GEMM reshaping.
#include "codae_common.h"

namespace codae {
namespace {

constexpr int NT = 256;

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// log-softmax of y[p .. p+s) evaluated at index t (numerically as torch.log_softmax)
__device__ __forceinline__ float log_softmax_at(const float* __restrict__ yr, int p, int s, int t) {
    float m = yr[p];
    for (int k = 1; k < s; ++k) m = fmaxf(m, yr[p + k]);
    float se = 0.f;
    for (int k = 0; k < s; ++k) se += expf(yr[p + k] - m);
    return (yr[p + t] - m) - logf(se);
}
__device__ __forceinline__ int argmax_first(const float* __restrict__ xr, int p, int s) {
    int t = 0;
    float best = xr[p];
    for (int k = 1; k < s; ++k)
        if (xr[p + k] > best) { best = xr[p + k]; t = k; }
    return t;
}

// pass 1: acc[v] = sum over the batch of  sum_s (x-y)^2   (regression)
//                                        -log_softmax(y)[argmax x]   (classification)
__global__ __launch_bounds__(NT) void combined_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             int B, int io, int nv, const int32_t* __restrict__ pos,
                                                             const int32_t* __restrict__ size,
                                                             const int32_t* __restrict__ type, double* __restrict__ acc) {
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < (int64_t)B * nv; e += (int64_t)gridDim.x * NT) {
        const int b = (int)(e / nv), v = (int)(e - (int64_t)b * nv);
        const float* xr = x + (int64_t)b * io;
        const float* yr = y + (int64_t)b * io;
        const int p = pos[v], s = size[v];
        float val = 0.f;
        if (type[v] == 0) {
            for (int k = 0; k < s; ++k) { const float d = xr[p + k] - yr[p + k]; val += d * d; }
        } else {
            val = -log_softmax_at(yr, p, s, argmax_first(xr, p, s));
        }
        atomicAdd(&acc[v], (double)val);
    }
}

// pass 2: loss = (1/nv) sum_v w_v L_v, L_v = sqrt(acc_v / (B s)) or acc_v / B; dy = dloss/dy
__global__ __launch_bounds__(NT) void combined_grad_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           int B, int io, int nv, const int32_t* __restrict__ pos,
                                                           const int32_t* __restrict__ size,
                                                           const int32_t* __restrict__ type,
                                                           const float* __restrict__ weight,
                                                           const double* __restrict__ acc, float* __restrict__ dy,
                                                           double* __restrict__ loss_out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double loss = 0.0;
        for (int v = 0; v < nv; ++v) {
            const double lv = type[v] == 0 ? sqrt(acc[v] / ((double)B * size[v])) : acc[v] / (double)B;
            loss += (double)weight[v] * lv;
        }
        *loss_out = loss / nv;
    }
    if (dy == nullptr) return;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < (int64_t)B * nv; e += (int64_t)gridDim.x * NT) {
        const int b = (int)(e / nv), v = (int)(e - (int64_t)b * nv);
        const float* xr = x + (int64_t)b * io;
        const float* yr = y + (int64_t)b * io;
        float* gr = dy + (int64_t)b * io;
        const int p = pos[v], s = size[v];
        const float c = weight[v] / (float)nv;
        if (type[v] == 0) {
            const float rmse = (float)sqrt(acc[v] / ((double)B * s));
            const float k0 = c / ((float)B * (float)s * rmse);
            for (int k = 0; k < s; ++k) gr[p + k] = k0 * (yr[p + k] - xr[p + k]);
        } else {
            const int t = argmax_first(xr, p, s);
            float m = yr[p];
            for (int k = 1; k < s; ++k) m = fmaxf(m, yr[p + k]);
            float se = 0.f;
            for (int k = 0; k < s; ++k) se += expf(yr[p + k] - m);
            for (int k = 0; k < s; ++k) gr[p + k] = c * (expf(yr[p + k] - m) / se - (k == t ? 1.f : 0.f)) / (float)B;
        }
    }
}

// monitor criterion (reduction "none", metering.py:131-152): out[b][v] = (x-y)^2 (size-1 regression) | NLL
__global__ __launch_bounds__(NT) void combined_full_kernel(const float* __restrict__ x, const float* __restrict__ y, int B,
                                                           int io, int nv, const int32_t* __restrict__ pos,
                                                           const int32_t* __restrict__ size,
                                                           const int32_t* __restrict__ type, float* __restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < (int64_t)B * nv; e += (int64_t)gridDim.x * NT) {
        const int b = (int)(e / nv), v = (int)(e - (int64_t)b * nv);
        const float* xr = x + (int64_t)b * io;
        const float* yr = y + (int64_t)b * io;
        const int p = pos[v], s = size[v];
        float val;
        if (type[v] == 0) { const float d = xr[p] - yr[p]; val = d * d; }
        else val = -log_softmax_at(yr, p, s, argmax_first(xr, p, s));
        out[e] = val;
    }
}

// Per-step accounting of the abalone sweep (train_dae_on_abalone.py:227-236): ONE workgroup; thread (r, v) walks rows r, r + R,
// ... of variable v and keeps {f_k, p_k}[k] for its variable in LDS (its own slots: no atomics); thread v then adds the R
// partials of its variable in row-lane order and folds them into the fp64 tables - every addition in a fixed order.
constexpr int MON_KMAX = 16;
__global__ __launch_bounds__(NT) void monitor_accumulate_kernel(const float* __restrict__ x, const float* __restrict__ y, int B, int io, int nv,
                                                                const int32_t* __restrict__ pos, const int32_t* __restrict__ size,
                                                                const int32_t* __restrict__ type, const float* __restrict__ us,
                                                                const float* __restrict__ um, const int32_t* __restrict__ mask_id,
                                                                const uint8_t* __restrict__ table, const int32_t* __restrict__ k_of_mask,
                                                                int k_max, double* __restrict__ acc) {
    extern __shared__ double part[];                     // [NT][2 * k_max]
    const int R = NT / nv;                               // row lanes (host checks nv <= NT)
    const int r = threadIdx.x / nv, v = threadIdx.x - r * nv;
    double* mine = part + (size_t)threadIdx.x * 2 * k_max;
    for (int k = 0; k < 2 * k_max; ++k) mine[k] = 0.0;
    if (r < R) {
        const int p = pos[v], s = size[v], t = type[v];
        for (int b = r; b < B; b += R) {
            const float* xr = x + (int64_t)b * io;
            const float* yr = y + (int64_t)b * io;
            float val;
            if (t == 0) {
                const float sc = us ? us[p] : 1.f, mn = um ? um[p] : 0.f;
                const float xu = xr[p] * sc + mn, yu = yr[p] * sc + mn;      // (data * scale) + min, fp32, as torch
                const float d = xu - yu;
                val = d * d;
            } else {
                val = -log_softmax_at(yr, p, s, argmax_first(xr, p, s));     // (one-hot columns are not de-normalised)
            }
            const int id = mask_id[b];
            const int k = k_of_mask[id] - 1;
            if (k >= 0 && k < k_max) {
                mine[k] += (double)val;
                if (table[(int64_t)id * io + p] == 0) mine[k_max + k] += (double)val;
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nv) {
        for (int k = 0; k < 2 * k_max; ++k) {
            double t = 0.0;
            for (int rr = 0; rr < R; ++rr) t += part[(size_t)(rr * nv + threadIdx.x) * 2 * k_max + k];
            acc[2 + (int64_t)k * nv + threadIdx.x] += t;
            part[(size_t)threadIdx.x * 2 * k_max + k] = t;       // (slot of row lane 0: read back by thread 0 below)
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double f = 0.0, pp = 0.0;
        for (int k = 0; k < k_max; ++k)
            for (int vv = 0; vv < nv; ++vv) { f += part[(size_t)vv * 2 * k_max + k]; pp += part[(size_t)vv * 2 * k_max + k_max + k]; }
        acc[0] += f; acc[1] += pp;
    }
}

// ||row||_2 of a [rows][E] matrix, one wave per row
__global__ __launch_bounds__(NT) void row_norm_kernel(const float* __restrict__ m, int64_t rows, int E, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const float* p = m + r * E;
        float s = 0.f;
        for (int k = lane; k < E; k += 64) s += p[k] * p[k];
        s = wave_sum_f(s);
        if (lane == 0) out[r] = sqrtf(s);
    }
}

// RankingLoss.get (metering.py:46-79): one workgroup per batch sample.
//   c     = blanked slot = sum_s s * (1 - fmask[b][s*E])                       (metering.py:56)
//   s_j   = cos(inventory[c][j], prediction[b][cE:(c+1)E])                      (:66-68)
//   rank  = #{ j in validation : s[idx_b] > s_j }                               (:72-75)
//   out  += 1 - rank / (V - 1)                                                  (:77)
__global__ __launch_bounds__(NT) void ranking_kernel(const float* __restrict__ pred, const float* __restrict__ fmask,
                                                     const int32_t* __restrict__ idx, int io, int S, int E,
                                                     const float* __restrict__ inv, const float* __restrict__ inv_norm,
                                                     int64_t n_obs, const int32_t* __restrict__ val, int V,
                                                     double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float q[];   // [E] + 8 scratch
    __shared__ float red[8];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int c = 0;
    for (int s = 1; s < S; ++s) c += (fmask[(int64_t)b * io + (int64_t)s * E] == 0.f) ? s : 0;
    const float* pq = pred + (int64_t)b * io + (int64_t)c * E;
    float sq = 0.f;
    for (int k = threadIdx.x; k < E; k += NT) { const float v = pq[k]; q[k] = v; sq += v * v; }
    sq = wave_sum_f(sq);
    if (lane == 0) red[w] = sq;
    __syncthreads();
    const float qn = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-8f);
    const float* invc = inv + (int64_t)c * n_obs * E;
    const float* nrmc = inv_norm + (int64_t)c * n_obs;
    // own similarity (every wave computes it: cheaper than another barrier)
    const int64_t me = idx[b];
    float d = 0.f;
    for (int k = lane; k < E; k += 64) d += q[k] * invc[me * E + k];
    d = wave_sum_f(d);
    const float own = d / (qn * fmaxf(nrmc[me], 1e-8f));
    int rank = 0;
    for (int j = w; j < V; j += 4) {
        const int64_t r = val[j];
        const float* pr = invc + r * E;
        float dj = 0.f;
        for (int k = lane; k < E; k += 64) dj += q[k] * pr[k];
        dj = wave_sum_f(dj);
        const float sj = dj / (qn * fmaxf(nrmc[r], 1e-8f));
        rank += (own > sj) ? 1 : 0;
    }
    __syncthreads();
    if (lane == 0) red[4 + w] = (float)rank;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double total = (double)red[4] + red[5] + red[6] + red[7];
        atomicAdd(out, 1.0 - total / (double)(V - 1));
    }
}


// ---- RankingLoss as a GEMM --------------------------------------------------------------------------
// The wave-per-item kernel above reads the whole validation inventory once PER SAMPLE (V x E floats: 53 MB at C3's
// shape, 1.4 TB per validation epoch - 20x the training epoch).  Here the similarities of a whole batch against a chunk
// of the validation inventory are one exact-fp32 MFMA GEMM per slot (gemm_f32.hip: pred[:, cE:(c+1)E] . inv_val_c^T),
// followed by a compare-count pass; the batch's mask ids come from the Corrupter's device tables, nothing goes
// through the host, and *out accumulates over the batches of an epoch.
//   rs[b] = {slot c, |q| clamped, own similarity, rank so far}
struct RankRow { int slot; float qn; float own; int rank; int self_col; int pad[3]; };      // pad[1]: duplicate group of the own row or -1

// one wave per sample: blanked slot from the mask table (metering.py:56: sum of the blanked slots' indices), |q|, s[idx_b]
__global__ __launch_bounds__(NT) void rank_prep_kernel(const float* __restrict__ pred, const int32_t* __restrict__ row_idx,
                                                       const int32_t* __restrict__ mask_id, const int32_t* __restrict__ mask_to_use,
                                                       int nb_run, int run, const uint8_t* __restrict__ mask_table, int B, int io, int S,
                                                       int E, const float* __restrict__ inv, const float* __restrict__ inv_norm,
                                                       int64_t n_obs, const int32_t* __restrict__ val_pos, const int32_t* __restrict__ val_group,
                                                       int n_val, RankRow* __restrict__ rs, int32_t* __restrict__ perm,
                                                       int32_t* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (b >= B) return;
    const int64_t me = row_idx[b];
    const int id = mask_id ? mask_id[b] : mask_to_use[me * nb_run + run];
    int c = 0;
    for (int s = 1; s < S; ++s) c += (mask_table[(int64_t)id * io + (int64_t)s * E] == 0) ? s : 0;
    c = c < S ? c : S - 1;                       // (k > 1 blanks several slots; the reference's formula is meant for k = 1)
    const float* q = pred + (int64_t)b * io + (int64_t)c * E;
    const float* r = inv + ((int64_t)c * n_obs + me) * E;
    float sq = 0.f, d = 0.f;
    for (int k = lane; k < E; k += 64) { const float v = q[k]; sq += v * v; d += v * r[k]; }
    sq = wave_sum_f(sq); d = wave_sum_f(d);
    if (lane == 0) {
        const float qn = fmaxf(sqrtf(sq), 1e-8f);
        RankRow o; o.slot = c; o.qn = qn; o.own = d / (qn * fmaxf(inv_norm[(int64_t)c * n_obs + me], 1e-8f)); o.rank = 0;
        // The sample's own row, when it is a validation row, is never counted by the reference: s[idx] > s[idx] compares a
        // value with itself.  Here `own` and the GEMM's column for that row are summed in different orders, so that
        // column is skipped by position instead.
        o.self_col = val_pos ? val_pos[me] : -1;
        // the rows of one slot, compacted (in arrival order: which GEMM row a sample lands on does not change its rank)
        const int k = atomicAdd(&counts[c], 1);
        perm[(int64_t)c * B + k] = b;
        // rows of the inventory that are exact copies of the sample's own row are never counted either (see codae_hip.h)
        o.pad[0] = k; o.pad[1] = (val_group && o.self_col >= 0) ? val_group[(int64_t)c * n_val + o.self_col] : -1; o.pad[2] = 0;
        rs[b] = o;
    }
}

// compact rows of slot c (k < counts[c]; sample perm[c][k]): rank += #{ j < n : own > dots[k][j] / (|q| |inv_j|) }; one wave per row
__global__ __launch_bounds__(NT) void rank_count_kernel(const float* __restrict__ dots, int64_t ld, int B, int n, int c, int v0,
                                                        const float* __restrict__ val_norm, const int32_t* __restrict__ group,
                                                        RankRow* __restrict__ rs, const int32_t* __restrict__ perm,
                                                        const int32_t* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (k >= counts[c]) return;
    const int b = perm[(int64_t)c * B + k];
    const RankRow me = rs[b];
    const float* d = dots + (int64_t)k * ld;
    int cnt = 0;
    const int skip = me.self_col - v0;
    const int grp = me.pad[1];          // (group: this chunk's slice of the slot's duplicate-group ids, or null)
    for (int j = lane; j < n; j += 64) {
        const bool same = (j == skip) || (group != nullptr && grp >= 0 && group[j] == grp);
        cnt += (!same && me.own > d[j] / (me.qn * fmaxf(val_norm[j], 1e-8f))) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) rs[b].rank = me.rank + cnt;
}

// q[k][:] = pred[perm[c][k]][cE:(c+1)E] for k < counts[c]: the slot's queries, contiguous (the GEMM's A operand)
__global__ __launch_bounds__(NT) void rank_gather_q_kernel(const float* __restrict__ pred, int io, int E, int B, int c,
                                                           const int32_t* __restrict__ perm, const int32_t* __restrict__ counts,
                                                           float* __restrict__ q) {
    const int n = counts[c];
    const int64_t total = (int64_t)n * E;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total; e += (int64_t)gridDim.x * NT) {
        const int k = (int)(e / E), x = (int)(e - (int64_t)k * E);
        q[e] = pred[(int64_t)perm[(int64_t)c * B + k] * io + (int64_t)c * E + x];
    }
}

// *out += sum_b 1 - rank_b / (V - 1), rows added in index order (one workgroup: the same bits every run)
__global__ __launch_bounds__(NT) void rank_finish_kernel(const RankRow* __restrict__ rs, int B, int V, double* __restrict__ out) {
    __shared__ double red[NT];
    double t = 0.0;
    for (int b = threadIdx.x; b < B; b += NT) t += 1.0 - (double)rs[b].rank / (double)(V - 1);
    red[threadIdx.x] = t;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out += red[0];
}

// rows val[j] of every slot's inventory, contiguous: dst[c][j][:] = inv[c][val[j]][:]
__global__ __launch_bounds__(NT) void gather_rows_kernel(const float* __restrict__ inv, int64_t n_obs, int E, int S,
                                                         const int32_t* __restrict__ val, int V, float* __restrict__ dst) {
    const int64_t total = (int64_t)S * V * E;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < total; e += (int64_t)gridDim.x * NT) {
        const int k = (int)(e % E);
        const int64_t cj = e / E;
        const int j = (int)(cj % V), c = (int)(cj / V);
        dst[e] = inv[((int64_t)c * n_obs + val[j]) * E + k];
    }
}

inline int grid_for(int64_t items) {
    int64_t b = (items + NT - 1) / NT;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

}  // namespace
}  // namespace codae

using namespace codae;

extern "C" {

int codae_combined_loss_fwd_bwd(const float* x, const float* y, int32_t B, int32_t io, int32_t n_var, const int32_t* var_pos,
                                const int32_t* var_size, const int32_t* var_type, const float* var_weight, double* acc,
                                float* dy, double* loss_out, void* stream) {
    CODAE_REQUIRE(x && y && var_pos && var_size && var_type && var_weight && acc && loss_out, "combined_loss: null argument");
    CODAE_REQUIRE(B > 0 && io > 0 && n_var > 0, "combined_loss: empty problem");
    hipStream_t s = (hipStream_t)stream;
    CODAE_HIP_CHECK(hipMemsetAsync(acc, 0, (size_t)n_var * sizeof(double), s));
    const int grid = grid_for((int64_t)B * n_var);
    hipLaunchKernelGGL(combined_reduce_kernel, dim3(grid), dim3(NT), 0, s, x, y, B, io, n_var, var_pos, var_size, var_type, acc);
    CODAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(combined_grad_kernel, dim3(grid), dim3(NT), 0, s, x, y, B, io, n_var, var_pos, var_size, var_type,
                       var_weight, acc, dy, loss_out);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int codae_combined_loss_full(const float* x, const float* y, int32_t B, int32_t io, int32_t n_var, const int32_t* var_pos,
                             const int32_t* var_size, const int32_t* var_type, float* out, void* stream) {
    CODAE_REQUIRE(x && y && var_pos && var_size && var_type && out && B > 0 && io > 0 && n_var > 0, "combined_full: bad argument");
    hipLaunchKernelGGL(combined_full_kernel, dim3(grid_for((int64_t)B * n_var)), dim3(NT), 0, (hipStream_t)stream, x, y, B, io,
                       n_var, var_pos, var_size, var_type, out);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int codae_monitor_accumulate(const float* x, const float* y, int32_t B, int32_t io, int32_t n_var, const int32_t* var_pos,
                             const int32_t* var_size, const int32_t* var_type, const float* undo_scale, const float* undo_min,
                             const int32_t* mask_id, const uint8_t* mask_table, const int32_t* k_of_mask, int32_t k_max,
                             double* acc, void* stream) {
    CODAE_REQUIRE(x && y && var_pos && var_size && var_type && mask_id && mask_table && k_of_mask && acc, "monitor_accumulate: null argument");
    CODAE_REQUIRE(B > 0 && io > 0 && n_var > 0 && n_var <= NT, "monitor_accumulate: bad sizes (B %d, io %d, n_var %d)", B, io, n_var);
    CODAE_REQUIRE(k_max >= 1 && k_max <= MON_KMAX, "monitor_accumulate: k_max %d outside [1, %d]", k_max, MON_KMAX);
    CODAE_REQUIRE((undo_scale == nullptr) == (undo_min == nullptr), "monitor_accumulate: undo_scale and undo_min come together");
    hipLaunchKernelGGL(monitor_accumulate_kernel, dim3(1), dim3(NT), (size_t)NT * 2 * k_max * sizeof(double), (hipStream_t)stream, x, y, B,
                       io, n_var, var_pos, var_size, var_type, undo_scale, undo_min, mask_id, mask_table, k_of_mask, k_max, acc);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int codae_row_norms(const float* m, int64_t rows, int32_t E, float* out, void* stream) {
    CODAE_REQUIRE(m && out && rows > 0 && E > 0, "row_norms: bad argument");
    int64_t g = (rows + 3) / 4;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(row_norm_kernel, dim3((unsigned)g), dim3(NT), 0, (hipStream_t)stream, m, rows, E, out);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int codae_ranking_loss(const float* pred, const float* fmask, const int32_t* idx, int32_t B, int32_t io, int32_t n_slots,
                       int32_t E, const float* inventory, const float* inventory_norm, int64_t n_obs, const int32_t* val_idx,
                       int32_t n_val, double* out, void* stream) {
    CODAE_REQUIRE(pred && fmask && idx && inventory && inventory_norm && val_idx && out, "ranking_loss: null argument");
    CODAE_REQUIRE(B > 0 && n_slots > 0 && E > 0 && io == n_slots * E && n_val > 1 && n_obs > 0, "ranking_loss: bad sizes");
    CODAE_REQUIRE((size_t)E * sizeof(float) <= 64 * 1024, "ranking_loss: embedding size %d too large for LDS staging", E);
    hipLaunchKernelGGL(ranking_kernel, dim3(B), dim3(NT), (size_t)E * sizeof(float), (hipStream_t)stream, pred, fmask, idx, io,
                       n_slots, E, inventory, inventory_norm, n_obs, val_idx, n_val, out);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int codae_gather_inventory_rows(const float* inventory, int64_t n_obs, int32_t E, int32_t n_slots, const int32_t* val_idx,
                                int32_t n_val, float* dst, void* stream) {
    CODAE_REQUIRE(inventory && val_idx && dst && n_obs > 0 && E > 0 && n_slots > 0 && n_val > 0, "gather_inventory_rows: bad argument");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((int64_t)n_slots * n_val * E)), dim3(NT), 0, (hipStream_t)stream, inventory,
                       n_obs, E, n_slots, val_idx, n_val, dst);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

int codae_ranking_loss_batched(const float* pred, int32_t B, int32_t io, int32_t n_slots, int32_t E, const int32_t* row_idx,
                               const int32_t* mask_id, const int32_t* mask_to_use, int32_t nb_run, int32_t run,
                               const uint8_t* mask_table, const float* inventory, const float* inventory_norm, int64_t n_obs,
                               const float* inv_val, const float* inv_val_norm, const int32_t* val_pos, const int32_t* val_group,
                               int32_t n_val, float* work, int32_t chunk, void* row_state, int32_t* perm_ws, float* q_ws, double* out,
                               void* stream) {
    CODAE_REQUIRE(pred && row_idx && (mask_id || mask_to_use) && mask_table && inventory && inventory_norm && inv_val && inv_val_norm &&
                      work && row_state && perm_ws && q_ws && out, "ranking_loss_batched: null argument");
    CODAE_REQUIRE(B > 0 && n_slots > 0 && E > 0 && io == n_slots * E && n_val > 1 && n_obs > 0 && chunk > 0, "ranking_loss_batched: bad sizes");
    CODAE_REQUIRE(mask_id || (nb_run > 0 && run >= 0 && run < nb_run), "ranking_loss_batched: run %d outside [0, %d)", run, nb_run);
    hipStream_t s = (hipStream_t)stream;
    RankRow* rs = reinterpret_cast<RankRow*>(row_state);
    int32_t* perm = perm_ws;                               // [n_slots][B]
    int32_t* counts = perm_ws + (int64_t)n_slots * B;      // [n_slots]
    CODAE_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)n_slots * sizeof(int32_t), s));
    const int rows_grid = (B + NT / 64 - 1) / (NT / 64);
    hipLaunchKernelGGL(rank_prep_kernel, dim3(rows_grid), dim3(NT), 0, s, pred, row_idx, mask_id, mask_to_use, nb_run, run, mask_table, B,
                       io, n_slots, E, inventory, inventory_norm, n_obs, val_pos, val_group, n_val, rs, perm, counts);
    CODAE_LAUNCH_CHECK();
    // per slot: its rows' queries gathered, then one GEMM per chunk of validation rows over THOSE rows only (the row count
    // stays on the device: the grid covers B rows, tiles past counts[c] return at once), then the compare-count pass
    for (int c = 0; c < n_slots; ++c) {
        hipLaunchKernelGGL(rank_gather_q_kernel, dim3(grid_for((int64_t)B * E / 4)), dim3(NT), 0, s, pred, io, E, B, c, perm, counts, q_ws);
        CODAE_LAUNCH_CHECK();
        for (int v0 = 0; v0 < n_val; v0 += chunk) {
            const int n = n_val - v0 < chunk ? n_val - v0 : chunk;
            GemmF32 g{};
            g.A = q_ws; g.a_rs = E; g.a_ks = 1;
            g.B = inv_val + ((int64_t)c * n_val + v0) * E; g.b_rs = E; g.b_ks = 1;
            g.C = work; g.ldc = chunk;
            g.M = B; g.N = n; g.K = E;
            g.m_dev = counts + c;
            int rc = gemm_f32(g, s);
            if (rc) return rc;
            hipLaunchKernelGGL(rank_count_kernel, dim3(rows_grid), dim3(NT), 0, s, work, (int64_t)chunk, B, n, c, v0,
                               inv_val_norm + (int64_t)c * n_val + v0, val_group ? val_group + (int64_t)c * n_val + v0 : nullptr, rs, perm,
                               counts);
            CODAE_LAUNCH_CHECK();
        }
    }
    hipLaunchKernelGGL(rank_finish_kernel, dim3(1), dim3(NT), 0, s, rs, B, n_val, out);
    CODAE_LAUNCH_CHECK();
    return CODAE_OK;
}

}  // extern "C"
