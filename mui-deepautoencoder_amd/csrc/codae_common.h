// Shared declarations for libcodae_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "codae_hip.h"

namespace codae {

void set_error(const char* fmt, ...);

#define CODAE_HIP_CHECK(expr)                                                              \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            codae::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return CODAE_E_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define CODAE_LAUNCH_CHECK()                                                               \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess) {                                                            \
            codae::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return CODAE_E_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define CODAE_REQUIRE(cond, ...)                                                           \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            codae::set_error(__VA_ARGS__);                                                 \
            return CODAE_E_INVALID;                                                        \
        }                                                                                  \
    } while (0)

typedef uint16_t bf16_t;  // raw bf16 storage

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even through the hardware convert (keeps NaN a NaN)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

// two fp32 -> one packed bf16 pair with a single v_cvt_pk_bf16_f32 (round-to-nearest-even)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// max(x, floor) as exactly one VALU op: fmaxf() puts a canonicalising v_max in front of the real one.
// The GEMM epilogues clamp at `floor` = 0 (ReLU) or -inf (identity), so the ReLU switch costs no branch.
__device__ __forceinline__ float clamp_below(float x, float floor) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(floor), "v"(x));
    return r;
}

// bias[j .. j+3] (or zeros when there is no bias / j is past N) as ONE unconditional 16-B load from a clamped address:
// with a branch around it the compiler waits for each of an epilogue's 6 bias loads before issuing the next one
// (~0.5 us of L2 latency apiece).  `valid`: any readable 16-B aligned address.
__device__ __forceinline__ float4 load_bias4(const float* __restrict__ bias, const void* valid, int j, int N) {
    const bool ok = bias != nullptr && j < N;
    const float4 v = *reinterpret_cast<const float4*>(ok ? bias + j : reinterpret_cast<const float*>(valid));
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// CODAE_* tuning / ablation variables, read ONCE (library load, codae_create, codae_reload_env): nothing on the
// launch path calls getenv (round 1 did, ~30 times per step)
struct EnvToggles {
    int gemm_tile = -1;          // CODAE_GEMM_TILE: s = 128 x 128 (0), q = 256 x 192 pipelined, every wave loads (3),
                                 // x = 256 x 192 pipelined, LDS-DMA on one wave per SIMD (6); -1 = automatic
    int gemm_dbg = 0;            // CODAE_GEMM_DBG timing-only ablation builds of the forward form
    int wgrad_splitk = 0;        // CODAE_WGRAD_SPLITK > 0 forces the split
    int small_tile_max = 200;    // CODAE_SMALL_TILE_MAX: forward-form launches of up to this many 128 x 128 tiles run on 64 x 64 tiles
    int small_stages = 4;        // CODAE_SMALL_STAGES=2: launches that cannot fill the chip keep the 2-stage double buffer
    int f32_gemm = 0;            // CODAE_F32_GEMM: native = v_mfma_f32_32x32x2_f32 always (1); default (0) and x3 (2) = three bf16 planes / six bf16
                                 // MFMA products wherever the shape is legal (gemm_f32.hip)
    int group_tile = -1;         // CODAE_GROUP_TILE: grouped weight gradients on 128 x 128 (0), 64 x 128 (1), 64 x 64 (2); -1 = automatic
    bool side_priority_set = false; int side_priority = 0;   // CODAE_SIDE_PRIORITY
    bool no_wt = false, single_stream = false, tail_on_side = false, no_fused_loss = false, flat_adam = false,
         no_fused_norm = false, no_chain = false, no_deep_small = false,      // CODAE_NO_DEEP_SMALL: 2-stage small GEMMs
         no_relu_bits = false,        // CODAE_NO_RELU_BITS: the data gradient reads the saved activation for its ReLU mask
         no_prefetch = false,         // CODAE_NO_PREFETCH: no touch of the next launch's weights under the epilogue
         no_defer_wgrad = false;      // CODAE_NO_DEFER_WGRAD: per-layer split-K weight gradients beside the data-gradient chain (round 2's backward)
};
const EnvToggles& env();
void env_reload();

// ---- generic exact-fp32 GEMM (gemm_f32.hip) --------------------------------
// C[i][j] = epilogue( sum_k A(i,k) * B(j,k) ), A(i,k) = A[i*a_rs + k*a_ks], B(j,k) = B[j*b_rs + k*b_ks]
struct GemmF32 {
    const float* A; int64_t a_rs, a_ks;
    const float* B; int64_t b_rs, b_ks;
    float* C; int64_t ldc;
    int M, N, K;
    const float* bias;       // [N] added before activation, or null
    int relu;                // max(v, 0)
    const float* relu_src;   // [M][ld_relu] multiply by (src > 0), or null
    int64_t ld_relu;
    float* colsum_part;      // [ceil(M / 64)][N] column sums of the stored values per 64-row block (plain stores, summed in
                             // a fixed order by launch_bias_finish: deterministic bias gradients), or null
    const int* m_dev;        // not null: the row count is min(M, *m_dev), read on the device (row tiles past it exit at once);
                             // the grid is still sized for M
    int split_k;             // > 1: K is cut into split_k ranges of whole 32-deep tiles (grid.z); range z writes its partial
                             // product to C + z * M * ldc (fp32 slabs, reduced in slab order by reduce_slabs_kernel); no
                             // bias / ReLU / column sums then
};
int gemm_f32(const GemmF32& g, hipStream_t s);
int gemm_f32x3(const GemmF32& g, hipStream_t s);      // gemm_f32x3.hip: the same product from three bf16 planes per operand
bool gemm_f32x3_takes(const GemmF32& g);              // its shape / alignment conditions (everything else stays on gemm_f32_kernel)
inline int gemm_f32_colsum_rows(int M) { return (M + 63) / 64; }

// several independent exact-fp32 GEMMs in one launch of the bf16-plane kernel (gemm_f32x3.hip): the fp32 engine's weight gradients
constexpr int CODAE_GROUP_MAX = 16;
struct GemmF32Group {
    int n;
    GemmF32 g[CODAE_GROUP_MAX];
    int wg_begin[CODAE_GROUP_MAX + 1];     // prefix sum of workgroups per GEMM (filled by the launcher)
};
int gemm_f32x3_grouped(GemmF32Group& grp, hipStream_t s);

// ---- bf16 MFMA GEMM (gemm_bf16.hip) ----------------------------------------
enum { OP_KC = 0,  // operand stored [rows][k] (k contiguous)
       OP_KS = 1   // operand stored [k][rows] (rows contiguous; transposed LDS reads)
};
// MSE loss folded into the epilogue of the last forward GEMM (train_dae_on_embedding.py:206-223):
// the kernel then writes dL/dy (bf16) instead of y and accumulates the metric sums.
struct LossFuse {
    int enabled;
    const float* data;            // dataset [n][io] fp32
    const int32_t* row_idx;       // [rows] or null
    const int32_t* mask_id;       // [rows] or null
    const int32_t* mask_to_use;   // [n][nb_run] or null
    int nb_run, run;
    const uint8_t* table;         // [n_masks][io]
    int io;
    int B;                        // valid batch rows; rows in [B, M) get dy = 0
    float inv_n;                  // 1 / (global rows * io)
    double* parts;                // [workgroups][2]: each workgroup's {sum (x-y)^2, sum (1-m)(x-y)^2}, plain stores; added up
                                  // in index order by launch_finish_loss (256 same-address double atomics cost the
                                  // kernel ~10 us of serialised tail, and their order is not reproducible)
};

struct GemmBf16 {
    const bf16_t* A; int64_t lda; int a_mode;   // output row index i
    const bf16_t* B; int64_t ldb; int b_mode;   // output col index j
    void* C; int64_t ldc; int c_f32;            // bf16 or fp32 output [M][N]
    int M, N, K;
    const float* bias;
    int relu;
    const bf16_t* relu_src; int64_t ld_relu;
    float* colsum_part;      // [tiles_m][N] per-tile column sums of the stored values (plain stores; see GemmF32), or null
    int split_k;             // >1: fp32 partial slabs C + z*M*ldc, K range split evenly in BK units
    LossFuse loss;           // enabled: C receives dy (bf16), colsum_part the last bias gradient's partial sums
    int dbg;                 // timing-only ablations (CODAE_GEMM_DBG): 1 no LDS-DMA, 2 no MFMA, 4 no epilogue stores
    // ReLU mask as one bit per element, [rows][ld_bits] bytes, bit k of byte (i, j / 8) = (stored value (i, j + k) > 0): written by a
    // forward launch (relu_bits_out), read by the data-gradient launch INSTEAD of the saved activation (relu_bits; relu_src stays
    // set for the kernels that do not take bits).  Pipelined kernels only: gemm_bf16_takes_relu_bits() says whether a launch
    // of this shape will write / read them.
    uint8_t* relu_bits_out; const uint8_t* relu_bits; int64_t ld_bits;
    const void* prefetch; int64_t prefetch_bytes;   // pipelined 256 x 192 kernel only: touched (one 4-B load per 128-B line, spread
                             // over the launch's workgroups) while the epilogue runs - the NEXT launch's weight matrix, which
                             // would otherwise come from HBM under its first K-tiles; null = nothing
    int coscheduled;         // this launch shares the chip with another stream's kernels (the data-gradient chain beside the per-layer
                             // weight gradients of the data-parallel / bucketed backward): the pipelined kernel keeps the COMPILER's
                             // instruction schedule - its pinned, hand-ordered phases (gemm_bf16_halftile.h) win 1.3 % when the launch
                             // has the chip to itself and lose 7.5 % of the step in that company (tools/abl/ab_dp.sh)
    double* sumsq_slots;     // fp32 output, 128 x 128 tile only: += sum of the stored values' squares, scattered over the
                             // CODAE_S_N_SLOTS clip_grad_norm_ slots (what sumsq_kernel would add in a pass of its own), or null
};
bool gemm_bf16_supported(int M, int N, int K);
bool gemm_bf16_takes_relu_bits(int M, int N);    // forward-form bf16 launch of this output shape runs on a pipelined kernel
int gemm_bf16_colsum_rows(const GemmBf16& g);   // rows of colsum_part this launch writes (= its tiles along M)
int gemm_bf16_loss_parts(const GemmBf16& g);    // workgroups of the fused-loss launch = rows of LossFuse::parts
int gemm_bf16(const GemmBf16& g, hipStream_t s);
int gemm_bf16_pipe(const GemmBf16& g, int cfg, hipStream_t s);   // gemm_bf16_pipe.hip

// several independent GEMMs (here: the weight gradients of every layer of a narrow stack) in ONE launch; tile
// configuration 128 x 128 for all of them
struct GemmBf16Group {
    int n;
    GemmBf16 g[CODAE_GROUP_MAX];
    int wg_begin[CODAE_GROUP_MAX + 1];     // prefix sum of workgroups per GEMM (filled by the launcher)
};
int gemm_bf16_grouped(GemmBf16Group& grp, hipStream_t s);
int gemm_bf16_pipe_grouped(GemmBf16Group& grp, hipStream_t s);   // 256 x 192 pipelined tiles, unsplit K (gemm_bf16_pipe.hip)

// ---- persistent fused chain for narrow stacks (chain_bf16.hip) ----------------
constexpr int CODAE_CHAIN_MAX_WIDTH = 512;
constexpr int CODAE_CHAIN_MAX_LAYERS = 16;
constexpr int CODAE_CHAIN_MAX_BIAS = 6144;          // floats: all layers' biases are staged in LDS
struct ChainArgs {
    int L, rows, B;                                  // rows: batch padded to a multiple of 64
    int width[CODAE_CHAIN_MAX_LAYERS + 1];           // width[l] = input of layer l, width[L] = output of the last
    uint32_t relu_flags;                             // bit l: layer l ends in a ReLU
    double* scalars;                                 // not null: workgroup 0 zeroes CODAE_S_GRAD_SQ and the norm slots
    const bf16_t* W[CODAE_CHAIN_MAX_LAYERS];         // bf16 weight shadow   [out][in]
    const bf16_t* Wt[CODAE_CHAIN_MAX_LAYERS];        // transposed shadow    [in][out]
    const float* bias[CODAE_CHAIN_MAX_LAYERS];
    bf16_t* act[CODAE_CHAIN_MAX_LAYERS];             // act[l] [rows][width[l]]: input of layer l (saved for the backward)
    bf16_t* dact[CODAE_CHAIN_MAX_LAYERS];            // dact[l] [rows][width[l+1]]: gradient of layer l's output
    float* colsum_part[CODAE_CHAIN_MAX_LAYERS];      // [rows / 16][width[l+1]] partial bias gradients
    const float* data; const int32_t* row_idx; const int32_t* mask_id; const uint8_t* mask_table;
    const int32_t* mask_to_use; int nb_run, run;
    float inv_n;                                     // 1 / (global rows * io)
    double* loss_parts;                              // [rows / 16][2]
    int do_backward;
};
bool chain_supported(int L, const int* in, const int* out);
int chain_rows_per_workgroup();
int launch_chain_step(const ChainArgs& a, hipStream_t s);

// ---- elementwise / reductions (elementwise.hip) ----------------------------
// out_ld: row stride of `out` in elements (0 = b->io: contiguous rows)
int launch_gather_corrupt(const codae_batch* b, void* out, int out_bf16, hipStream_t s, int64_t out_ld = 0);
int launch_cast_bf16(const float* src, bf16_t* dst, int64_t n, hipStream_t s);
int launch_corrupt(const float* x, const float* mask, float* out, int64_t n, hipStream_t s);
int launch_expand_masks(const int32_t* mask_id, const uint8_t* table, const int32_t* k_of_mask, int B, int io,
                        int k_max, float* masks_out, float* fmask_out, hipStream_t s);
// y fp32 [B][io]; x gathered from batch; writes dy (fp32 or bf16), metric sums, optional colsum_part
// [mse_loss_colsum_rows(B)][io] (partial sums of the last bias gradient, one row per block)
// loss_parts [mse_loss_colsum_rows(B)][2]: per-block metric sums (see LossFuse::parts)
int launch_mse_loss(const codae_batch* b, const float* y, void* dy, int dy_bf16, float inv_n, float* colsum_part,
                    double* loss_parts, int want_grad, hipStream_t s, int64_t dy_ld = 0);    // dy_ld: row stride of dy (0 = io)
int mse_loss_colsum_rows(int B);
int launch_mse_dense(const float* x, const float* y, const float* fmask, float* dy, int64_t n, float inv_n,
                     double* scalars, hipStream_t s);
int launch_sumsq(const float* g, int64_t n, double* out, hipStream_t s);
int launch_sumsq_to(const float* g, int64_t n, double* acc, hipStream_t s);      // *acc += sum g^2 (one address, <= 64 blocks)
int launch_clip_coef(const double* total_sq, float max_norm, double* coef_out, hipStream_t s);
// coef_in != null: use that precomputed clip coefficient instead of folding grad_sq + slots
// step_dev != null: take the step count (bias corrections) from that device scalar instead of hp->step
int launch_clip_adam(float* p, float* g, float* m, float* v, int64_t n, const codae_hyper* hp,
                     const double* grad_sq, bf16_t* shadow, const double* coef_in, hipStream_t s,
                     const double* step_dev = nullptr);
int launch_set_scalar(double* dst, double value, hipStream_t s);
// bf16 engine: Adam over the weight matrices in tiles, writing the bf16 shadow AND (layers >= transposed_from) the
// transposed shadow in the same pass; the flat bias block [bias_off, bias_off + bias_n) rides along
int launch_clip_adam_tiled(float* p, float* g, float* m, float* v, const codae_hyper* hp, const double* grad_sq,
                           bf16_t* shadow, bf16_t* shadow_t, int n_layers, const int64_t* w_off, const int* rows,
                           const int* cols, int transposed_from, int64_t bias_off, int64_t bias_n, hipStream_t s,
                           const double* step_dev = nullptr);
// dst[c][r] = src[r][c] for n bf16 matrices (element offsets off[i], shapes rows[i] x cols[i]) in one launch
int launch_transpose_bf16(const bf16_t* src, bf16_t* dst, int n, const int64_t* off, const int* rows, const int* cols,
                          hipStream_t s);
int gemm_bf16_timeline(unsigned long long* host_out, int n_wg);   // CODAE_GEMM_DBG=8 stamps
// sumsq != null: += sum out^2 (slot-scattered)
int launch_reduce_slabs_epi(const float* slabs, int n_slabs, int64_t stride, int M, int N, float* C, int64_t ldc, const float* bias,
                            int relu, const float* relu_src, int64_t ld_relu, float* colsum_part, hipStream_t s);
int launch_reduce_slabs(const float* slabs, int n_slabs, int64_t slab_stride, float* out, int64_t n, double* sumsq,
                        hipStream_t s);
// parts != null: SQ_FULL += sum parts[i][0], SQ_PARTIAL += sum parts[i][1] (fixed order) and the step's loss from
// them; parts == null: the step's sum was accumulated in scalars[STEP_SQ] (dense stand-alone loss)
int launch_finish_loss(double* scalars, double inv_n, hipStream_t s, const double* parts = nullptr, int n_parts = 0);
int launch_cast_f32(const bf16_t* src, float* dst, int64_t n, hipStream_t s);
// out[n] = sum_m src[m][n], rows added in index order (deterministic; stand-alone primitive)
int launch_colsum_f32(const float* src, int M, int N, float* out, hipStream_t s);
// parts[p][n] = sum of rows [64 p, 64 p + 64) of src (first stage of the engine's bias gradient of a dense dy)
int launch_colsum_parts_f32(const float* src, int M, int N, float* parts, hipStream_t s);
// Second stage of every bias gradient: out_j[c] = sum_p parts_j[p][c] (p ascending), up to 64 jobs per launch;
// sumsq != null: += sum out^2 (slot-scattered, clip_grad_norm_)
struct BiasFinishJobs {
    int n;
    const float* parts[64];
    float* out[64];
    int rows[64], cols[64];
    int col_begin[65];       // prefix sum of cols: block -> job lookup
};
// optional last block of the same launch: what finish_loss_kernel does with per-workgroup metric sums, WITHOUT resetting the
// norm accumulators (the chain kernel zeroed them before the weight gradients started adding to them)
struct LossFinish { double* scalars; double inv_n; const double* parts; int n_parts; };
int launch_bias_finish(const BiasFinishJobs& jobs, double* sumsq, hipStream_t s, const LossFinish* loss = nullptr);

}  // namespace codae
