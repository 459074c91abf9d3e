/*
 * codae_hip.h — C ABI of libcodae_hip.so: the MI355X (gfx950) implementation of
 * CODAE's denoising-autoencoder training hot path.
 *
 * The reference (victordeleau/MUI-DeepAutoEncoder) is pure Python on PyTorch and
 * has no FFI of its own; the boundary this library sits behind is the Python
 * class surface of `codae.model` / `codae.tool` as used by
 * script/train_dae_on_embedding.py and script/train_dae_on_abalone.py
 * (SURVEY.md section 8b).  Each entry point below names the reference lines it
 * replaces.  The host-side mirror (mui-deepautoencoder_amd/codae, ctypes) is the
 * only caller; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain C, no torch types: device pointers + sizes; `stream` is a hipStream_t
 *     passed as void* (torch.cuda.current_stream().cuda_stream).
 *   - every function returns 0 on success or a negative CODAE_E_* code and never
 *     throws; codae_last_error() returns a thread-local message.
 *   - no function synchronises the device or allocates device memory: all
 *     buffers (parameters, gradients, Adam state, bf16 shadows, activation
 *     workspace, scalars) are caller-allocated and borrowed for the call.
 *   - a handle is used by one host thread at a time (one process per GPU).
 *   - all matrices are row-major; Linear weights are W[out][in] fp32 as in
 *     torch.nn.Linear (embedding_denoising_autoencoder.py:63).
 */
#ifndef CODAE_HIP_H
#define CODAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped on EVERY change of a struct layout, an enum value or a function signature below.
 *   1: round 1 as first published (9-field codae_buffers)
 *   2: codae_buffers.shadow_wt, CODAE_S_ADAM_STEP / CODAE_S_COUNT 80, codae_struct_sizes, codae_reload_env,
 *      codae_train_step_graph, codae_step_backward_async, codae_side_stream, codae_join, codae_profile_stride
 *   3: codae_buffers.bias_parts, codae_sizes.bias_part_bytes, CODAE_K_CHAIN / CODAE_K_BIAS_FINISH, codae_span_sumsq,
 *      codae_step_update_span, codae_sync_transposed, codae_dgrad_bf16's partial-sum workspace
 *   4: + codae_ranking_loss_batched, codae_gather_inventory_rows, codae_step_path (new entries only; no layout change)
 *   5: + codae_monitor_accumulate; codae_ranking_loss_batched takes val_group (rows of the validation inventory that are
 *      exact duplicates of each other) behind val_pos; + codae_dp_unique_id / _init / _destroy, codae_train_step_dp
 * The binding must refuse a library whose codae_abi_version() differs and must check its own struct sizes against
 * codae_struct_sizes() at load (mui-deepautoencoder_amd/codae/hip/__init__.py does both). */
#define CODAE_ABI_VERSION 5

enum {
    CODAE_OK = 0,
    CODAE_E_INVALID = -1,   /* bad argument / shape the kernels cannot take */
    CODAE_E_HIP = -2,       /* a HIP runtime call or launch failed */
    CODAE_E_UNSUPPORTED = -3
};

/* arithmetic of the GEMM chain */
enum {
    CODAE_PREC_F32 = 0,  /* parity mode: fp32 operands and fp32 accumulation; products formed from three bf16 planes per operand
                          * (gemm_f32x3.hip: six v_mfma_f32_16x16x32_bf16 per fp32 product, closer to float64 than an fp32 fma chain)
                          * wherever rows are 16-byte aligned and K is in whole 32-deep tiles, v_mfma_f32_32x32x2_f32 elsewhere and
                          * everywhere under CODAE_F32_GEMM=native */
    CODAE_PREC_BF16 = 1  /* throughput mode: bf16 operands, fp32 accumulate, v_mfma_f32_16x16x32_bf16; every layer width a
                          * multiple of 8 (16-byte rows; the activation buffers pad their rows to multiples of 64 themselves);
                          * codae_create returns CODAE_E_UNSUPPORTED otherwise (the caller falls back to CODAE_PREC_F32) */
};

typedef struct codae_engine* codae_handle;

/* Network description: the Linear stack built by
 * EmbeddingDenoisingAutoencoder.__init__ (embedding_denoising_autoencoder.py:49-129)
 * or MixedVariableDenoisingAutoencoder.__init__ (mixed_variable_...py:45-125). */
typedef struct {
    int32_t n_layers;
    const int32_t* in_features;  /* [n_layers] */
    const int32_t* out_features; /* [n_layers] */
    const uint8_t* relu;         /* [n_layers] 1 = ReLU follows this Linear */
    int32_t max_batch;           /* rows the workspaces are sized for */
    int32_t precision;           /* CODAE_PREC_* */
} codae_spec;

/* Byte sizes / element offsets the caller needs to allocate the borrowed buffers. */
typedef struct {
    int64_t n_param;        /* elements of the flat fp32 parameter vector (with padding) */
    int64_t n_weight;       /* elements of the flat bf16 weight shadow (BF16 mode, else 0) */
    int64_t act_bytes;      /* activation workspace */
    int64_t dact_bytes;     /* activation-gradient workspace: min(n_layers + 1, 16) buffers (at least 3), one per layer
                               so that the two backward streams never wait for each other; deeper stacks rotate */
    int64_t slab_bytes;     /* split-K partial slabs for the weight-gradient GEMM (BF16 mode) */
    int64_t bias_part_bytes; /* partial column sums of the bias gradients (codae_buffers.bias_parts) */
    int32_t n_scalars;      /* doubles in the scalar block (CODAE_S_*) */
} codae_sizes;

/* Borrowed device buffers. */
typedef struct {
    float* params;      /* flat: per layer W[out*in] then b[out], each padded to 64 floats */
    float* grads;       /* same layout */
    float* adam_m;      /* same layout (exp_avg)    — may be NULL if codae_step_update is unused */
    float* adam_v;      /* same layout (exp_avg_sq) */
    void* shadow_w;     /* bf16 copy of the weights, per layer [out][in]; BF16 mode only */
    void* acts;         /* act_bytes */
    void* dacts;        /* dact_bytes */
    void* slabs;        /* slab_bytes */
    double* scalars;    /* n_scalars doubles, see CODAE_S_* */
    void* shadow_wt;    /* optional (BF16 mode): bf16 TRANSPOSED weights, per layer [in][out] at the same offsets as
                           shadow_w; when present the data-gradient GEMM reads it k-contiguously (forward-form
                           kernel) instead of reading W through transposed LDS reads; n_weight elements */
    float* bias_parts;  /* bias_part_bytes: every producer of an activation gradient leaves per-row-block partial column
                           sums here with plain stores; one small kernel per backward call adds them up in a fixed order
                           into grads' bias block (autograd's bias gradient, train_dae_on_embedding.py:210) - the step has
                           no float atomics, so the same inputs give the same bits on every run (ABI 3) */
} codae_buffers;

/* indices into codae_buffers.scalars (device memory, accumulated across calls until zeroed) */
enum {
    CODAE_S_SQ_FULL = 0,     /* sum (x-y)^2              (train_dae_on_embedding.py:218-220) */
    CODAE_S_SQ_PARTIAL = 1,  /* sum (1-fmask)(x-y)^2     (train_dae_on_embedding.py:223)     */
    CODAE_S_GRAD_SQ = 2,     /* sum g^2 of the last codae_step_update (pre-clip)             */
    CODAE_S_LAST_LOSS = 3,   /* mean MSE of the last step (train_dae_on_embedding.py:206)    */
    CODAE_S_STEP_SQ = 4,     /* scratch: sum (x-y)^2 of the current step                     */
    CODAE_S_CLIP_COEF = 5,   /* (5, 6) reserved                                                             */
    CODAE_S_GRAD_SQ_SLOTS = 8, /* 64 partial sums of g^2: same-address atomics serialise (~12 ns each), so
                                  reduction kernels scatter over these slots; sum g^2 = GRAD_SQ + sum(slots) */
    CODAE_S_N_SLOTS = 64,
    CODAE_S_ADAM_STEP = 72,  /* graph replay: Adam's step count t, written before each codae_train_step_graph launch */
    CODAE_S_COUNT = 80
};

/* One minibatch of the hot loop (train_dae_on_embedding.py:194-203). */
typedef struct {
    const float* data;       /* dataset matrix [n_rows][io] fp32 resident in HBM
                                (ConcatenatedEmbeddingDataset.data, concatenated_embedding_dataset.py:53-74) */
    const int32_t* row_idx;  /* [B] rows of `data` forming the batch (DataLoader+collate_embedding,
                                data_tool.py:96-103); NULL = rows 0..B-1 */
    const int32_t* mask_id;  /* [B] row of mask_table per sample = Corrupter.mask_to_use[idx][run]
                                (data_tool.py:252-260); NULL = no corruption */
    const uint8_t* mask_table; /* [n_masks][io] 0/1 = Corrupter.binary_masks (data_tool.py:202-209) */
    int32_t B;
    int32_t io;
    /* alternative to mask_id: look the id up on the device, id = mask_to_use[row * nb_run + run]
     * with row = row_idx[b] (Corrupter.mask_to_use, data_tool.py:222-226); used when mask_id is NULL */
    const int32_t* mask_to_use;
    int32_t nb_run;
    int32_t run;
} codae_batch;

typedef struct {
    float lr, weight_decay, beta1, beta2, eps; /* torch.optim.Adam (train_dae_on_embedding.py:160-163) */
    float max_grad_norm;   /* clip_grad_norm_(params, 1) (:213); <= 0 disables clipping */
    int32_t step;          /* 1-based Adam step index t */
    float loss_scale_rows; /* rows of the GLOBAL batch (data-parallel: sum over ranks); 0 = batch.B */
} codae_hyper;

const char* codae_last_error(void);
int codae_abi_version(void);
/* sizeof() of the structs above as THIS library was compiled, then CODAE_S_COUNT and CODAE_K_COUNT:
 * out[0..6] = {codae_spec, codae_sizes, codae_buffers, codae_batch, codae_hyper, CODAE_S_COUNT, CODAE_K_COUNT}.
 * A binding compares them with its own declarations before the first call (a short codae_buffers would make the
 * engine read shadow_wt past the caller's struct). */
#define CODAE_N_STRUCTS 7
int codae_struct_sizes(int32_t* out, int32_t capacity);
/* The CODAE_* tuning / ablation environment variables (CODAE_GEMM_TILE, CODAE_SINGLE_STREAM, CODAE_NO_FUSED_LOSS ...)
 * are read when the library is first used and at every codae_create, never on the launch path; a caller that
 * changes one and wants the stand-alone GEMM entry points to see it calls this. */
int codae_reload_env(void);

/* ---- handle ------------------------------------------------------------- */
int codae_create(const codae_spec* spec, codae_handle* out);
int codae_destroy(codae_handle h);
int codae_get_sizes(codae_handle h, codae_sizes* out);
/* element offset of layer l's weight / bias inside the flat parameter vector,
 * and of its bf16 shadow inside shadow_w */
int codae_param_offsets(codae_handle h, int32_t layer, int64_t* w_off, int64_t* b_off, int64_t* shadow_off);

/* ---- drop-in path: model(c_input) / loss.backward() ---------------------- */
/* forward(x) = decode(encode(x)) (embedding_denoising_autoencoder.py:137-185).
 * x [B][in0] fp32, y [B][outL] fp32.  layer_lo/layer_hi select a sub-chain
 * [layer_lo, layer_hi) for encode()/decode().  save_for_backward != 0 keeps the
 * activations in bufs->acts. */
int codae_forward(codae_handle h, const codae_buffers* bufs, const float* x, float* y, int32_t B,
                  int32_t layer_lo, int32_t layer_hi, int32_t save_for_backward, void* stream);
/* autograd of the chain (loss.backward(), train_dae_on_embedding.py:210):
 * dy [B][out of layer_hi-1] fp32 -> bufs->grads of layers [layer_lo, layer_hi) (overwritten),
 * optional dx [B][in of layer_lo] fp32.  Uses the activations the matching codae_forward left. */
int codae_backward(codae_handle h, const codae_buffers* bufs, const float* dy, float* dx, int32_t B,
                   int32_t layer_lo, int32_t layer_hi, void* stream);
/* refresh the bf16 weight shadows from bufs->params (after an external optimizer step) */
int codae_sync_shadows(codae_handle h, const codae_buffers* bufs, void* stream);

/* ---- fused training step: train_dae_on_embedding.py:198-223 --------------- */
/* gather+corrupt -> forward -> MSE(mean) loss + dL/dy + metric sums.
 * out_y (optional, [B][io] fp32) receives the reconstruction. */
int codae_step_forward_loss(codae_handle h, const codae_buffers* bufs, const codae_batch* batch,
                            const codae_hyper* hyper, float* out_y, void* stream);
/* backward for layers layer_hi-1 ... layer_lo (descending); call with decreasing
 * ranges to hand gradient buckets to the all-reduce as they complete. */
int codae_step_backward(codae_handle h, const codae_buffers* bufs, int32_t B, int32_t layer_lo,
                        int32_t layer_hi, void* stream);
/* global grad norm -> clip -> Adam -> bf16 shadow refresh (:212-215) */
int codae_step_update(codae_handle h, const codae_buffers* bufs, const codae_hyper* hyper, void* stream);
/* codae_train_step replayed from a hipGraph: the first call (and any call whose batch shape / pointers / hyper-
 * parameters differ from the captured ones) captures the whole step - both streams of the backward included - and
 * instantiates it; every call then costs one scalar write (Adam's step count, kept in device memory because kernel
 * arguments are frozen at capture) and one hipGraphLaunch instead of ~55 kernel launches.  For launch-bound shapes
 * (small batches: the reference's stock BATCH_SIZE 128).  batch->row_idx / mask_id must point to buffers whose
 * CONTENTS the caller refreshes between calls (same addresses).  Same results as codae_train_step. */
int codae_train_step_graph(codae_handle h, const codae_buffers* bufs, const codae_batch* batch,
                           const codae_hyper* hyper, void* stream);
/* Data-parallel form of codae_step_backward: returns WITHOUT making `stream` wait for the engine's side stream.
 * When it returns, the weight gradients of [layer_lo, layer_hi) are complete in enqueue order on the stream
 * codae_side_stream() reports (on `stream` itself when that is NULL), the bias and data gradients on `stream`.
 * The caller orders its consumer (a bucket all-reduce) behind the side stream, goes on with the next bucket, and calls
 * codae_join() before anything on `stream` reads the weight gradients (codae_step_update does so itself).
 * Replaces the autograd hooks torch's DistributedDataParallel would put on loss.backward()
 * (script/train_dae_on_embedding.py:210). */
int codae_step_backward_async(codae_handle h, const codae_buffers* bufs, int32_t B, int32_t layer_lo,
                              int32_t layer_hi, void* stream);
/* The stream the weight-gradient GEMMs and slab reduces run on (created on first use); NULL when the engine runs
 * everything on the caller's stream (CODAE_SINGLE_STREAM). Owned by the handle. */
int codae_side_stream(codae_handle h, void** stream_out);
/* Make `stream` wait for everything the engine still has in flight on its side stream (weight-gradient GEMMs and
 * slab reduces of a backward issued with codae_step_backward_async).  Call before anything on `stream`, another
 * stream or the host reads the weight gradients; codae_step_update does so itself. */
int codae_join(codae_handle h, void* stream);

/* ---- sharded data-parallel update (reduce-scatter -> Adam on 1/N of the parameters -> all-gather of the bf16 shadows)
 * What torch's ZeroRedundancyOptimizer would do around optimizer.step() (script/train_dae_on_embedding.py:212-215): each
 * rank owns a contiguous shard of every gradient bucket.  The caller (codae.train.DataParallel(sharded=True)) runs the
 * collectives; these three entry points are the arithmetic in between. */
/* *acc (device double) += sum g[i]^2, i in [0, n): a rank's share of clip_grad_norm_'s total norm */
int codae_span_sumsq(const float* g, int64_t n, double* acc, void* stream);
/* clip + Adam on elements [lo, hi) of the flat parameter / gradient / moment vectors, *total_sq (device double) being
 * the GLOBAL sum g^2 (all-reduced over the ranks); writes the bf16 shadow of the range (BF16 mode).  lo, hi multiples
 * of 4.  Does not touch the transposed shadow: codae_sync_transposed after the shadows have been all-gathered. */
int codae_step_update_span(codae_handle h, const codae_buffers* bufs, const codae_hyper* hyper, int64_t lo, int64_t hi,
                           const double* total_sq, void* stream);
/* shadow_wt <- transpose(shadow_w) for every layer that has a data gradient (BF16 mode; no-op otherwise) */
int codae_sync_transposed(codae_handle h, const codae_buffers* bufs, void* stream);
/* all three, single GPU.  Narrow stacks (bf16, every width a multiple of 64 and <= 512, at most 15 layers, widths summing
 * to <= 6144, batch <= 2048 rows) take the persistent fused chain instead of per-layer launches: ONE kernel for gather +
 * corruption + all forward layers + loss + the whole data-gradient chain (a workgroup walks 16 batch rows through every
 * layer; weights stream from L2 through per-wave LDS rings), ONE grouped launch for every layer's weight gradient (with
 * the norm's sum g^2), bias finish + loss finish, Adam: 4 launches instead of ~55 (the reference's stock BATCH_SIZE 128
 * on a narrow stack and BASELINE config 2 are launch-bound).  Same arithmetic, same buffers; CODAE_NO_CHAIN=1 keeps
 * the per-layer path; codae_step_path tells which one a batch size takes.
 * Wide stacks (per-layer path, batch >= 1024 rows, at most 15 layers, >= 200 tiles of 256 x 192 over all weight gradients): the
 * backward is the data-gradient chain followed by ONE grouped launch for every layer's weight gradient with the whole batch as
 * its k extent - no split-K slabs, no reduce pass (CODAE_NO_DEFER_WGRAD=1: the per-layer weight gradients of
 * codae_step_backward, which data-parallel callers use bucket by bucket). */
int codae_train_step(codae_handle h, const codae_buffers* bufs, const codae_batch* batch,
                     const codae_hyper* hyper, void* stream);
/* ---- data parallel with a library-owned RCCL communicator (one process per GPU) ----------------------------
 * RCCL is dlopen()ed on first use (the copy already in the process, else the system's): the library does not link it.
 * codae_dp_unique_id: rank 0 fills `out` (capacity >= 128 bytes) with an ncclUniqueId and ships it to the other ranks by any
 * means; codae_dp_init (collective: every rank calls it with the same id) creates this engine's communicator and its
 * highest-priority collective stream; codae_destroy / codae_dp_destroy release them. */
int codae_dp_unique_id(void* out, int32_t capacity);
int codae_dp_init(codae_handle h, const void* unique_id, int32_t rank, int32_t world);
int codae_dp_destroy(codae_handle h);
/* One data-parallel optimizer step with the collectives inside the call: forward + loss, the backward in n_buckets layer
 * ranges [bucket_lo[i], bucket_hi[i]) from the top layer down to layer 0, each range's weight gradients all-reduced (SUM, in
 * place, fp32) on the communicator's stream as soon as they exist - no host round trip between buckets -, the bias block
 * last, ONE wait of `stream` for the collectives, then clip + Adam on the summed gradients.  hyper->loss_scale_rows = the
 * GLOBAL batch rows.  (What DistributedDataParallel's bucket hooks + optimizer.step do around script/train_dae_on_embedding.py:210-215.) */
int codae_train_step_dp(codae_handle h, const codae_buffers* bufs, const codae_batch* batch, const codae_hyper* hyper,
                        int32_t n_buckets, const int32_t* bucket_lo, const int32_t* bucket_hi, void* stream);
/* 1 if codae_train_step / codae_eval_step with B rows run the persistent chain on this engine and these buffers, 0 if the
 * per-layer launches (tests and bench lines name the path they measured) */
int codae_step_path(codae_handle h, const codae_buffers* bufs, int32_t B);
/* validation body (:245-258): forward + metric sums only */
int codae_eval_step(codae_handle h, const codae_buffers* bufs, const codae_batch* batch, float* out_y,
                    void* stream);

/* ---- per-kernel timing (bench.py roofline leg) ------------------------------ */
/* kernel classes recorded by the engine's step/forward/backward entry points */
enum {
    CODAE_K_GEMM_FWD = 0,   /* y = act(x W^T + b) */
    CODAE_K_GEMM_DGRAD = 1, /* dx = (dy W) * relu' */
    CODAE_K_GEMM_WGRAD = 2, /* dW = dy^T x (the GEMM launch only); the grouped launch of a wide stack's step is reported as one
                             * record per weight gradient inside it, each with elapsed / count */
    CODAE_K_LOSS = 3,       /* MSE loss fwd+bwd; in the fused bf16 step: the last forward GEMM with the loss in its epilogue */
    CODAE_K_GATHER = 4,
    CODAE_K_SUMSQ = 5,
    CODAE_K_ADAM = 6,
    CODAE_K_SLAB_REDUCE = 7,
    CODAE_K_CHAIN = 8,       /* narrow stacks: gather + forward chain + loss + data-gradient chain in one launch */
    CODAE_K_BIAS_FINISH = 9, /* partial column sums -> bias gradients (+ their share of sum g^2) */
    CODAE_K_COUNT = 10
};
/* Start recording a hipEvent pair around every launch whose class bit is set in class_mask
 * (bit k = CODAE_K_k), on the stream the launch uses; at most max_records pairs are kept. */
int codae_profile_begin(codae_handle h, uint32_t class_mask, int32_t max_records);
/* Stop recording, wait for the recorded events and return per record the class and the elapsed
 * milliseconds.  n_out receives the number of records written (<= capacity). */
/* Time launches only in every n-th training step (default 1 = every step): an event pair costs 2-4 us of stream time,
 * which matters when the region being timed is also the throughput measurement. */
int codae_profile_stride(codae_handle h, int32_t every_n_steps);
int codae_profile_end(codae_handle h, int32_t* kinds, float* ms, int32_t capacity, int32_t* n_out);

/* ---- stand-alone ops (also used by the drop-in classes) ------------------- */
/* model.corrupt(input, mask) = input.clone()*mask (embedding_...py:226-239) */
int codae_corrupt(const float* x, const float* mask, float* out, int64_t n, void* stream);
/* Corrupter.get_masks (data_tool.py:239-262): masks[k][b][:] = table[id_b] if k_of_mask[id_b]==k+1 else 0;
 * fmask = sum_k masks[k].  masks_out is [k_max][B][io] contiguous, fmask_out [B][io]. */
int codae_expand_masks(const int32_t* mask_id, const uint8_t* mask_table, const int32_t* k_of_mask,
                       int32_t B, int32_t io, int32_t k_max, float* masks_out, float* fmask_out,
                       void* stream);
/* MSELoss fwd+bwd + metric sums on dense tensors: dy = 2 (y-x) * inv_n; scalars as CODAE_S_*.
 * fmask may be NULL (then SQ_PARTIAL is not touched). */
int codae_mse_loss_fwd_bwd(const float* x, const float* y, const float* fmask, float* dy, int64_t n,
                           float inv_n, double* scalars, void* stream);
/* clip_grad_norm_ + Adam on flat vectors (train_dae_on_embedding.py:212-215) */
int codae_clip_adam(float* params, float* grads, float* adam_m, float* adam_v, int64_t n,
                    const codae_hyper* hyper, double* scalars, void* stream);

/* ---- "next" rows: abalone loss and validation rank metric ---- */
/* CombinedCriterion(reduction="mean") forward + gradient (codae/tool/metering.py:155-180): per variable v with
 * span [var_pos[v], +var_size[v]): type 0 regression -> w_v * sqrt(mean (x-y)^2), type 1 classification ->
 * w_v * mean_B NLL(log_softmax(y_span), argmax x_span); loss = sum / n_var.  acc: n_var doubles of scratch;
 * dy [B][io] (may be NULL), loss_out: one double. */
int codae_combined_loss_fwd_bwd(const float* x, const float* y, int32_t B, int32_t io, int32_t n_var, const int32_t* var_pos,
                                const int32_t* var_size, const int32_t* var_type, const float* var_weight, double* acc,
                                float* dy, double* loss_out, void* stream);
/* CombinedCriterion(reduction="none") (metering.py:131-152): out[B][n_var] = squared error (size-1 regression) or NLL */
int codae_combined_loss_full(const float* x, const float* y, int32_t B, int32_t io, int32_t n_var, const int32_t* var_pos,
                             const int32_t* var_size, const int32_t* var_type, float* out, void* stream);
/* The abalone script's per-step accounting (script/train_dae_on_abalone.py:227-236 of the reference; metering.py:131-152,
 * 187-204) without leaving the device: the monitor criterion L[b][v] of (x', y') - squared error of a size-1 regression
 * variable, NLL of a one-hot block - where x' = x * undo_scale + undo_min per column (Normalizer.undo, data_tool.py:80-90;
 * both NULL = identity; a one-hot column has scale 1, min 0), added into
 *   acc[0]                       f   += sum_{b,v} L[b][v]
 *   acc[1]                       p   += sum_{b,v} L[b][v] [variable v is blanked in sample b]        (get_partial)
 *   acc[2 + k*n_var + v]         f_k += sum_{b: k_b = k+1} L[b][v]                                    (get_per_k)
 *   acc[2 + (k_max+k)*n_var + v] p_k += sum_{b: k_b = k+1} L[b][v] [v blanked in b]
 * with k_b = k_of_mask[mask_id[b]] and "blanked" = mask_table[mask_id[b]][var_pos[v]] == 0 (the Corrupter's tables).  fp64
 * accumulators, one workgroup, additions in a fixed order (the same bits every run); read them once per epoch.  k_max <= 16. */
int codae_monitor_accumulate(const float* x, const float* y, int32_t B, int32_t io, int32_t n_var, const int32_t* var_pos,
                             const int32_t* var_size, const int32_t* var_type, const float* undo_scale, const float* undo_min,
                             const int32_t* mask_id, const uint8_t* mask_table, const int32_t* k_of_mask, int32_t k_max,
                             double* acc, void* stream);
/* out[r] = ||m[r][:]||_2 */
int codae_row_norms(const float* m, int64_t rows, int32_t E, float* out, void* stream);
/* RankingLoss.get (metering.py:46-79): *out += sum_b 1 - rank_b / (n_val - 1); inventory [n_slots][n_obs][E] =
 * dataset.data_per_category (unscaled), inventory_norm [n_slots][n_obs] its row norms, idx[B] dataset index of
 * each sample, val_idx[n_val] the validation indices. */
int codae_ranking_loss(const float* pred, const float* fmask, const int32_t* idx, int32_t B, int32_t io, int32_t n_slots,
                       int32_t E, const float* inventory, const float* inventory_norm, int64_t n_obs, const int32_t* val_idx,
                       int32_t n_val, double* out, void* stream);

/* RankingLoss.get for a whole validation batch as GEMMs (metering.py:46-79), nothing through the host:
 * blanked slot of sample b from mask_table[id_b] with id_b = mask_id[b] or mask_to_use[row_idx[b] * nb_run + run] (the
 * Corrupter's device tables, as codae_batch); similarities pred[:, slot] . inv_val^T by the fp32 GEMM of the parity engine, chunk
 * validation rows at a time; *out += sum_b 1 - rank_b / (n_val - 1) (accumulates over the batches of an epoch).
 * inv_val [n_slots][n_val][E] = codae_gather_inventory_rows(inventory, val_idx), inv_val_norm its row norms;
 * val_pos [n_obs]: position of an observation in val_idx or -1 (a sample's own validation row is never counted: the
 * reference compares s[idx] with itself there), may be NULL.  val_group [n_slots][n_val] (or NULL): rows of one slot's
 * validation inventory with the same group id hold the same bytes; the reference never counts such a row against the
 * sample it duplicates either (its two similarities come out of ONE cosine_similarity call and s[idx] > s[j] is false for
 * equal values), while here the sample's own similarity and the GEMM's column are summed in different orders - so exact
 * duplicates are skipped by identity, not by arithmetic coincidence.  Every slot's GEMM runs over the rows that blank THAT slot
 * only (compacted on the device).  Workspaces: work B * chunk floats; row_state 32 * B bytes; perm_ws (n_slots * B +
 * n_slots) int32; q_ws B * E floats. */
int codae_ranking_loss_batched(const float* pred, int32_t B, int32_t io, int32_t n_slots, int32_t E, const int32_t* row_idx,
                               const int32_t* mask_id, const int32_t* mask_to_use, int32_t nb_run, int32_t run,
                               const uint8_t* mask_table, const float* inventory, const float* inventory_norm, int64_t n_obs,
                               const float* inv_val, const float* inv_val_norm, const int32_t* val_pos, const int32_t* val_group,
                               int32_t n_val, float* work, int32_t chunk, void* row_state, int32_t* perm_ws, float* q_ws, double* out,
                               void* stream);
/* dst[c][j][:] = inventory[c][val_idx[j]][:]  (inventory [n_slots][n_obs][E]) */
int codae_gather_inventory_rows(const float* inventory, int64_t n_obs, int32_t E, int32_t n_slots, const int32_t* val_idx,
                                int32_t n_val, float* dst, void* stream);

/* ---- GEMM primitives (exported for kernel-level parity tests / benchmarks) - */
/* y[M][N] = act(x[M][K] . W[N][K]^T + b[N]), fp32 in / fp32 out (the parity engine's GEMM: CODAE_PREC_F32 above) */
int codae_linear_f32(const float* x, const float* W, const float* b, float* y, int32_t M, int32_t N,
                     int32_t K, int32_t relu, void* stream);
/* dx[M][K] = (dy[M][N] . W[N][K]) * [relu_src > 0]   (relu_src [M][K] or NULL) */
int codae_dgrad_f32(const float* dy, const float* W, const float* relu_src, float* dx, int32_t M,
                    int32_t N, int32_t K, void* stream);
/* dW[N][K] = dy[M][N]^T . x[M][K] ; db[N] = colsum(dy) (db may be NULL) */
int codae_wgrad_f32(const float* dy, const float* x, float* dW, float* db, int32_t M, int32_t N,
                    int32_t K, void* stream);
/* bf16 counterparts; x, W, dy, y, dx are bf16 (uint16 storage) unless noted.
 * y_f32 != 0 -> y is fp32. */
int codae_linear_bf16(const void* x, const void* W, const float* b, void* y, int32_t y_f32, int32_t M,
                      int32_t N, int32_t K, int32_t relu, void* stream);
/* db_prev (optional, [K]) = column sums of the stored dx; needs db_ws: ceil(M / 128) * K floats of scratch */
int codae_dgrad_bf16(const void* dy, const void* W, const void* relu_src, void* dx, float* db_prev, float* db_ws,
                     int32_t M, int32_t N, int32_t K, void* stream);
int codae_wgrad_bf16(const void* dy, const void* x, float* dW, void* slabs, int64_t slab_bytes,
                     int32_t M, int32_t N, int32_t K, void* stream);
/* Tuning aid: with CODAE_GEMM_DBG=8 the forward-form bf16 GEMM stamps a 100 MHz wall clock per workgroup (entry, first
 * MFMA phase, end of K loop, stores issued, stores retired, XCC id: 6 words each); this copies the first n_wg records to
 * host memory (synchronises the device). */
int codae_debug_gemm_timeline(uint64_t* host_out, int32_t n_wg);
int codae_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
/* dst[c][r] = src[r][c] on bf16 matrices (rows, cols multiples of 8): the kernel that refreshes codae_buffers.shadow_wt
 * after an EXTERNAL parameter update (codae_sync_shadows); codae_step_update writes it inside its Adam pass
 * (W.t() in torch.nn.Linear's data gradient, embedding_denoising_autoencoder.py:137-151). */
int codae_transpose_bf16(const void* src, void* dst, int32_t rows, int32_t cols, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CODAE_HIP_H */
