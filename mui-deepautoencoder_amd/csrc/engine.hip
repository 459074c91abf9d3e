// C-ABI entry points of libcodae_hip.so and the per-model engine that chains the kernels
// into the DAE training step (script/train_dae_on_embedding.py:198-215 of the reference).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "codae_common.h"
#include <cstring>
#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only: the library is dlopen()ed (codae_dp_init), libcodae_hip.so does not link it

namespace codae {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static EnvToggles g_env;
static bool g_env_loaded = false;

void env_reload() {
    EnvToggles e;
    if (const char* t = getenv("CODAE_GEMM_TILE")) {
        switch (t[0]) { case 's': e.gemm_tile = 0; break; case 'q': e.gemm_tile = 3; break; case 'x': e.gemm_tile = 6; break; case 'm': e.gemm_tile = 7; break; default: break; }
    }
    if (const char* d = getenv("CODAE_GEMM_DBG")) e.gemm_dbg = atoi(d);
    if (const char* k = getenv("CODAE_WGRAD_SPLITK")) e.wgrad_splitk = atoi(k) > 0 ? atoi(k) : 0;
    if (const char* k = getenv("CODAE_GROUP_TILE")) e.group_tile = (k[0] >= '0' && k[0] <= '2') ? k[0] - '0' : -1;
    if (const char* pr = getenv("CODAE_SIDE_PRIORITY")) { e.side_priority_set = true; e.side_priority = atoi(pr); }
    e.no_wt = getenv("CODAE_NO_WT") != nullptr;
    e.single_stream = getenv("CODAE_SINGLE_STREAM") != nullptr;
    e.tail_on_side = getenv("CODAE_TAIL_ON_SIDE") != nullptr;
    e.no_fused_loss = getenv("CODAE_NO_FUSED_LOSS") != nullptr;
    e.flat_adam = getenv("CODAE_FLAT_ADAM") != nullptr;
    e.no_fused_norm = getenv("CODAE_NO_FUSED_NORM") != nullptr;
    e.no_chain = getenv("CODAE_NO_CHAIN") != nullptr;
    e.no_deep_small = getenv("CODAE_NO_DEEP_SMALL") != nullptr;
    e.no_defer_wgrad = getenv("CODAE_NO_DEFER_WGRAD") != nullptr;
    e.no_prefetch = getenv("CODAE_NO_PREFETCH") != nullptr;
    e.no_relu_bits = getenv("CODAE_NO_RELU_BITS") != nullptr;
    if (const char* k = getenv("CODAE_F32_GEMM")) e.f32_gemm = k[0] == 'n' ? 1 : (k[0] == 'x' ? 2 : 0);
    if (const char* k = getenv("CODAE_SMALL_TILE_MAX")) e.small_tile_max = atoi(k);
    if (const char* k = getenv("CODAE_SMALL_STAGES")) e.small_stages = atoi(k) == 2 ? 2 : 4;
    g_env = e;
    g_env_loaded = true;
}

const EnvToggles& env() {
    if (!g_env_loaded) env_reload();
    return g_env;
}

}  // namespace codae

using namespace codae;

// Flat parameter layout: [W_0 | W_1 | ... | W_{L-1} | b_0 | ... | b_{L-1}], every tensor padded to
// a multiple of 64 floats (256 B) so each starts 16-B aligned and the whole vector can be swept
// 16 B per lane by the norm / Adam kernels.  Padding stays zero under Adam (g = 0, p = 0).
constexpr int CODAE_MAX_DACT = 16;

struct codae_engine {
    int L = 0;
    std::vector<int> in, out;
    // row strides of the activation / activation-gradient buffers: bf16 mode pads every width to a multiple of 64 (the K extent of
    // the GEMM that reads the buffer walks whole 64-deep tiles) with ZERO pad columns - allocated zeroed, never written - so that
    // whatever the weight operand holds past a row's real width (the head of its next row) is multiplied by 0.  fp32: the width.
    std::vector<int> in_ld, out_ld;
    std::vector<uint8_t> relu;
    int max_batch = 0, max_rows = 0;  // max_rows = max_batch rounded up to 64
    int prec = CODAE_PREC_F32;
    std::vector<int64_t> w_off, b_off;
    int64_t n_param = 0, bias_begin = 0;
    int maxw = 0;
    std::vector<int64_t> act_off;  // byte offsets of act[0..L-1] and y (index L) inside bufs->acts
    int64_t act_bytes = 0, dact_one = 0, slab_bytes = 0;
    std::vector<int64_t> bits_off;   // byte offset inside bufs->acts of the 1-bit ReLU mask of act[l] ([max_rows][in_ld[l] / 8]), or -1
    int n_dact = 3;          // rotating activation-gradient buffers: min(L + 1, CODAE_MAX_DACT)
    std::vector<int> split_k;
    // bias gradients: per layer a block of partial column-sum rows in bufs->bias_parts, filled by whichever kernel
    // produces that layer's activation gradient; parts_pending[l] = rows waiting for finish_bias (0 = none)
    std::vector<int64_t> part_off;
    int64_t part_floats = 0;
    int64_t loss_part_off = 0;       // (floats, 8-byte aligned) per-workgroup metric sums of the loss kernels
    int loss_part_cap = 0;
    bool chain_ok = false;           // narrow bf16 stack: codae_train_step may take the persistent fused chain
    mutable std::vector<int> parts_pending;
    // optional per-launch hipEvent pairs (codae_profile_begin / _end)
    // backward on two streams: the weight-gradient GEMMs (+ slab reduce) run on `side`, concurrently with
    // the data-gradient chain on the caller's stream (they only share the read-only dA_l)
    mutable hipStream_t side = nullptr;
    mutable hipEvent_t ev_ready[CODAE_MAX_DACT] = {}, ev_join = nullptr, ev_w[CODAE_MAX_DACT] = {};
    mutable unsigned ready_turn = 0;        // ev_ready is used round robin: an event is re-recorded 16 hand-offs later at the earliest
    // dA buffer i is still being read by a side-stream wgrad (event ev_w[i]); kept across calls so that a backward
    // issued bucket by bucket without joins (codae_step_backward_async) stays ordered
    mutable bool w_pending[CODAE_MAX_DACT] = {};
    mutable bool side_dirty = false;      // side-stream work not yet joined into the caller's stream
    mutable bool bwd_coscheduled = false; // inside a backward whose weight gradients run beside the data-gradient chain (GemmBf16::coscheduled)
    // single-GPU fused step: the slab reduce of every layer also accumulates sum g^2 (clip_grad_norm_)
    mutable bool norm_in_backward = false;
    mutable bool norm_scalars_zero = false;   // finish_loss of this step zeroed GRAD_SQ + its slots and nothing added since
    mutable bool prof_on = false;
    mutable uint32_t prof_mask = 0;
    mutable int prof_n = 0;
    mutable int prof_every = 1, prof_step = 0;   // launches are timed in every prof_every-th training step only
    // codae_train_step_graph: the captured step and what it was captured for
    mutable hipGraphExec_t graph_exec = nullptr;
    mutable bool capturing = false;          // inside stream capture: device-side Adam step, everything joined at the end
    struct GraphKey { codae_batch batch; codae_hyper hyper; codae_buffers bufs; } mutable graph_key{};
    mutable std::vector<hipEvent_t> prof_start, prof_stop;
    mutable std::vector<int> prof_kind;
    mutable std::vector<int> prof_count;    // launches covered by the record (a GroupScope spans several)
    mutable bool prof_group = false;        // inside a GroupScope of the forward class: no per-launch pairs
    EnvToggles cfg;                         // the CODAE_* toggles as they stood at codae_create
    struct DpState* dp = nullptr;           // data parallel with a library-owned RCCL communicator (codae_dp_init), else null
    int esize() const { return prec == CODAE_PREC_BF16 ? 2 : 4; }
    int rows_for(int B) const { return prec == CODAE_PREC_BF16 ? (int)round_up(B, 64) : B; }
};

// ---- library-owned RCCL communicator ("the library owns ... (DP) the ncclComm_t") -------------------------
// RCCL is resolved at run time - the copy torch already loaded when there is one (soname librccl.so.1), else the system's - so
// that libcodae_hip.so loads on a box without it and a process never holds two RCCLs.
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;

static int rccl_load() {
    if (g_rccl.lib != nullptr) return CODAE_OK;
    void* lib = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);            // already in the process (torch's)
        if (lib != nullptr) break;
    }
    if (lib == nullptr)
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib != nullptr) break;
        }
    if (lib == nullptr) { codae::set_error("codae_dp: librccl.so not found (%s)", dlerror()); return CODAE_E_UNSUPPORTED; }
    RcclApi a;
    a.lib = lib;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(lib, "ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString) {
        codae::set_error("codae_dp: librccl.so lacks an entry point");
        return CODAE_E_UNSUPPORTED;
    }
    g_rccl = a;
    return CODAE_OK;
}

#define CODAE_RCCL_CHECK(expr)                                                                         \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess) {                                                                       \
            codae::set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
            return CODAE_E_HIP;                                                                        \
        }                                                                                              \
    } while (0)

constexpr int CODAE_DP_MAX_BUCKETS = 64;
struct DpState {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;                       // the collectives' own stream (highest priority: its own queue pool)
    hipEvent_t ev_bucket[CODAE_DP_MAX_BUCKETS + 1] = {};   // bucket i's gradients complete (+ the bias block)
    hipEvent_t ev_done = nullptr;                       // every collective of the step complete
};

namespace {

// records a start/stop event pair around the launches made while it is alive
struct ProfScope {
    const codae_engine* e;
    hipStream_t s;
    int slot = -1;
    ProfScope(const codae_engine* e_, int kind, hipStream_t s_) : e(e_), s(s_) {
        if (e->prof_group && kind == CODAE_K_GEMM_FWD) return;
        if (e->prof_on && ((e->prof_mask >> kind) & 1u) && (e->prof_step % e->prof_every) == 0 &&
            e->prof_n < (int)e->prof_start.size()) {
            slot = e->prof_n++;
            e->prof_kind[slot] = kind;
            e->prof_count[slot] = 1;
            (void)hipEventRecord(e->prof_start[slot], s);
        }
    }
    // the launches made inside count as `n` of this class (a grouped launch: one record, reported as n launches of elapsed / n)
    void counts_as(int n) { if (slot >= 0) e->prof_count[slot] = n; }
    ~ProfScope() {
        if (slot >= 0) (void)hipEventRecord(e->prof_stop[slot], s);
    }
};

// One event pair around a run of back-to-back launches of one class on one stream (the forward layers of a
// step: dependent kernels, zero gap between them): the pair's own cost (2-4 us of stream time) is paid once per run
// instead of once per launch, and the per-launch figure (elapsed / launches) is within 0.3 us of rocprofv3's.
struct GroupScope {
    const codae_engine* e;
    hipStream_t s;
    int slot = -1;
    GroupScope(const codae_engine* e_, int kind, hipStream_t s_) : e(e_), s(s_) {
        if (e->prof_on && ((e->prof_mask >> kind) & 1u) && (e->prof_step % e->prof_every) == 0 &&
            e->prof_n < (int)e->prof_start.size()) {
            slot = e->prof_n++;
            e->prof_kind[slot] = kind;
            e->prof_count[slot] = 0;
            e->prof_group = true;
            (void)hipEventRecord(e->prof_start[slot], s);
        }
    }
    void launched() { if (slot >= 0) ++e->prof_count[slot]; }
    void close() {
        if (slot >= 0) { (void)hipEventRecord(e->prof_stop[slot], s); e->prof_group = false; slot = -1; }
    }
    ~GroupScope() { close(); }
};

// Split-K factor of the weight-gradient GEMM dW[N][K] = dA^T H over `rows` batch rows: enough
// K-slices that the output tiles cover the chip once (256 x 192 tiles), or ~2 workgroups per CU
// with the 128 x 128 tile when the big one cannot fill it.  Must agree with gemm_bf16_tile_big.
// exact-fp32 weight gradient (128 x 128 tiles, two workgroups per CU): enough K ranges for ~1.5 workgroups per CU - a
// 1536 x 1536 gradient is 144 tiles, whose 8192-row reductions were the critical path of the whole backward (0.97 ms per
// layer on the side stream); never for batches of <= 1024 rows
int choose_split_k_f32(int N, int K, int rows) {
    if (env().wgrad_splitk > 0) return env().wgrad_splitk <= rows / 32 ? env().wgrad_splitk : 1;
    if (rows <= 1024) return 1;
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
    if (env().f32_gemm != 1 && N % 4 == 0 && K % 4 == 0) {
        // the bf16-plane kernel (gemm_f32x3.hip) runs ONE workgroup per CU: the split that leaves the last round of 256 fullest
        // (1536 x 1536: 144 tiles x 7 = 1008 of 1024 slots), at least 8 K-tiles per range, the smallest such split on a tie
        int best = 1;
        double best_fill = 0.0;
        for (int s = 1; s <= 8 && s <= rows / 256; ++s) {
            const int wgs = tiles * s, rounds = (wgs + 255) / 256;
            const double fill = (double)wgs / (256.0 * rounds);
            if (fill > best_fill + 0.02) { best_fill = fill; best = s; }
        }
        return best;
    }
    int s = (384 + tiles / 2) / tiles;
    if (s > 8) s = 8;
    if (s > rows / 256) s = rows / 256;
    return s < 1 ? 1 : s;
}

int choose_split_k(int N, int K, int rows) {
    const int kt = rows / 64;
    int s;
    // a batch of <= 256 rows is 1-4 K-tiles: the launch is all epilogue (the fp32 output), which a split multiplies and
    // follows with a reduce (stock BATCH_SIZE 128 at io 1536: 18.8 + 13 us per layer split in two)
    if (kt <= 4 && env().wgrad_splitk <= 0) return 1;
    if (env().wgrad_splitk > 0) {
        s = env().wgrad_splitk;
    } else {
        const int tiles_big = ((N + 255) / 256) * ((K + 191) / 192);
        s = (256 + tiles_big / 2) / tiles_big;
        if (s > 8) s = 8;
        if (s > kt) s = kt;
        if (s < 1) s = 1;
        if (tiles_big * s >= 160) return s;
        const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
        s = (512 + tiles / 2) / tiles;
        if (s > 8) s = 8;
    }
    if (s > kt) s = kt;
    if (s < 1) s = 1;
    return s;
}

inline void* act_ptr(const codae_engine* e, const codae_buffers* b, int l) {
    return reinterpret_cast<char*>(b->acts) + e->act_off[l];
}
inline float* part_ptr(const codae_engine* e, const codae_buffers* b, int l) { return b->bias_parts + e->part_off[l]; }
inline double* loss_parts_ptr(const codae_engine* e, const codae_buffers* b) {
    return reinterpret_cast<double*>(b->bias_parts + e->loss_part_off);
}
inline void* dact_ptr(const codae_engine* e, const codae_buffers* b, int l) {
    return reinterpret_cast<char*>(b->dacts) + (int64_t)(l % e->n_dact) * e->dact_one;   // see backward_range
}

int check_common(codae_handle h, const codae_buffers* b, int B) {
    CODAE_REQUIRE(h != nullptr && b != nullptr, "null handle or buffers");
    CODAE_REQUIRE(B > 0 && B <= h->max_batch, "batch %d outside (0, %d]", B, h->max_batch);
    CODAE_REQUIRE(b->params && b->acts, "params/acts buffer missing");
    CODAE_REQUIRE(h->prec != CODAE_PREC_BF16 || b->shadow_w, "bf16 mode needs shadow_w");
    return CODAE_OK;
}

// db_l = sum of layer l's pending partial column-sum rows, in row order (every layer with pending rows, one launch);
// with_norm: += sum db^2 into the clip_grad_norm_ slots
int finish_bias(const codae_engine* e, const codae_buffers* b, hipStream_t s, bool with_norm, const LossFinish* loss = nullptr) {
    BiasFinishJobs jobs;
    jobs.n = 0;
    int cols = 0;
    for (int l = 0; l < e->L; ++l) {
        if (e->parts_pending[l] <= 0) continue;
        if (jobs.n == 64) {                                   // (more than 64 layers pending: several launches)
            jobs.col_begin[jobs.n] = cols;
            int rc = launch_bias_finish(jobs, with_norm ? b->scalars + CODAE_S_GRAD_SQ : nullptr, s);
            if (rc) return rc;
            jobs.n = 0; cols = 0;
        }
        jobs.parts[jobs.n] = part_ptr(e, b, l);
        jobs.out[jobs.n] = b->grads + e->b_off[l];
        jobs.rows[jobs.n] = e->parts_pending[l];
        jobs.cols[jobs.n] = e->out[l];
        jobs.col_begin[jobs.n] = cols;
        cols += e->out[l];
        ++jobs.n;
        e->parts_pending[l] = 0;
    }
    if (jobs.n == 0) return CODAE_OK;
    jobs.col_begin[jobs.n] = cols;
    ProfScope prof(e, CODAE_K_BIAS_FINISH, s);
    return launch_bias_finish(jobs, with_norm ? b->scalars + CODAE_S_GRAD_SQ : nullptr, s, loss);
}

// Every 16 rows stream all the weights from L2.  Up to 128 workgroups (16 per XCD) that costs nothing extra per row: 3 x 128
// stack, ms / step chain vs per-layer: 1024 rows 0.158 / 0.31, 2048 rows 0.161 / 0.31.  With every CU streaming (4096 rows:
// 0.83 / 0.34, 8192: 1.71 / 0.40) the XCDs' L2s cannot feed them and the per-layer GEMMs (weights read once per 256 rows) win.
constexpr int CHAIN_MAX_ROWS = 2048;

bool chain_eligible(const codae_engine* e, const codae_buffers* b, int B) {
    return e->chain_ok && b->shadow_wt != nullptr && e->rows_for(B) <= CHAIN_MAX_ROWS;
}

// gather + forward chain + loss (+ data-gradient chain) of a narrow stack: one launch; then the loss finish
// (backward: the loss finish is left to the caller's bias-finish launch - *loss_out describes it - and the kernel zeroes
//  the norm accumulators)
int run_chain(const codae_engine* e, const codae_buffers* b, const codae_batch* batch, const codae_hyper* hyper, bool backward,
              hipStream_t s, LossFinish* loss_out = nullptr) {
    const int B = batch->B, L = e->L, rows = e->rows_for(B);
    ChainArgs a{};
    a.L = L; a.rows = rows; a.B = B;
    for (int l = 0; l < L; ++l) {
        a.width[l] = e->in[l];
        if (e->relu[l]) a.relu_flags |= 1u << l;
        a.W[l] = reinterpret_cast<const bf16_t*>(b->shadow_w) + e->w_off[l];
        a.Wt[l] = reinterpret_cast<const bf16_t*>(b->shadow_wt) + e->w_off[l];
        a.bias[l] = b->params + e->b_off[l];
        a.act[l] = reinterpret_cast<bf16_t*>(act_ptr(e, b, l));
        a.dact[l] = reinterpret_cast<bf16_t*>(dact_ptr(e, b, l));
        a.colsum_part[l] = part_ptr(e, b, l);
    }
    a.width[L] = e->out[L - 1];
    a.data = batch->data; a.row_idx = batch->row_idx; a.mask_id = batch->mask_id; a.mask_table = batch->mask_table;
    a.mask_to_use = batch->mask_to_use; a.nb_run = batch->nb_run; a.run = batch->run;
    const double n_glob = (double)(hyper != nullptr && hyper->loss_scale_rows > 0.f ? hyper->loss_scale_rows : (float)B) * batch->io;
    a.inv_n = (float)(1.0 / n_glob);
    a.loss_parts = loss_parts_ptr(e, b);
    a.do_backward = backward ? 1 : 0;
    a.scalars = loss_out != nullptr ? b->scalars : nullptr;
    const int n_wg = rows / chain_rows_per_workgroup();
    CODAE_REQUIRE(n_wg <= e->loss_part_cap, "chain: %d workgroups exceed the partial-sum rows (%d)", n_wg, e->loss_part_cap);
    int rc;
    {
        ProfScope prof(e, CODAE_K_CHAIN, s);
        rc = launch_chain_step(a, s);
    }
    if (rc) return rc;
    if (backward)
        for (int l = 0; l < L; ++l) e->parts_pending[l] = n_wg;
    e->norm_scalars_zero = true;
    if (loss_out != nullptr) {
        loss_out->scalars = b->scalars; loss_out->inv_n = 1.0 / ((double)B * batch->io);
        loss_out->parts = loss_parts_ptr(e, b); loss_out->n_parts = n_wg;
        return CODAE_OK;
    }
    return launch_finish_loss(b->scalars, 1.0 / ((double)B * batch->io), s, loss_parts_ptr(e, b), n_wg);
}

// every layer's weight gradient dW_l = dA_l^T H_l in one grouped launch (fp32 straight into grads: no split-K slabs)
// with_norm: each tile also adds its sum g^2 to the clip_grad_norm_ slots (no separate pass over the gradients)
int run_wgrad_grouped(const codae_engine* e, const codae_buffers* b, int rows, bool with_norm, hipStream_t s) {
    for (int base = 0; base < e->L; base += CODAE_GROUP_MAX) {
        GemmBf16Group grp{};
        grp.n = 0;
        for (int l = base; l < e->L && grp.n < CODAE_GROUP_MAX; ++l) {
            GemmBf16& g = grp.g[grp.n++];
            g.A = reinterpret_cast<const bf16_t*>(dact_ptr(e, b, l)); g.lda = e->out_ld[l]; g.a_mode = OP_KS;
            g.B = reinterpret_cast<const bf16_t*>(act_ptr(e, b, l)); g.ldb = e->in_ld[l]; g.b_mode = OP_KS;
            g.C = b->grads + e->w_off[l]; g.ldc = e->in[l]; g.c_f32 = 1;
            g.M = e->out[l]; g.N = e->in[l]; g.K = rows; g.split_k = 1;
            g.sumsq_slots = with_norm ? b->scalars + CODAE_S_GRAD_SQ_SLOTS : nullptr;
        }
        ProfScope prof(e, CODAE_K_GEMM_WGRAD, s);
        int rc = gemm_bf16_grouped(grp, s);
        if (rc) return rc;
    }
    return CODAE_OK;
}

// Wide bf16 stack, single-GPU fused step: every layer's weight gradient in ONE launch of 256 x 192 tiles with K = the whole batch
// (no split-K slabs, no reduce pass; sum g^2 from the epilogue), issued AFTER the data-gradient chain, which then has the chip to
// itself (defer_wgrad_mode).  Needs dA_l of every layer alive at once (n_dact > L) and enough tiles in total to fill the chip (C3: 48 tiles per
// layer -> round 2 split K five ways: 47 MB of fp32 slabs written and read back per layer, 0.28 ms of reduce launches per step;
// all ten together: 480 tiles on 256 CUs).  Small batches gain even more: their per-layer launches are all fixed cost (3 x 512 at the
// reference's stock batch 128: 0.436 -> 0.402 ms/step; batch 512: 0.663 -> 0.433).
// 0: per-layer weight gradients; 1: one grouped launch of the pipelined 256 x 192 tile; 2: one grouped launch of the one-barrier
// kernel on 128 x 128 / 64 x 128 / 64 x 64 tiles (run_wgrad_grouped: stacks too narrow to fill the chip with the big tile)
int defer_wgrad_mode(const codae_engine* e, int rows) {
    if (e->prec == CODAE_PREC_F32) {
        // 3: the exact-fp32 engine on the bf16-plane GEMMs - every weight gradient in one launch of gemm_f32x3_grouped_kernel, K unsplit
        // (what mode 1 is to the bf16 engine: per layer it is 144 tiles, split 7 ways into fp32 slabs and reduced: 0.2 ms of reduce
        // launches per C3 step and a prologue / epilogue per 37 K-tiles)
        if (env().f32_gemm == 1 || e->cfg.no_defer_wgrad || e->cfg.single_stream || e->L > CODAE_GROUP_MAX || e->n_dact <= e->L || rows < 256 ||
            rows % 32 != 0)
            return 0;
        int total = 0;
        for (int l = 0; l < e->L; ++l) {
            if (e->in[l] % 4 != 0 || e->out[l] % 4 != 0) return 0;
            total += ((e->out[l] + 127) / 128) * ((e->in[l] + 127) / 128);
        }
        return total >= 256 ? 3 : 0;
    }
    if (e->prec != CODAE_PREC_BF16 || e->cfg.no_defer_wgrad || e->cfg.single_stream || e->L > CODAE_GROUP_MAX || e->n_dact <= e->L || rows < 64)
        return 0;
    int total = 0, total_small = 0;
    for (int l = 0; l < e->L; ++l) {
        total += ((e->out[l] + 255) / 256) * ((e->in[l] + 191) / 192);
        total_small += ((e->out[l] + 63) / 64) * ((e->in[l] + 63) / 64);
        if ((int64_t)rows * e->out_ld[l] * 2 >= (int64_t)1 << 32 || (int64_t)rows * e->in_ld[l] * 2 >= (int64_t)1 << 32) return 0;
    }
    // (round 3 first kept the per-layer backward when one layer alone fills the chip - C5: 31.2 ms/step against 32.9 grouped; with
    //  the 4 x 8 tile order of the pipelined kernel the grouped launch wins there too: 29.5 against 30.2)
    if (total >= 200) return 1;
    return total_small >= 128 ? 2 : 0;
}

int run_wgrad_deferred_f32(const codae_engine* e, const codae_buffers* b, int rows, hipStream_t s) {
    GemmF32Group grp{};
    grp.n = 0;
    for (int l = e->L - 1; l >= 0; --l) {            // (backward order: the layers whose operands were touched last come first)
        GemmF32& g = grp.g[grp.n++];
        const int N = e->out[l], K = e->in[l];
        g.A = reinterpret_cast<const float*>(dact_ptr(e, b, l)); g.a_rs = 1; g.a_ks = N;
        g.B = reinterpret_cast<const float*>(act_ptr(e, b, l)); g.b_rs = 1; g.b_ks = K;
        g.C = b->grads + e->w_off[l]; g.ldc = K;
        g.M = N; g.N = K; g.K = rows; g.split_k = 1;
    }
    ProfScope prof(e, CODAE_K_GEMM_WGRAD, s);
    prof.counts_as(grp.n);
    return gemm_f32x3_grouped(grp, s);
}

int run_wgrad_deferred(const codae_engine* e, const codae_buffers* b, int rows, hipStream_t s) {
    GemmBf16Group grp{};
    grp.n = 0;
    for (int l = e->L - 1; l >= 0; --l) {            // (backward order: the layers whose operands were touched last come first)
        GemmBf16& g = grp.g[grp.n++];
        g.A = reinterpret_cast<const bf16_t*>(dact_ptr(e, b, l)); g.lda = e->out_ld[l]; g.a_mode = OP_KS;
        g.B = reinterpret_cast<const bf16_t*>(act_ptr(e, b, l)); g.ldb = e->in_ld[l]; g.b_mode = OP_KS;
        g.C = b->grads + e->w_off[l]; g.ldc = e->in[l]; g.c_f32 = 1;
        g.M = e->out[l]; g.N = e->in[l]; g.K = rows; g.split_k = 1;
        g.sumsq_slots = e->norm_in_backward ? b->scalars + CODAE_S_GRAD_SQ_SLOTS : nullptr;
    }
    ProfScope prof(e, CODAE_K_GEMM_WGRAD, s);
    prof.counts_as(grp.n);
    return gemm_bf16_pipe_grouped(grp, s);
}

// Exact-fp32 GEMM of a launch too small to fill the chip (forward / data gradient of a small batch): K split over
// workgroups into fp32 slabs (slab slot 2: the caller's stream), then the reduce that applies the GEMM's epilogue.  Same
// fp32 arithmetic in another summation order.  3 x 512 at batch 128, whole parity-mode step: 3.09 -> see DESIGN.md.
int gemm_f32_small(const codae_engine* e, const codae_buffers* b, const GemmF32& g, hipStream_t s) {
    const int tiles = ((g.M + 127) / 128) * ((g.N + 127) / 128);
    const int kt = (g.K + 31) / 32;
    int S = tiles >= 128 ? 1 : (256 + tiles / 2) / tiles;
    if (S > kt / 4) S = kt / 4;                       // at least 4 K-tiles per range
    if (S > 16) S = 16;
    while (S > 1 && (int64_t)S * g.M * g.N * 4 > e->slab_bytes) --S;
    if (S <= 1 || b->slabs == nullptr || e->cfg.no_deep_small || (g.N % 4) != 0 || (g.ldc % 4) != 0 || g.m_dev != nullptr ||
        (g.relu_src != nullptr && (g.ld_relu % 4) != 0))
        return gemm_f32(g, s);
    float* slab = reinterpret_cast<float*>(reinterpret_cast<char*>(b->slabs) + 2 * e->slab_bytes);
    GemmF32 p = g;
    p.C = slab; p.ldc = g.N; p.bias = nullptr; p.relu = 0; p.relu_src = nullptr; p.ld_relu = 0; p.colsum_part = nullptr; p.split_k = S;
    int rc = gemm_f32(p, s);
    if (rc) return rc;
    return launch_reduce_slabs_epi(slab, S, (int64_t)g.M * g.N, g.M, g.N, g.C, g.ldc, g.bias, g.relu, g.relu_src, g.ld_relu, g.colsum_part, s);
}

// y = act(x W^T + b) for layer l
int run_linear(const codae_engine* e, const codae_buffers* b, int l, const void* x, void* y, bool y_f32, int rows,
               hipStream_t s) {
    const int N = e->out[l], K = e->in[l];
    ProfScope prof(e, CODAE_K_GEMM_FWD, s);
    if (e->prec == CODAE_PREC_BF16) {
        GemmBf16 g{};
        // K = the PADDED input width: the extra k columns are zeros in x (see in_ld) times the head of W's next row
        g.A = reinterpret_cast<const bf16_t*>(x); g.lda = e->in_ld[l]; g.a_mode = OP_KC;
        g.B = reinterpret_cast<const bf16_t*>(b->shadow_w) + e->w_off[l]; g.ldb = K; g.b_mode = OP_KC;
        g.C = y; g.ldc = y_f32 ? N : e->out_ld[l]; g.c_f32 = y_f32 ? 1 : 0;
        g.M = rows; g.N = N; g.K = e->in_ld[l];
        g.bias = b->params + e->b_off[l]; g.relu = e->relu[l];
        g.split_k = 1;
        if (!y_f32 && l + 1 < e->L && e->bits_off[l + 1] >= 0 && e->relu[l] && y == act_ptr(e, b, l + 1) && gemm_bf16_takes_relu_bits(rows, N)) {
            g.relu_bits_out = reinterpret_cast<uint8_t*>(b->acts) + e->bits_off[l + 1];
            g.ld_bits = e->in_ld[l + 1] / 8;
        }
        if (l + 1 < e->L && !e->cfg.no_prefetch) {        // the next layer's weights, touched under this launch's epilogue
            g.prefetch = reinterpret_cast<const bf16_t*>(b->shadow_w) + e->w_off[l + 1];
            g.prefetch_bytes = (int64_t)e->in[l + 1] * e->out[l + 1] * 2;
        }
        return gemm_bf16(g, s);
    }
    GemmF32 g{};
    g.A = reinterpret_cast<const float*>(x); g.a_rs = K; g.a_ks = 1;
    g.B = b->params + e->w_off[l]; g.b_rs = K; g.b_ks = 1;
    g.C = reinterpret_cast<float*>(y); g.ldc = N;
    g.M = rows; g.N = N; g.K = K;
    g.bias = b->params + e->b_off[l]; g.relu = e->relu[l];
    return gemm_f32_small(e, b, g, s);
}

// dW_l = dA_l^T act[l] on stream s.  bf16: split-K partial slabs (slab buffer `slot`), then the reduce on the same stream.
int run_wgrad(const codae_engine* e, const codae_buffers* b, int l, int rows, hipStream_t s, int slot = 0) {
    const int N = e->out[l], K = e->in[l];
    float* dW = b->grads + e->w_off[l];
    if (e->prec == CODAE_PREC_BF16) {
        const int S = e->split_k[l] <= rows / 64 ? e->split_k[l] : rows / 64;
        GemmBf16 g{};
        g.A = reinterpret_cast<const bf16_t*>(dact_ptr(e, b, l)); g.lda = e->out_ld[l]; g.a_mode = OP_KS;
        g.B = reinterpret_cast<const bf16_t*>(act_ptr(e, b, l)); g.ldb = e->in_ld[l]; g.b_mode = OP_KS;
        g.M = N; g.N = K; g.K = rows;
        g.ldc = K; g.c_f32 = 1; g.split_k = S;
        if (S > 1) {
            CODAE_REQUIRE(b->slabs != nullptr, "bf16 wgrad needs the slab workspace");
            char* slab = reinterpret_cast<char*>(b->slabs) + (int64_t)slot * e->slab_bytes;
            g.C = slab;
            int rc;
            {
                ProfScope prof(e, CODAE_K_GEMM_WGRAD, s);
                rc = gemm_bf16(g, s);
            }
            if (rc) return rc;
            ProfScope prof(e, CODAE_K_SLAB_REDUCE, s);
            return launch_reduce_slabs(reinterpret_cast<const float*>(slab), S, (int64_t)N * K, dW, (int64_t)N * K,
                                       e->norm_in_backward ? b->scalars + CODAE_S_GRAD_SQ : nullptr, s);
        }
        g.C = dW;
        if (e->norm_in_backward) g.sumsq_slots = b->scalars + CODAE_S_GRAD_SQ_SLOTS;    // (unsplit: the epilogue sees the final values)
        ProfScope prof(e, CODAE_K_GEMM_WGRAD, s);
        return gemm_bf16(g, s);
    }
    GemmF32 g{};
    g.A = reinterpret_cast<const float*>(dact_ptr(e, b, l)); g.a_rs = 1; g.a_ks = N;
    g.B = reinterpret_cast<const float*>(act_ptr(e, b, l)); g.b_rs = 1; g.b_ks = K;
    g.C = dW; g.ldc = K;
    g.M = N; g.N = K; g.K = rows;
    int S = e->split_k[l];
    if (S > rows / 32) S = rows / 32;
    if (S > 1 && b->slabs != nullptr) {
        // K split over workgroups: fp32 slabs, added in slab order by the reduce (same fp32 arithmetic, another summation order)
        float* slab = reinterpret_cast<float*>(reinterpret_cast<char*>(b->slabs) + (int64_t)slot * e->slab_bytes);
        g.C = slab; g.split_k = S;
        int rc;
        {
            ProfScope prof(e, CODAE_K_GEMM_WGRAD, s);
            rc = gemm_f32(g, s);
        }
        if (rc) return rc;
        ProfScope prof(e, CODAE_K_SLAB_REDUCE, s);
        return launch_reduce_slabs(slab, S, (int64_t)N * K, dW, (int64_t)N * K, nullptr, s);
    }
    ProfScope prof(e, CODAE_K_GEMM_WGRAD, s);
    return gemm_f32(g, s);
}

// dA_{l-1} = (dA_l W_l) * [act[l] > 0]  (+ column sums -> db_{l-1});  l == 0 with dx: plain dX in fp32
int run_dgrad(const codae_engine* e, const codae_buffers* b, int l, int rows, float* dx_f32, hipStream_t s) {
    const int N = e->out[l], K = e->in[l];
    const bool to_dx = (dx_f32 != nullptr);
    ProfScope prof(e, CODAE_K_GEMM_DGRAD, s);
    if (e->prec == CODAE_PREC_BF16) {
        GemmBf16 g{};
        g.coscheduled = e->bwd_coscheduled ? 1 : 0;
        g.A = reinterpret_cast<const bf16_t*>(dact_ptr(e, b, l)); g.lda = e->out_ld[l]; g.a_mode = OP_KC;
        if (b->shadow_wt != nullptr && l >= 1 && !e->cfg.no_wt) {
            // dx[m][k] = sum_n dy[m][n] Wt[k][n]: both operands k-contiguous -> the forward-form kernel
            g.B = reinterpret_cast<const bf16_t*>(b->shadow_wt) + e->w_off[l]; g.ldb = N; g.b_mode = OP_KC;
        } else {
            g.B = reinterpret_cast<const bf16_t*>(b->shadow_w) + e->w_off[l]; g.ldb = K; g.b_mode = OP_KS;
        }
        g.M = rows; g.N = K; g.K = e->out_ld[l];          // (k runs over the padded output width: zero pad columns of dA)
        g.ldc = K; g.split_k = 1;
        if (to_dx) {
            g.C = dx_f32; g.c_f32 = 1;
        } else {
            g.C = dact_ptr(e, b, l - 1); g.c_f32 = 0; g.ldc = e->out_ld[l - 1];
            if (e->relu[l - 1]) {
                g.relu_src = reinterpret_cast<const bf16_t*>(act_ptr(e, b, l)); g.ld_relu = e->in_ld[l];
                // (the forward launch that wrote act[l] had this output shape: rows x in[l]; same predicate on both sides)
                if (e->bits_off[l] >= 0 && gemm_bf16_takes_relu_bits(rows, K)) {
                    g.relu_bits = reinterpret_cast<const uint8_t*>(b->acts) + e->bits_off[l];
                    g.ld_bits = e->in_ld[l] / 8;
                }
            }
            g.colsum_part = part_ptr(e, b, l - 1);
            e->parts_pending[l - 1] = gemm_bf16_colsum_rows(g);
            if (b->shadow_wt != nullptr && l >= 2 && !e->cfg.no_wt && !e->cfg.no_prefetch) {      // the next data gradient's operand
                g.prefetch = reinterpret_cast<const bf16_t*>(b->shadow_wt) + e->w_off[l - 1];
                g.prefetch_bytes = (int64_t)e->in[l - 1] * e->out[l - 1] * 2;
            }
        }
        return gemm_bf16(g, s);
    }
    GemmF32 g{};
    g.A = reinterpret_cast<const float*>(dact_ptr(e, b, l)); g.a_rs = N; g.a_ks = 1;
    g.B = b->params + e->w_off[l]; g.b_rs = 1; g.b_ks = K;
    g.M = rows; g.N = K; g.K = N;
    g.ldc = K;
    if (to_dx) {
        g.C = dx_f32;
    } else {
        g.C = reinterpret_cast<float*>(dact_ptr(e, b, l - 1));
        if (e->relu[l - 1]) { g.relu_src = reinterpret_cast<const float*>(act_ptr(e, b, l)); g.ld_relu = K; }
        g.colsum_part = part_ptr(e, b, l - 1);
        e->parts_pending[l - 1] = gemm_f32_colsum_rows(rows);
    }
    return gemm_f32_small(e, b, g, s);
}

// Wt_l [in][out] <- W_l [out][in] (bf16) for every layer that has a data gradient (l >= 1)
int refresh_transposed(const codae_engine* e, const codae_buffers* b, hipStream_t s, int l_lo = 1, int l_hi = -1) {
    if (e->prec != CODAE_PREC_BF16 || b->shadow_wt == nullptr || e->L < 2) return CODAE_OK;
    if (l_hi < 0) l_hi = e->L;
    for (int base = l_lo; base < l_hi; base += 64) {             // the launch carries at most 64 matrices
        int64_t off[64]; int rows[64], cols[64]; int n = 0;
        for (int l = base; l < l_hi && n < 64; ++l) { off[n] = e->w_off[l]; rows[n] = e->out[l]; cols[n] = e->in[l]; ++n; }
        int rc = launch_transpose_bf16(reinterpret_cast<const bf16_t*>(b->shadow_w), reinterpret_cast<bf16_t*>(b->shadow_wt),
                                       n, off, rows, cols, s);
        if (rc) return rc;
    }
    return CODAE_OK;
}

// rows [B, rows) of a [rows][width] working-precision matrix -> 0 (bf16 mode pads the batch to 64)
int zero_pad_rows(const codae_engine* e, void* base, int B, int rows, int width, hipStream_t s) {
    if (rows > B) {
        char* p = reinterpret_cast<char*>(base) + (int64_t)B * width * e->esize();
        CODAE_HIP_CHECK(hipMemsetAsync(p, 0, (int64_t)(rows - B) * width * e->esize(), s));
    }
    return CODAE_OK;
}

int ensure_side_stream(const codae_engine* h) {
    if (h->side != nullptr) return CODAE_OK;
    // The side streams must land on a hardware queue of their own: two streams that share one execute serially
    // with ~11 us between dependent kernels (seen with RCCL initialised: every stream of the process on one queue,
    // the whole backward serial).  The runtime keeps a separate queue pool per priority level, so the side streams
    // take the LOWEST priority: never the queue of the caller's normal-priority stream, nor of RCCL's
    // high-priority ones; and the dgrad chain (critical path, caller's stream) is dispatched first.
    int prio_least = 0, prio_greatest = 0;
    CODAE_HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    const int prio = h->cfg.side_priority_set ? h->cfg.side_priority : prio_least;
    CODAE_HIP_CHECK(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, prio));
    for (int i = 0; i < CODAE_MAX_DACT; ++i) CODAE_HIP_CHECK(hipEventCreateWithFlags(&h->ev_ready[i], hipEventDisableTiming));
    CODAE_HIP_CHECK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    for (int i = 0; i < CODAE_MAX_DACT; ++i) CODAE_HIP_CHECK(hipEventCreateWithFlags(&h->ev_w[i], hipEventDisableTiming));
    return CODAE_OK;
}

// s waits for everything the backward put on the side streams
int join_side(const codae_engine* h, hipStream_t s) {
    if (!h->side_dirty || h->side == nullptr) return CODAE_OK;
    CODAE_HIP_CHECK(hipEventRecord(h->ev_join, h->side));
    CODAE_HIP_CHECK(hipStreamWaitEvent(s, h->ev_join, 0));
    for (int i = 0; i < CODAE_MAX_DACT; ++i) h->w_pending[i] = false;
    h->side_dirty = false;
    return CODAE_OK;
}

// Backward over layers hi-1 ... lo.  Step path (dx == nullptr, step_mode): the chain continues
// below `lo` in a later call, so the dgrad of every layer > 0 runs.  Drop-in sub-chain: the dgrad
// of layer `lo` runs only when the caller wants dx, and then lands unmasked in fp32.
//
// Two streams: wgrad_l (+ its slab reduce) goes to the side stream as soon as dA_l exists, the
// dgrad chain stays on `s`; at K = 1536 a third of every GEMM launch is ramp + output drain with
// all CUs in the same phase, and the two kernels fill each other's bubbles.  dA lives in one buffer per
// layer (n_dact = L + 1, up to CODAE_MAX_DACT): nothing is overwritten within a step, so the dgrad chain never
// waits for the side stream; deeper stacks rotate (dgrad_l writes buffer (l-1) % n, which wgrad_{l-1+n} was
// reading, and waits for that wgrad's event).  The last weight gradient (layer 0) runs on `s` itself, beside
// wgrad_1 on the side stream.  With join the call returns with `s` waiting for every side-stream kernel.
int backward_range(codae_handle h, const codae_buffers* b, int B, int lo, int hi, float* dx, bool step_mode,
                   hipStream_t s, bool join = true) {
    const int rows = h->rows_for(B);
    const bool dual = !h->cfg.single_stream;
    if (dual) {
        int rc = ensure_side_stream(h);
        if (rc) return rc;
    }
    // (Measured and dropped: the slab reduces on the caller's stream one layer late, or on a third stream - no
    // gain, 1.83 / 1.84 vs 1.82 ms at the time; every GEMM on the caller's stream with only the reduces beside
    // them - 0.27 ms slower: each cross-queue event costs ~10 us.)
    // (the drop-in backward - codae_backward over the whole stack without an input gradient - takes the same schedule)
    const int defer = ((step_mode || dx == nullptr) && join && lo == 0 && hi == h->L && h->L >= 2) ? defer_wgrad_mode(h, rows) : 0;
    if (defer != 0) {
        // data-gradient chain alone on the chip, then all weight gradients in one launch (see defer_wgrad_mode); one stream
        int rc = join_side(h, s);                  // (an earlier bucketed backward may have left work on the side stream)
        if (rc) return rc;
        for (int l = hi - 1; l >= 1; --l) {
            rc = run_dgrad(h, b, l, rows, nullptr, s);
            if (rc) return rc;
        }
        rc = defer == 3 ? run_wgrad_deferred_f32(h, b, rows, s)
                        : (defer == 1 ? run_wgrad_deferred(h, b, rows, s) : run_wgrad_grouped(h, b, rows, h->norm_in_backward, s));
        if (rc) return rc;
        return finish_bias(h, b, s, h->norm_in_backward);
    }
    bool* w_pending = h->w_pending;
    struct Cosched {                       // (reset on every way out of the function)
        codae_handle h;
        ~Cosched() { h->bwd_coscheduled = false; }
    } cosched_guard{h};
    h->bwd_coscheduled = dual;
    for (int l = hi - 1; l >= lo; --l) {
        int rc;
        if (!dual) {
            rc = run_wgrad(h, b, l, rows, s);
        } else if (join && step_mode && l == 0 && hi - lo >= 3 && !h->cfg.tail_on_side) {
            // tail: the side stream still owes wgrad_1 when the dgrad chain ends, and the caller's stream has
            // nothing left to do: the last weight gradient runs here (third slab buffer), beside wgrad_1
            rc = run_wgrad(h, b, l, rows, s, 2);
        } else {
            hipEvent_t ready = h->ev_ready[h->ready_turn++ % CODAE_MAX_DACT];
            CODAE_HIP_CHECK(hipEventRecord(ready, s));                      // dA_l (and act[l]) are complete on s
            CODAE_HIP_CHECK(hipStreamWaitEvent(h->side, ready, 0));
            rc = run_wgrad(h, b, l, rows, h->side, l & 1);                  // two alternating slab buffers
            if (rc == CODAE_OK && h->n_dact <= h->L) {   // (with a buffer per layer nothing is overwritten within a step)
                CODAE_HIP_CHECK(hipEventRecord(h->ev_w[l % h->n_dact], h->side));
                w_pending[l % h->n_dact] = true;
            }
        }
        if (rc) return rc;
        const bool chain = step_mode ? (l > 0) : (l > lo);
        const bool to_dx = !chain && !step_mode && dx != nullptr;
        if (chain || to_dx) {
            // the buffer dgrad_l writes, (l-1) % n, was last read by wgrad_{l-1+n}
            const int wb = (l - 1 + h->n_dact) % h->n_dact;
            if (dual && chain && w_pending[wb]) {
                CODAE_HIP_CHECK(hipStreamWaitEvent(s, h->ev_w[wb], 0));
                w_pending[wb] = false;
            }
            rc = chain ? run_dgrad(h, b, l, rows, nullptr, s) : run_dgrad(h, b, l, B, dx, s);
            if (rc) return rc;
        }
    }
    // bias gradients of every layer whose activation gradient was produced since the last finish (this range's dgrads; the
    // loss, when the range starts at the top): partial rows -> grads' bias block, on the caller's stream.  A backward issued
    // bucket by bucket without joins (data parallel: one bucket per layer) leaves the partial rows pending until its LAST
    // range (lo == 0) - one finish launch per step instead of one per bucket (10 x 6 us at C3); codae_step_update finishes
    // whatever a caller that stopped short left behind.
    if (join || lo == 0) {
        int rc = finish_bias(h, b, s, h->norm_in_backward);
        if (rc) return rc;
    }
    if (dual) {
        h->side_dirty = true;
        if (join) return join_side(h, s);
    }
    return CODAE_OK;
}

}  // namespace

extern "C" {

const char* codae_last_error(void) { return g_err; }
int codae_abi_version(void) { return CODAE_ABI_VERSION; }

int codae_reload_env(void) {
    env_reload();
    return CODAE_OK;
}

int codae_struct_sizes(int32_t* out, int32_t capacity) {
    CODAE_REQUIRE(out != nullptr && capacity >= CODAE_N_STRUCTS, "codae_struct_sizes: need room for %d entries", CODAE_N_STRUCTS);
    out[0] = (int32_t)sizeof(codae_spec);
    out[1] = (int32_t)sizeof(codae_sizes);
    out[2] = (int32_t)sizeof(codae_buffers);
    out[3] = (int32_t)sizeof(codae_batch);
    out[4] = (int32_t)sizeof(codae_hyper);
    out[5] = CODAE_S_COUNT;
    out[6] = CODAE_K_COUNT;
    return CODAE_OK;
}

int codae_create(const codae_spec* spec, codae_handle* out) {
    CODAE_REQUIRE(spec && out, "codae_create: null argument");
    CODAE_REQUIRE(spec->n_layers > 0 && spec->n_layers <= 64, "codae_create: n_layers %d", spec->n_layers);
    CODAE_REQUIRE(spec->max_batch > 0, "codae_create: max_batch %d", spec->max_batch);
    CODAE_REQUIRE(spec->precision == CODAE_PREC_F32 || spec->precision == CODAE_PREC_BF16, "codae_create: precision");
    for (int l = 0; l < spec->n_layers; ++l) {
        CODAE_REQUIRE(spec->in_features[l] > 0 && spec->out_features[l] > 0, "codae_create: layer %d has an empty side", l);
        CODAE_REQUIRE(l == 0 || spec->in_features[l] == spec->out_features[l - 1],
                      "codae_create: layer %d input %d != previous output %d", l, spec->in_features[l], spec->out_features[l - 1]);
        if (spec->precision == CODAE_PREC_BF16) {
            // 16-byte rows (8 bf16) everywhere: vector loads / stores of the gather, the epilogues and the k-strided staging.  (Round 2
            // needed multiples of 64: every width is also a GEMM k extent; the activation buffers now carry that padding themselves.)
            if (spec->in_features[l] % 8 != 0 || spec->out_features[l] % 8 != 0) {
                set_error("codae_create: bf16 mode needs every layer width to be a multiple of 8 (layer %d is %d -> %d); use CODAE_PREC_F32",
                          l, spec->in_features[l], spec->out_features[l]);
                return CODAE_E_UNSUPPORTED;
            }
        }
    }
    env_reload();                 // the one place (besides library load / codae_reload_env) the CODAE_* variables are read
    codae_engine* e = new codae_engine();
    e->cfg = env();
    e->L = spec->n_layers;
    e->prec = spec->precision;
    e->max_batch = spec->max_batch;
    e->max_rows = (int)round_up(spec->max_batch, 64);
    int64_t off = 0;
    for (int l = 0; l < e->L; ++l) {
        e->in.push_back(spec->in_features[l]);
        e->out.push_back(spec->out_features[l]);
        e->relu.push_back(spec->relu[l] ? 1 : 0);
        e->in_ld.push_back(e->prec == CODAE_PREC_BF16 ? (int)round_up(e->in[l], 64) : e->in[l]);
        e->out_ld.push_back(e->prec == CODAE_PREC_BF16 ? (int)round_up(e->out[l], 64) : e->out[l]);
        e->w_off.push_back(off);
        off += round_up((int64_t)e->in[l] * e->out[l], 64);
        if (e->in_ld[l] > e->maxw) e->maxw = e->in_ld[l];
        if (e->out_ld[l] > e->maxw) e->maxw = e->out_ld[l];
    }
    e->bias_begin = off;
    for (int l = 0; l < e->L; ++l) {
        e->b_off.push_back(off);
        off += round_up(e->out[l], 64);
    }
    e->n_param = off;
    int64_t a = 0;
    for (int l = 0; l < e->L; ++l) {
        e->act_off.push_back(a);
        a += round_up((int64_t)e->max_rows * e->in_ld[l] * e->esize(), 256);
    }
    e->act_off.push_back(a);  // y, always fp32
    a += round_up((int64_t)e->max_rows * e->out[e->L - 1] * 4, 256);
    // 1 bit per element of every activation that a ReLU produced (the data gradient reads these instead of the activation)
    e->bits_off.assign(e->L, -1);
    if (e->prec == CODAE_PREC_BF16 && !e->cfg.no_relu_bits) {
        for (int l = 1; l < e->L; ++l) {
            if (!e->relu[l - 1]) continue;
            e->bits_off[l] = a;
            a += round_up((int64_t)e->max_rows * (e->in_ld[l] / 8), 256);
        }
    }
    e->act_bytes = a;
    e->dact_one = round_up((int64_t)e->max_rows * e->maxw * e->esize(), 256);
    // one activation-gradient buffer per layer when the stack is shallow enough (no write-after-read waits between the
    // two backward streams, whose barrier packets cost ~11 us each); deeper stacks rotate through CODAE_MAX_DACT
    e->n_dact = e->L + 1 <= CODAE_MAX_DACT ? e->L + 1 : CODAE_MAX_DACT;
    if (e->n_dact < 3) e->n_dact = 3;
    if (e->prec == CODAE_PREC_BF16 && e->n_dact <= e->L) {
        // rotating buffers are shared by layers of different widths: one layer's zero pad columns would be another's data
        for (int l = 0; l < e->L; ++l)
            if (e->in[l] % 64 != 0 || e->out[l] % 64 != 0) {
                set_error("codae_create: bf16 stacks deeper than %d layers need widths that are multiples of 64 (layer %d is %d -> %d)",
                          CODAE_MAX_DACT - 1, l, e->in[l], e->out[l]);
                delete e;
                return CODAE_E_UNSUPPORTED;
            }
    }
    e->chain_ok = e->prec == CODAE_PREC_BF16 && !e->cfg.no_chain && e->L + 1 <= CODAE_MAX_DACT &&
                  chain_supported(e->L, e->in.data(), e->out.data()) && e->in[0] == e->out[e->L - 1];
    // partial column-sum rows per layer: a producer writes at most one row per 64 batch rows (exact-fp32 GEMM, dense
    // colsum), the stand-alone loss kernel of the last layer one per 32, the persistent chain one per 16
    e->parts_pending.assign(e->L, 0);
    e->part_floats = 0;
    const int chain_rows = e->max_rows < CHAIN_MAX_ROWS ? e->max_rows : CHAIN_MAX_ROWS;
    for (int l = 0; l < e->L; ++l) {
        e->part_off.push_back(e->part_floats);
        int64_t rows_cap = (l == e->L - 1) ? (e->max_rows + 31) / 32 : (e->max_rows + 63) / 64;
        if (e->chain_ok && chain_rows / 16 > rows_cap) rows_cap = chain_rows / 16;
        e->part_floats += round_up(rows_cap * (int64_t)e->out[l], 64);
    }
    {   // + the loss kernels' per-workgroup metric sums: [workgroups][2] doubles
        const int io = e->out[e->L - 1];
        const int by_rows = (e->max_rows + 31) / 32;                                             // stand-alone loss kernel
        const int by_tiles = ((e->max_rows + 63) / 64) * ((io + 63) / 64);                       // fused into the last GEMM (smallest tile)
        e->loss_part_cap = by_rows > by_tiles ? by_rows : by_tiles;
        if (e->chain_ok && chain_rows / 16 > e->loss_part_cap) e->loss_part_cap = chain_rows / 16;
        e->loss_part_off = e->part_floats;
        e->part_floats += round_up((int64_t)e->loss_part_cap * 4, 64);
    }
    e->slab_bytes = 0;
    for (int l = 0; l < e->L; ++l) {
        int s = 1;
        if (e->prec == CODAE_PREC_BF16) s = choose_split_k(e->out[l], e->in[l], e->max_rows);
        else s = choose_split_k_f32(e->out[l], e->in[l], e->max_rows);
        e->split_k.push_back(s);
        if (s > 1) {
            const int64_t bytes = (int64_t)s * e->in[l] * e->out[l] * 4;
            if (bytes > e->slab_bytes) e->slab_bytes = bytes;
        }
    }
    if (e->prec == CODAE_PREC_F32 && e->max_rows <= 512) {
        // room for the split-K slabs of a small batch's forward / data-gradient launches (gemm_f32_small)
        const int64_t bytes = (int64_t)16 * e->max_rows * e->maxw * 4;
        if (bytes > e->slab_bytes) e->slab_bytes = bytes;
    }
    *out = e;
    return CODAE_OK;
}

static void profile_release(codae_handle h) {
    for (hipEvent_t ev : h->prof_start) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : h->prof_stop) (void)hipEventDestroy(ev);
    h->prof_start.clear(); h->prof_stop.clear(); h->prof_kind.clear(); h->prof_count.clear(); h->prof_group = false;
    h->prof_on = false; h->prof_n = 0;
}

int codae_destroy(codae_handle h) {
    if (h) profile_release(h);
    if (h && h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
    if (h && h->side) {
        (void)hipStreamSynchronize(h->side);
        for (int i = 0; i < CODAE_MAX_DACT; ++i) if (h->ev_ready[i]) (void)hipEventDestroy(h->ev_ready[i]);
        (void)hipEventDestroy(h->ev_join);
        for (int i = 0; i < CODAE_MAX_DACT; ++i) (void)hipEventDestroy(h->ev_w[i]);
        (void)hipStreamDestroy(h->side);
    }
    if (h && h->dp) (void)codae_dp_destroy(h);
    delete h;
    return CODAE_OK;
}

int codae_profile_begin(codae_handle h, uint32_t class_mask, int32_t max_records) {
    CODAE_REQUIRE(h && max_records > 0 && max_records <= (1 << 20), "codae_profile_begin: bad arguments");
    profile_release(h);
    h->prof_start.resize(max_records); h->prof_stop.resize(max_records); h->prof_kind.assign(max_records, -1); h->prof_count.assign(max_records, 1);
    for (int i = 0; i < max_records; ++i) {
        CODAE_HIP_CHECK(hipEventCreate(&h->prof_start[i]));
        CODAE_HIP_CHECK(hipEventCreate(&h->prof_stop[i]));
    }
    h->prof_mask = class_mask; h->prof_n = 0; h->prof_on = true; h->prof_step = 0;
    return CODAE_OK;
}

int codae_profile_stride(codae_handle h, int32_t every_n_steps) {
    CODAE_REQUIRE(h && every_n_steps >= 1, "codae_profile_stride: bad arguments");
    h->prof_every = every_n_steps;
    return CODAE_OK;
}

int codae_profile_end(codae_handle h, int32_t* kinds, float* ms, int32_t capacity, int32_t* n_out) {
    CODAE_REQUIRE(h && kinds && ms && n_out, "codae_profile_end: null argument");
    h->prof_on = false;
    int n = 0;
    for (int i = 0; i < h->prof_n && n < capacity; ++i) {
        CODAE_HIP_CHECK(hipEventSynchronize(h->prof_stop[i]));
        float t = 0.f;
        CODAE_HIP_CHECK(hipEventElapsedTime(&t, h->prof_start[i], h->prof_stop[i]));
        const int c = h->prof_count[i];
        // a group record is reported as `c` launches of elapsed / c each
        for (int k = 0; k < c && n < capacity; ++k) { kinds[n] = h->prof_kind[i]; ms[n] = t / (float)c; ++n; }
    }
    *n_out = n;
    profile_release(h);
    return CODAE_OK;
}

int codae_get_sizes(codae_handle h, codae_sizes* out) {
    CODAE_REQUIRE(h && out, "codae_get_sizes: null argument");
    out->n_param = h->n_param;
    // (+ 64 rows of slack behind the last matrix: a GEMM whose k extent is a padded width reads the weight operand up to 63
    //  elements / rows past a row's / the matrix's end - multiplied by the zero pad columns of the activations, see in_ld)
    out->n_weight = h->prec == CODAE_PREC_BF16 ? h->n_param + (int64_t)64 * h->maxw : 0;
    out->act_bytes = h->act_bytes;
    out->dact_bytes = h->n_dact * h->dact_one;
    out->slab_bytes = 3 * h->slab_bytes;   // side stream: two alternating slab buffers; caller's stream (tail wgrad): the third
    out->bias_part_bytes = h->part_floats * 4;
    out->n_scalars = CODAE_S_COUNT;
    return CODAE_OK;
}

int codae_param_offsets(codae_handle h, int32_t layer, int64_t* w_off, int64_t* b_off, int64_t* shadow_off) {
    CODAE_REQUIRE(h && layer >= 0 && layer < h->L, "codae_param_offsets: layer %d", layer);
    if (w_off) *w_off = h->w_off[layer];
    if (b_off) *b_off = h->b_off[layer];
    if (shadow_off) *shadow_off = h->w_off[layer];
    return CODAE_OK;
}

int codae_sync_shadows(codae_handle h, const codae_buffers* b, void* stream) {
    CODAE_REQUIRE(h && b && b->params, "codae_sync_shadows: null argument");
    if (h->prec != CODAE_PREC_BF16) return CODAE_OK;
    CODAE_REQUIRE(b->shadow_w, "codae_sync_shadows: shadow_w missing");
    int rc = launch_cast_bf16(b->params, reinterpret_cast<bf16_t*>(b->shadow_w), h->n_param, (hipStream_t)stream);
    if (rc) return rc;
    return refresh_transposed(h, b, (hipStream_t)stream);
}

int codae_forward(codae_handle h, const codae_buffers* b, const float* x, float* y, int32_t B, int32_t layer_lo,
                  int32_t layer_hi, int32_t save_for_backward, void* stream) {
    int rc = check_common(h, b, B);
    if (rc) return rc;
    CODAE_REQUIRE(x && y, "codae_forward: null x or y");
    CODAE_REQUIRE(layer_lo >= 0 && layer_lo < layer_hi && layer_hi <= h->L, "codae_forward: layer range [%d, %d)", layer_lo, layer_hi);
    (void)save_for_backward;  // activations always live in the workspace; a later forward overwrites them
    hipStream_t s = (hipStream_t)stream;
    const int rows = h->rows_for(B);
    // ingest x into act[layer_lo] in working precision (no gather, no mask)
    codae_batch in{};
    in.data = x; in.B = B; in.io = h->in[layer_lo];
    rc = launch_gather_corrupt(&in, act_ptr(h, b, layer_lo), h->prec == CODAE_PREC_BF16, s, h->in_ld[layer_lo]);
    if (rc) return rc;
    rc = zero_pad_rows(h, act_ptr(h, b, layer_lo), B, rows, h->in_ld[layer_lo], s);
    if (rc) return rc;
    for (int l = layer_lo; l < layer_hi; ++l) {
        const bool last = (l == layer_hi - 1);
        if (last) {
            // the chain's result goes straight to the caller's fp32 tensor (B rows only)
            rc = run_linear(h, b, l, act_ptr(h, b, l), y, true, B, s);
        } else {
            rc = run_linear(h, b, l, act_ptr(h, b, l), act_ptr(h, b, l + 1), false, rows, s);
        }
        if (rc) return rc;
    }
    return CODAE_OK;
}

int codae_backward(codae_handle h, const codae_buffers* b, const float* dy, float* dx, int32_t B, int32_t layer_lo,
                   int32_t layer_hi, void* stream) {
    int rc = check_common(h, b, B);
    if (rc) return rc;
    CODAE_REQUIRE(dy && b->grads && b->dacts, "codae_backward: null dy / grads / dacts");
    CODAE_REQUIRE(layer_lo >= 0 && layer_lo < layer_hi && layer_hi <= h->L, "codae_backward: layer range [%d, %d)", layer_lo, layer_hi);
    hipStream_t s = (hipStream_t)stream;
    const int rows = h->rows_for(B);
    const int top = layer_hi - 1;
    CODAE_REQUIRE(b->bias_parts, "codae_backward: bias_parts missing");
    // dA_top = dy in working precision; db_top = column sums of dy (per 64-row block here, added up after the chain)
    codae_batch in{};
    in.data = dy; in.B = B; in.io = h->out[top];
    rc = launch_gather_corrupt(&in, dact_ptr(h, b, top), h->prec == CODAE_PREC_BF16, s, h->out_ld[top]);
    if (rc) return rc;
    rc = zero_pad_rows(h, dact_ptr(h, b, top), B, rows, h->out_ld[top], s);
    if (rc) return rc;
    rc = launch_colsum_parts_f32(dy, B, h->out[top], part_ptr(h, b, top), s);
    if (rc) return rc;
    h->parts_pending[top] = (B + 63) / 64;
    return backward_range(h, b, B, layer_lo, layer_hi, dx, false, s);
}

int codae_step_forward_loss(codae_handle h, const codae_buffers* b, const codae_batch* batch, const codae_hyper* hyper,
                            float* out_y, void* stream) {
    CODAE_REQUIRE(batch != nullptr, "codae_step_forward_loss: null batch");
    int rc = check_common(h, b, batch->B);
    if (rc) return rc;
    CODAE_REQUIRE(batch->io == h->in[0] && batch->io == h->out[h->L - 1], "batch.io %d does not match the model (%d -> %d)",
                  batch->io, h->in[0], h->out[h->L - 1]);
    CODAE_REQUIRE(b->grads && b->dacts && b->scalars, "codae_step_forward_loss: grads / dacts / scalars missing");
    CODAE_REQUIRE(b->bias_parts, "codae_step_forward_loss: bias_parts missing");
    if (h->prof_on) ++h->prof_step;
    hipStream_t s = (hipStream_t)stream;
    const int B = batch->B, L = h->L;
    const int rows = h->rows_for(B);
    const bool bf = h->prec == CODAE_PREC_BF16;
    {
        ProfScope prof(h, CODAE_K_GATHER, s);
        rc = launch_gather_corrupt(batch, act_ptr(h, b, 0), bf, s, h->in_ld[0]);
    }
    if (rc) return rc;
    rc = zero_pad_rows(h, act_ptr(h, b, 0), B, rows, h->in_ld[0], s);
    if (rc) return rc;
    float* y = out_y ? out_y : reinterpret_cast<float*>(act_ptr(h, b, L));
    // bf16 training step: the loss is folded into the last forward GEMM's epilogue (y never stored)
    const bool fuse_loss = bf && hyper != nullptr && out_y == nullptr && !h->cfg.no_fused_loss;
    GroupScope fwd_group(h, CODAE_K_GEMM_FWD, s);       // the plain forward launches of this step, back to back
    for (int l = 0; l < L; ++l) {
        const bool last = (l == L - 1);
        if (last && fuse_loss) {
            fwd_group.close();
            const double n_glob = (double)(hyper->loss_scale_rows > 0.f ? hyper->loss_scale_rows : (float)B) * batch->io;
            GemmBf16 g{};
            g.A = reinterpret_cast<const bf16_t*>(act_ptr(h, b, l)); g.lda = h->in_ld[l]; g.a_mode = OP_KC;
            g.B = reinterpret_cast<const bf16_t*>(b->shadow_w) + h->w_off[l]; g.ldb = h->in[l]; g.b_mode = OP_KC;
            g.C = dact_ptr(h, b, l); g.ldc = h->out_ld[l]; g.c_f32 = 0;
            g.M = rows; g.N = h->out[l]; g.K = h->in_ld[l];
            g.bias = b->params + h->b_off[l]; g.relu = 0; g.split_k = 1;
            g.colsum_part = part_ptr(h, b, l);
            g.loss.enabled = 1; g.loss.data = batch->data; g.loss.row_idx = batch->row_idx; g.loss.mask_id = batch->mask_id;
            g.loss.mask_to_use = batch->mask_to_use; g.loss.nb_run = batch->nb_run; g.loss.run = batch->run;
            g.loss.table = batch->mask_table; g.loss.io = batch->io; g.loss.B = B; g.loss.inv_n = (float)(1.0 / n_glob);
            g.loss.parts = loss_parts_ptr(h, b);
            if (b->shadow_wt != nullptr && l >= 1 && !h->cfg.no_wt && !h->cfg.no_prefetch) {     // the first data gradient's operand
                g.prefetch = reinterpret_cast<const bf16_t*>(b->shadow_wt) + h->w_off[l];
                g.prefetch_bytes = (int64_t)h->in[l] * h->out[l] * 2;
            }
            const int n_loss_parts = gemm_bf16_loss_parts(g);
            CODAE_REQUIRE(n_loss_parts <= h->loss_part_cap, "fused loss: %d workgroups exceed the partial-sum rows (%d)", n_loss_parts, h->loss_part_cap);
            {
                ProfScope prof(h, CODAE_K_LOSS, s);     // last layer + loss in one launch: its own class, not a plain forward GEMM
                rc = gemm_bf16(g, s);
            }
            if (rc) return rc;
            h->parts_pending[l] = gemm_bf16_colsum_rows(g);
            h->norm_scalars_zero = true;
            return launch_finish_loss(b->scalars, 1.0 / ((double)B * batch->io), s, loss_parts_ptr(h, b), n_loss_parts);
        }
        rc = last ? run_linear(h, b, l, act_ptr(h, b, l), y, true, B, s)
                  : run_linear(h, b, l, act_ptr(h, b, l), act_ptr(h, b, l + 1), false, rows, s);
        if (rc) return rc;
        fwd_group.launched();
    }
    fwd_group.close();
    if (hyper != nullptr) {
        const double n_glob = (double)(hyper->loss_scale_rows > 0.f ? hyper->loss_scale_rows : (float)B) * batch->io;
        rc = zero_pad_rows(h, dact_ptr(h, b, L - 1), B, rows, h->out_ld[L - 1], s);
        if (rc) return rc;
        {
            ProfScope prof(h, CODAE_K_LOSS, s);
            rc = launch_mse_loss(batch, y, dact_ptr(h, b, L - 1), bf, (float)(1.0 / n_glob), part_ptr(h, b, L - 1),
                                 loss_parts_ptr(h, b), 1, s, h->out_ld[L - 1]);
        }
        if (rc) return rc;
        h->parts_pending[L - 1] = mse_loss_colsum_rows(B);
        h->norm_scalars_zero = true;
        return launch_finish_loss(b->scalars, 1.0 / ((double)B * batch->io), s, loss_parts_ptr(h, b), mse_loss_colsum_rows(B));
    }
    rc = launch_mse_loss(batch, y, nullptr, 0, 0.f, nullptr, loss_parts_ptr(h, b), 0, s);
    if (rc) return rc;
    return launch_finish_loss(b->scalars, 1.0 / ((double)B * batch->io), s, loss_parts_ptr(h, b), mse_loss_colsum_rows(B));
}

int codae_eval_step(codae_handle h, const codae_buffers* b, const codae_batch* batch, float* out_y, void* stream) {
    return codae_step_forward_loss(h, b, batch, nullptr, out_y, stream);
}

int codae_step_backward(codae_handle h, const codae_buffers* b, int32_t B, int32_t layer_lo, int32_t layer_hi, void* stream) {
    int rc = check_common(h, b, B);
    if (rc) return rc;
    CODAE_REQUIRE(b->grads && b->dacts, "codae_step_backward: grads / dacts missing");
    CODAE_REQUIRE(layer_lo >= 0 && layer_lo < layer_hi && layer_hi <= h->L, "codae_step_backward: layer range [%d, %d)", layer_lo, layer_hi);
    return backward_range(h, b, B, layer_lo, layer_hi, nullptr, true, (hipStream_t)stream);
}

int codae_step_backward_async(codae_handle h, const codae_buffers* b, int32_t B, int32_t layer_lo, int32_t layer_hi, void* stream) {
    int rc = check_common(h, b, B);
    if (rc) return rc;
    CODAE_REQUIRE(b->grads && b->dacts, "codae_step_backward_async: grads / dacts missing");
    CODAE_REQUIRE(layer_lo >= 0 && layer_lo < layer_hi && layer_hi <= h->L, "codae_step_backward_async: layer range [%d, %d)", layer_lo, layer_hi);
    return backward_range(h, b, B, layer_lo, layer_hi, nullptr, true, (hipStream_t)stream, false);
}

int codae_side_stream(codae_handle h, void** out) {
    CODAE_REQUIRE(h && out, "codae_side_stream: null argument");
    *out = nullptr;
    if (h->cfg.single_stream) return CODAE_OK;      // everything runs on the caller's stream
    int rc = ensure_side_stream(h);
    if (rc) return rc;
    *out = h->side;
    return CODAE_OK;
}

static int update_impl(codae_handle h, const codae_buffers* b, const codae_hyper* hyper, hipStream_t s, bool weights_norm_done) {
    CODAE_REQUIRE(h && b && hyper, "codae_step_update: null argument");
    CODAE_REQUIRE(b->params && b->grads && b->adam_m && b->adam_v && b->scalars, "codae_step_update: buffer missing");
    {
        int rcw = join_side(h, s);            // (a backward issued with codae_step_backward_async)
        if (rcw) return rcw;
    }
    {
        bool pending = false;
        for (int l = 0; l < h->L; ++l) pending = pending || h->parts_pending[l] > 0;
        if (pending && b->bias_parts != nullptr) {           // (a bucketed backward that did not reach layer 0)
            int rcb = finish_bias(h, b, s, false);
            if (rcb) return rcb;
        }
    }
    const bool scalars_zero = h->norm_scalars_zero;
    h->norm_scalars_zero = false;
    (void)scalars_zero;
    if (hyper->max_grad_norm > 0.f && !weights_norm_done) {    // (else: sum g^2 was accumulated by the slab reduces / the
        ProfScope prof(h, CODAE_K_SUMSQ, s);                    //  grouped weight-gradient epilogues and the bias finish)
        int rc;
        {
            if (!scalars_zero) {                 // (a stand-alone update: no step_forward_loss of this step cleared them)
                CODAE_HIP_CHECK(hipMemsetAsync(b->scalars + CODAE_S_GRAD_SQ, 0, sizeof(double), s));
                CODAE_HIP_CHECK(hipMemsetAsync(b->scalars + CODAE_S_GRAD_SQ_SLOTS, 0, CODAE_S_N_SLOTS * sizeof(double), s));
            }
            rc = launch_sumsq(b->grads, h->n_param, b->scalars + CODAE_S_GRAD_SQ, s);
        }
        if (rc) return rc;
    }
    bf16_t* shadow = h->prec == CODAE_PREC_BF16 ? reinterpret_cast<bf16_t*>(b->shadow_w) : nullptr;
    CODAE_REQUIRE(h->prec != CODAE_PREC_BF16 || shadow, "codae_step_update: shadow_w missing");
    // (Measured and dropped: per-layer Adam kernels on the side stream beside the NEXT forward - HBM / L2 contention
    // slowed those GEMMs from 46.8 to 56.9 us each and the step from 1.76 to 1.84 ms.)
    const double* step_dev = h->capturing ? b->scalars + CODAE_S_ADAM_STEP : nullptr;
    ProfScope prof(h, CODAE_K_ADAM, s);
    if (h->prec == CODAE_PREC_BF16 && b->shadow_wt != nullptr && h->L <= 64 && !h->cfg.flat_adam) {
        // one tiled pass: p, m, v, the bf16 shadow and the transposed shadow of every layer that has a data gradient
        return launch_clip_adam_tiled(b->params, b->grads, b->adam_m, b->adam_v, hyper, b->scalars + CODAE_S_GRAD_SQ, shadow,
                                      reinterpret_cast<bf16_t*>(b->shadow_wt), h->L, h->w_off.data(), h->out.data(), h->in.data(),
                                      1, h->bias_begin, h->n_param - h->bias_begin, s, step_dev);
    }
    int rca = launch_clip_adam(b->params, b->grads, b->adam_m, b->adam_v, h->n_param, hyper, b->scalars + CODAE_S_GRAD_SQ,
                               shadow, nullptr, s, step_dev);
    if (rca) return rca;
    return refresh_transposed(h, b, s);
}

int codae_step_update(codae_handle h, const codae_buffers* b, const codae_hyper* hyper, void* stream) {
    return update_impl(h, b, hyper, (hipStream_t)stream, false);
}

int codae_span_sumsq(const float* g, int64_t n, double* acc, void* stream) {
    return launch_sumsq_to(g, n, acc, (hipStream_t)stream);
}

int codae_step_update_span(codae_handle h, const codae_buffers* b, const codae_hyper* hyper, int64_t lo, int64_t hi,
                           const double* total_sq, void* stream) {
    CODAE_REQUIRE(h && b && hyper, "codae_step_update_span: null argument");
    CODAE_REQUIRE(b->params && b->grads && b->adam_m && b->adam_v && b->scalars, "codae_step_update_span: buffer missing");
    CODAE_REQUIRE(lo >= 0 && lo < hi && hi <= h->n_param && lo % 4 == 0 && hi % 4 == 0, "codae_step_update_span: range [%lld, %lld)",
                  (long long)lo, (long long)hi);
    CODAE_REQUIRE(hyper->max_grad_norm <= 0.f || total_sq != nullptr, "codae_step_update_span: clipping needs the global sum g^2");
    hipStream_t s = (hipStream_t)stream;
    int rc = join_side(h, s);
    if (rc) return rc;
    h->norm_scalars_zero = false;
    const double* coef = nullptr;
    if (hyper->max_grad_norm > 0.f) {
        rc = launch_clip_coef(total_sq, hyper->max_grad_norm, b->scalars + CODAE_S_CLIP_COEF, s);
        if (rc) return rc;
        coef = b->scalars + CODAE_S_CLIP_COEF;
    }
    bf16_t* shadow = h->prec == CODAE_PREC_BF16 ? reinterpret_cast<bf16_t*>(b->shadow_w) : nullptr;
    CODAE_REQUIRE(h->prec != CODAE_PREC_BF16 || shadow, "codae_step_update_span: shadow_w missing");
    ProfScope prof(h, CODAE_K_ADAM, s);
    return launch_clip_adam(b->params + lo, b->grads + lo, b->adam_m + lo, b->adam_v + lo, hi - lo, hyper, nullptr,
                            shadow ? shadow + lo : nullptr, coef, s);
}

int codae_sync_transposed(codae_handle h, const codae_buffers* b, void* stream) {
    CODAE_REQUIRE(h && b, "codae_sync_transposed: null argument");
    return refresh_transposed(h, b, (hipStream_t)stream);
}

int codae_join(codae_handle h, void* stream) {
    CODAE_REQUIRE(h != nullptr, "codae_join: null handle");
    return join_side(h, (hipStream_t)stream);
}

int codae_step_path(codae_handle h, const codae_buffers* b, int32_t B) {
    return (h != nullptr && b != nullptr && B > 0 && B <= h->max_batch && chain_eligible(h, b, B)) ? 1 : 0;
}

int codae_train_step(codae_handle h, const codae_buffers* b, const codae_batch* batch, const codae_hyper* hyper, void* stream) {
    CODAE_REQUIRE(hyper != nullptr, "codae_train_step: null hyper");
    CODAE_REQUIRE(batch != nullptr, "codae_train_step: null batch");
    if (h != nullptr && b != nullptr && chain_eligible(h, b, batch->B)) {
        // narrow stack: persistent fused chain + grouped weight gradients (6 launches for the whole step)
        int rcc = check_common(h, b, batch->B);
        if (rcc) return rcc;
        CODAE_REQUIRE(batch->io == h->in[0], "batch.io %d does not match the model (%d)", batch->io, h->in[0]);
        CODAE_REQUIRE(b->grads && b->dacts && b->scalars && b->bias_parts, "codae_train_step: grads / dacts / scalars / bias_parts missing");
        CODAE_REQUIRE(batch->data && batch->B > 0 && (!(batch->mask_id || batch->mask_to_use) || batch->mask_table), "codae_train_step: bad batch");
        CODAE_REQUIRE(!batch->mask_to_use || batch->mask_id || (batch->nb_run > 0 && batch->run >= 0 && batch->run < batch->nb_run),
                      "codae_train_step: run %d outside [0, %d)", batch->run, batch->nb_run);
        if (h->prof_on) ++h->prof_step;
        hipStream_t s = (hipStream_t)stream;
        rcc = join_side(h, s);
        if (rcc) return rcc;
        LossFinish lf{};
        rcc = run_chain(h, b, batch, hyper, true, s, &lf);
        if (rcc) return rcc;
        const bool with_norm = hyper->max_grad_norm > 0.f && !h->cfg.no_fused_norm;     // (finish_loss zeroed the norm slots)
        rcc = run_wgrad_grouped(h, b, h->rows_for(batch->B), with_norm, s);
        if (rcc) return rcc;
        rcc = finish_bias(h, b, s, with_norm, &lf);
        if (rcc) return rcc;
        return update_impl(h, b, hyper, s, with_norm);
    }
    int rc = codae_step_forward_loss(h, b, batch, hyper, nullptr, stream);
    if (rc) return rc;
    // single GPU: nothing happens to the gradients between backward and update, so the norm can be
    // gathered while the split-K slabs are reduced (finish_loss zeroed GRAD_SQ before the backward)
    // bf16: every weight gradient leaves its sum g^2 behind - split ones in the slab reduce, unsplit ones in the GEMM epilogue
    const bool all_slabbed = h->prec == CODAE_PREC_BF16 && hyper->max_grad_norm > 0.f && !h->cfg.no_fused_norm;
    h->norm_in_backward = all_slabbed;
    rc = codae_step_backward(h, b, batch->B, 0, h->L, stream);
    h->norm_in_backward = false;
    if (rc) return rc;
    return update_impl(h, b, hyper, (hipStream_t)stream, all_slabbed);
}

static bool same_bytes(const void* a, const void* b, size_t n) { return memcmp(a, b, n) == 0; }

int codae_train_step_graph(codae_handle h, const codae_buffers* b, const codae_batch* batch, const codae_hyper* hyper, void* stream) {
    CODAE_REQUIRE(h && b && batch && hyper, "codae_train_step_graph: null argument");
    CODAE_REQUIRE(!h->prof_on, "codae_train_step_graph: launch profiling (codae_profile_begin) cannot run inside a graph");
    CODAE_REQUIRE(b->scalars != nullptr, "codae_train_step_graph: scalars missing");
    CODAE_REQUIRE(stream != nullptr, "codae_train_step_graph: the default (null) stream cannot be captured - pass a created stream");
    hipStream_t s = (hipStream_t)stream;
    // what the captured kernel arguments depend on (the step count is the one thing that may change)
    codae_hyper hk = *hyper;
    hk.step = 0;
    const bool fresh = h->graph_exec == nullptr || !same_bytes(&h->graph_key.batch, batch, sizeof(*batch)) ||
                       !same_bytes(&h->graph_key.hyper, &hk, sizeof(hk)) || !same_bytes(&h->graph_key.bufs, b, sizeof(*b));
    if (fresh) {
        if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
        int rc = check_common(h, b, batch->B);
        if (rc) return rc;
        if (!h->cfg.single_stream) {
            rc = ensure_side_stream(h);                      // (no stream / event creation inside the capture)
            if (rc) return rc;
        }
        rc = codae_join(h, stream);                          // nothing recorded outside may be waited for inside
        if (rc) return rc;
        CODAE_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        h->capturing = true;
        rc = codae_train_step(h, b, batch, hyper, stream);
        h->capturing = false;
        hipGraph_t graph = nullptr;
        const hipError_t ee = hipStreamEndCapture(s, &graph);
        if (rc != CODAE_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (ee != hipSuccess || graph == nullptr) {
            set_error("codae_train_step_graph: stream capture failed: %s", hipGetErrorString(ee));
            return CODAE_E_HIP;
        }
        const hipError_t ei = hipGraphInstantiate(&h->graph_exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ei != hipSuccess) {
            h->graph_exec = nullptr;
            set_error("codae_train_step_graph: hipGraphInstantiate failed: %s", hipGetErrorString(ei));
            return CODAE_E_HIP;
        }
        h->graph_key.batch = *batch; h->graph_key.hyper = hk; h->graph_key.bufs = *b;
    }
    int rc = launch_set_scalar(b->scalars + CODAE_S_ADAM_STEP, (double)hyper->step, s);
    if (rc) return rc;
    CODAE_HIP_CHECK(hipGraphLaunch(h->graph_exec, s));
    return CODAE_OK;
}

// ---- data parallel with a library-owned RCCL communicator --------------------------------

int codae_dp_unique_id(void* out, int32_t capacity) {
    CODAE_REQUIRE(out != nullptr && capacity >= (int32_t)sizeof(ncclUniqueId), "codae_dp_unique_id: need %d bytes", (int)sizeof(ncclUniqueId));
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    CODAE_RCCL_CHECK(g_rccl.GetUniqueId(&id));
    memcpy(out, &id, sizeof(id));
    return CODAE_OK;
}

int codae_dp_init(codae_handle h, const void* unique_id, int32_t rank, int32_t world) {
    CODAE_REQUIRE(h != nullptr && unique_id != nullptr && world >= 1 && rank >= 0 && rank < world, "codae_dp_init: bad arguments");
    CODAE_REQUIRE(h->dp == nullptr, "codae_dp_init: this engine already has a communicator");
    int rc = rccl_load();
    if (rc) return rc;
    DpState* d = new DpState();
    d->rank = rank; d->world = world;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&d->comm, world, id, rank);          // (collective: every rank calls it)
    if (r != ncclSuccess) { set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); delete d; return CODAE_E_HIP; }
    int prio_least = 0, prio_greatest = 0;
    hipError_t e1 = hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (e1 == hipSuccess) e1 = hipStreamCreateWithPriority(&d->stream, hipStreamNonBlocking, prio_greatest);
    for (int i = 0; i <= CODAE_DP_MAX_BUCKETS && e1 == hipSuccess; ++i) e1 = hipEventCreateWithFlags(&d->ev_bucket[i], hipEventDisableTiming);
    if (e1 == hipSuccess) e1 = hipEventCreateWithFlags(&d->ev_done, hipEventDisableTiming);
    if (e1 != hipSuccess) {
        set_error("codae_dp_init: stream / event creation failed: %s", hipGetErrorString(e1));
        (void)g_rccl.CommDestroy(d->comm);
        delete d;
        return CODAE_E_HIP;
    }
    h->dp = d;
    return CODAE_OK;
}

int codae_dp_destroy(codae_handle h) {
    if (h == nullptr || h->dp == nullptr) return CODAE_OK;
    DpState* d = h->dp;
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    if (d->comm) (void)g_rccl.CommDestroy(d->comm);
    for (int i = 0; i <= CODAE_DP_MAX_BUCKETS; ++i) if (d->ev_bucket[i]) (void)hipEventDestroy(d->ev_bucket[i]);
    if (d->ev_done) (void)hipEventDestroy(d->ev_done);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    delete d;
    h->dp = nullptr;
    return CODAE_OK;
}

// One data-parallel optimizer step, collectives included: forward + loss; the backward bucket by bucket (layer ranges
// [lo_i, hi_i), top down, ending at layer 0), each bucket's weight gradients all-reduced (SUM, in place, fp32) on the
// communicator's own stream as soon as they exist on the engine's side stream - no host round trip between buckets -; the
// bias block last; then the caller's stream waits ONCE for the collectives and runs clip + Adam on the summed gradients.
// hyper->loss_scale_rows must hold the GLOBAL batch rows (so that the SUM is the global-batch mean gradient).
int codae_train_step_dp(codae_handle h, const codae_buffers* b, const codae_batch* batch, const codae_hyper* hyper, int32_t n_buckets,
                        const int32_t* bucket_lo, const int32_t* bucket_hi, void* stream) {
    CODAE_REQUIRE(h != nullptr && h->dp != nullptr, "codae_train_step_dp: codae_dp_init has not been called on this engine");
    CODAE_REQUIRE(b && batch && hyper && bucket_lo && bucket_hi, "codae_train_step_dp: null argument");
    CODAE_REQUIRE(n_buckets >= 1 && n_buckets <= CODAE_DP_MAX_BUCKETS, "codae_train_step_dp: %d buckets", n_buckets);
    CODAE_REQUIRE(bucket_hi[0] == h->L && bucket_lo[n_buckets - 1] == 0, "codae_train_step_dp: buckets must run from the top layer down to layer 0");
    for (int i = 0; i < n_buckets; ++i) {
        CODAE_REQUIRE(bucket_lo[i] >= 0 && bucket_lo[i] < bucket_hi[i] && (i == 0 || bucket_hi[i] == bucket_lo[i - 1]),
                      "codae_train_step_dp: bucket %d = [%d, %d) does not continue the one above it", i, bucket_lo[i], bucket_hi[i]);
    }
    DpState* d = h->dp;
    hipStream_t s = (hipStream_t)stream;
    int rc = codae_step_forward_loss(h, b, batch, hyper, nullptr, stream);
    if (rc) return rc;
    for (int i = 0; i < n_buckets; ++i) {
        rc = backward_range(h, b, batch->B, bucket_lo[i], bucket_hi[i], nullptr, true, s, false);
        if (rc) return rc;
        // the bucket's weight gradients are complete, in order, on the side stream (or on s itself in single-stream mode)
        hipStream_t producer = h->side != nullptr && !h->cfg.single_stream ? h->side : s;
        CODAE_HIP_CHECK(hipEventRecord(d->ev_bucket[i], producer));
        CODAE_HIP_CHECK(hipStreamWaitEvent(d->stream, d->ev_bucket[i], 0));
        const int64_t lo_off = h->w_off[bucket_lo[i]];
        const int64_t hi_off = bucket_hi[i] < h->L ? h->w_off[bucket_hi[i]] : h->bias_begin;
        float* span = b->grads + lo_off;
        CODAE_RCCL_CHECK(g_rccl.AllReduce(span, span, (size_t)(hi_off - lo_off), ncclFloat, ncclSum, d->comm, d->stream));
    }
    rc = join_side(h, s);
    if (rc) return rc;
    // (the last range finished every bias gradient on s)
    CODAE_HIP_CHECK(hipEventRecord(d->ev_bucket[n_buckets], s));
    CODAE_HIP_CHECK(hipStreamWaitEvent(d->stream, d->ev_bucket[n_buckets], 0));
    float* bias = b->grads + h->bias_begin;
    CODAE_RCCL_CHECK(g_rccl.AllReduce(bias, bias, (size_t)(h->n_param - h->bias_begin), ncclFloat, ncclSum, d->comm, d->stream));
    CODAE_HIP_CHECK(hipEventRecord(d->ev_done, d->stream));
    CODAE_HIP_CHECK(hipStreamWaitEvent(s, d->ev_done, 0));
    return update_impl(h, b, hyper, s, false);
}

// ---- stand-alone ops --------------------------------------------------------------------

int codae_corrupt(const float* x, const float* mask, float* out, int64_t n, void* stream) {
    return launch_corrupt(x, mask, out, n, (hipStream_t)stream);
}

int codae_expand_masks(const int32_t* mask_id, const uint8_t* mask_table, const int32_t* k_of_mask, int32_t B, int32_t io,
                       int32_t k_max, float* masks_out, float* fmask_out, void* stream) {
    return launch_expand_masks(mask_id, mask_table, k_of_mask, B, io, k_max, masks_out, fmask_out, (hipStream_t)stream);
}

int codae_mse_loss_fwd_bwd(const float* x, const float* y, const float* fmask, float* dy, int64_t n, float inv_n,
                           double* scalars, void* stream) {
    int rc = launch_mse_dense(x, y, fmask, dy, n, inv_n, scalars, (hipStream_t)stream);
    if (rc) return rc;
    return launch_finish_loss(scalars, 1.0 / (double)n, (hipStream_t)stream);
}

int codae_clip_adam(float* params, float* grads, float* adam_m, float* adam_v, int64_t n, const codae_hyper* hyper,
                    double* scalars, void* stream) {
    CODAE_REQUIRE(hyper && scalars, "codae_clip_adam: null argument");
    hipStream_t s = (hipStream_t)stream;
    if (hyper->max_grad_norm > 0.f) {
        CODAE_HIP_CHECK(hipMemsetAsync(scalars + CODAE_S_GRAD_SQ, 0, sizeof(double), s));
        CODAE_HIP_CHECK(hipMemsetAsync(scalars + CODAE_S_GRAD_SQ_SLOTS, 0, CODAE_S_N_SLOTS * sizeof(double), s));
        int rc = launch_sumsq(grads, n, scalars + CODAE_S_GRAD_SQ, s);
        if (rc) return rc;
    }
    return launch_clip_adam(params, grads, adam_m, adam_v, n, hyper, scalars + CODAE_S_GRAD_SQ, nullptr, nullptr, s);
}

// ---- GEMM primitives ------------------------------------------------------------------------

int codae_linear_f32(const float* x, const float* W, const float* bias, float* y, int32_t M, int32_t N, int32_t K,
                     int32_t relu, void* stream) {
    CODAE_REQUIRE(x && W && y, "codae_linear_f32: null operand");
    GemmF32 g{};
    g.A = x; g.a_rs = K; g.a_ks = 1;
    g.B = W; g.b_rs = K; g.b_ks = 1;
    g.C = y; g.ldc = N; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.relu = relu;
    return gemm_f32(g, (hipStream_t)stream);
}

int codae_dgrad_f32(const float* dy, const float* W, const float* relu_src, float* dx, int32_t M, int32_t N, int32_t K,
                    void* stream) {
    CODAE_REQUIRE(dy && W && dx, "codae_dgrad_f32: null operand");
    GemmF32 g{};
    g.A = dy; g.a_rs = N; g.a_ks = 1;
    g.B = W; g.b_rs = 1; g.b_ks = K;
    g.C = dx; g.ldc = K; g.M = M; g.N = K; g.K = N;
    g.relu_src = relu_src; g.ld_relu = K;
    return gemm_f32(g, (hipStream_t)stream);
}

int codae_wgrad_f32(const float* dy, const float* x, float* dW, float* db, int32_t M, int32_t N, int32_t K, void* stream) {
    CODAE_REQUIRE(dy && x && dW, "codae_wgrad_f32: null operand");
    GemmF32 g{};
    g.A = dy; g.a_rs = 1; g.a_ks = N;
    g.B = x; g.b_rs = 1; g.b_ks = K;
    g.C = dW; g.ldc = K; g.M = N; g.N = K; g.K = M;
    int rc = gemm_f32(g, (hipStream_t)stream);
    if (rc) return rc;
    if (db) return launch_colsum_f32(dy, M, N, db, (hipStream_t)stream);
    return CODAE_OK;
}

int codae_linear_bf16(const void* x, const void* W, const float* bias, void* y, int32_t y_f32, int32_t M, int32_t N,
                      int32_t K, int32_t relu, void* stream) {
    CODAE_REQUIRE(x && W && y, "codae_linear_bf16: null operand");
    GemmBf16 g{};
    g.A = reinterpret_cast<const bf16_t*>(x); g.lda = K; g.a_mode = OP_KC;
    g.B = reinterpret_cast<const bf16_t*>(W); g.ldb = K; g.b_mode = OP_KC;
    g.C = y; g.ldc = N; g.c_f32 = y_f32; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.relu = relu; g.split_k = 1;
    return gemm_bf16(g, (hipStream_t)stream);
}

int codae_dgrad_bf16(const void* dy, const void* W, const void* relu_src, void* dx, float* db_prev, float* db_ws, int32_t M,
                     int32_t N, int32_t K, void* stream) {
    CODAE_REQUIRE(dy && W && dx, "codae_dgrad_bf16: null operand");
    CODAE_REQUIRE(db_prev == nullptr || db_ws != nullptr, "codae_dgrad_bf16: db_prev needs the db_ws scratch");
    GemmBf16 g{};
    g.A = reinterpret_cast<const bf16_t*>(dy); g.lda = N; g.a_mode = OP_KC;
    g.B = reinterpret_cast<const bf16_t*>(W); g.ldb = K; g.b_mode = OP_KS;
    g.C = dx; g.ldc = K; g.c_f32 = 0; g.M = M; g.N = K; g.K = N;
    g.relu_src = reinterpret_cast<const bf16_t*>(relu_src); g.ld_relu = K;
    g.colsum_part = db_prev ? db_ws : nullptr; g.split_k = 1;
    int rc = gemm_bf16(g, (hipStream_t)stream);
    if (rc || db_prev == nullptr) return rc;
    BiasFinishJobs jobs;
    jobs.n = 1; jobs.parts[0] = db_ws; jobs.out[0] = db_prev; jobs.rows[0] = gemm_bf16_colsum_rows(g); jobs.cols[0] = K;
    jobs.col_begin[0] = 0; jobs.col_begin[1] = K;
    return launch_bias_finish(jobs, nullptr, (hipStream_t)stream);
}

int codae_wgrad_bf16(const void* dy, const void* x, float* dW, void* slabs, int64_t slab_bytes, int32_t M, int32_t N,
                     int32_t K, void* stream) {
    CODAE_REQUIRE(dy && x && dW, "codae_wgrad_bf16: null operand");
    CODAE_REQUIRE(M % 64 == 0, "codae_wgrad_bf16: batch rows %d must be a multiple of 64 (pad with zero rows)", M);
    int S = choose_split_k(N, K, M);
    while (S > 1 && (slabs == nullptr || (int64_t)S * N * K * 4 > slab_bytes)) --S;
    GemmBf16 g{};
    g.A = reinterpret_cast<const bf16_t*>(dy); g.lda = N; g.a_mode = OP_KS;
    g.B = reinterpret_cast<const bf16_t*>(x); g.ldb = K; g.b_mode = OP_KS;
    g.ldc = K; g.c_f32 = 1; g.M = N; g.N = K; g.K = M; g.split_k = S;
    g.C = S > 1 ? slabs : (void*)dW;
    int rc = gemm_bf16(g, (hipStream_t)stream);
    if (rc) return rc;
    if (S > 1) return launch_reduce_slabs(reinterpret_cast<const float*>(slabs), S, (int64_t)N * K, dW, (int64_t)N * K, nullptr, (hipStream_t)stream);
    return CODAE_OK;
}

int codae_debug_gemm_timeline(uint64_t* host_out, int32_t n_wg) {
    return gemm_bf16_timeline(reinterpret_cast<unsigned long long*>(host_out), n_wg);
}

int codae_transpose_bf16(const void* src, void* dst, int32_t rows, int32_t cols, void* stream) {
    CODAE_REQUIRE(src && dst && rows > 0 && cols > 0, "transpose: bad args");
    const int64_t off = 0;
    return launch_transpose_bf16(reinterpret_cast<const bf16_t*>(src), reinterpret_cast<bf16_t*>(dst), 1, &off, &rows, &cols,
                                 (hipStream_t)stream);
}

int codae_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    return launch_cast_bf16(src, reinterpret_cast<bf16_t*>(dst), n, (hipStream_t)stream);
}

}  // extern "C"
