// jwas_hip.hip -- context, memory and the C ABI of libjwas_hip.so (see include/jwas_hip.h).
// gfx950 only.  No CPU fallback: every entry point either runs the HIP path or returns an error.
#include "../../include/jwas_hip.h"
#include "sweep.hpp"
#include "f64_path.hpp"
#include "step_launch.hpp"

#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <mutex>
#include <atomic>
#include <condition_variable>
#include <chrono>

using namespace jw;

static_assert(JWAS_HIP_MAX_TRAITS == kMaxT, "trait limit mismatch");
static_assert(JWAS_HIP_MAX_STATES == kMaxStates, "state limit mismatch");

struct jwas_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    int64_t n = 0, p = 0, ld = 0;
    int nslices = 0;                    // 256-row slices
    int upd_nslices = 0;                // slices of the UPDATE role: = nslices (dense), 1024-row slices on 2-bit packed storage (update_role_wide)
    int nrg = 0, ncg = 1;               // k_update_partial grid: row groups x column groups
    int spg = 8;                        // slices per row group
    float* X = nullptr;                 // dense fp32 storage ...
    uint8_t* Q = nullptr;               // ... or the reference's 2-bit packed storage [p][ld/4] + per-marker means
    float* qmean = nullptr;
    bool packed = false;
    int centered = 1;
    float* w = nullptr;                 // [ld] residual weights R^-1 (ones unless jwas_hip_set_weights; pad rows 0)
    bool weighted = false;

    // Active block configuration (a view of one entry of `sets`; several block sizes can be resident so the host
    // can pick per sweep: big blocks when few markers change, smaller ones when many do).
    struct BlockSet {
        int bs; int64_t nblocks; float *gram, *cross, *corr; double* partials;
        // grouped launches (jwas_hip_setup_groups; k_group_step): gm = 2 or 4 blocks per launch (0: not set up); gcross[0]: cross-Grams
        // of consecutive PAIRS of blocks (2 bs markers: pair q at q (2 bs)^2, rows = markers of pair q-1), gcross[1] (gm = 4): of
        // consecutive groups of four; gcbuf: the corrections [2][gm bs] cG | [2][bs] cW | [2 bs] cP | [bs] zeros; gidx / gdelta:
        // [2][gm bs] the merged change lists of a group (ping-pong; header lines: ctx.ev[parity])
        int gm; float* gcross[2]; float* gcbuf; int32_t* gidx; float* gdelta;
        unsigned long long* gpp;        // tagged hand-over words of the ping-pong samplers (SamplerArgs::pp_*, GroupArgs::pp_*): (6 + 3 gm) bs + 8
    };
    unsigned pp_epoch = 0;              // tag of the last ping-pong launch (31 bits, never 0: a word of the zeroed buffer matches no launch)
    int set_index = 0;                  // entry of `sets` that is selected
    std::vector<BlockSet> sets;
    std::vector<int64_t> starts;        // explicit block starts (nblocks + 1 entries, last = p), empty = uniform blocks
    int64_t* d_starts = nullptr;        // ... on the device
    int block_size = 0;
    int64_t nblocks = 0;
    float* xpx = nullptr;
    float* gram = nullptr;
    float* cross = nullptr;             // cross-Grams X_{b-1}'X_b, block b at offset b*bs*bs (block 0 unused)
    float* corr = nullptr;              // [2][kMaxT][bs] lookahead corrections (ping-pong: read by launch k, written for k+1)

    int method = -1, ntraits = 0;
    float* r = nullptr;                 // [2][kMaxT][ld] ping-pong; buffer 0 is current between sweeps
    float *alpha = nullptr, *beta = nullptr;
    void* delta = nullptr;
    float *mean_a = nullptr, *mean_a2 = nullptr, *mean_d = nullptr;

    double* partials = nullptr;
    Events* ev = nullptr;               // [2]
    // independent-block mode (allocated on first use)
    double* ipartials = nullptr;        // [nblocks][t][nrg][bs]
    Events* ev_all = nullptr;           // [nblocks] per-block change lists
    int32_t* ev_offs = nullptr;         // [nblocks + 1] exclusive scan of the counts; [nblocks] = total
    int32_t* idx_all = nullptr;         // [p] compacted change list, (block, marker) order
    float* delta_all = nullptr;         // [kMaxT][p]
    int ind_traits = 0;
    DevParams* dparams = nullptr;
    unsigned long long* counters = nullptr;
    double* fin_out = nullptr;          // [nslices][kMaxT*kMaxT + kMaxT]
    int* sync_cnt = nullptr;            // [2][nrg] arrival counters of the update role's cooperative dense apply
    double* stat_out = nullptr;         // [kStatGrid][kNStat]
    double* host_buf = nullptr;         // pinned staging for fin_out + stat_out + counters
    double* prep_d = nullptr;           // [kPrepD][p] per-sweep marker constants (k_prepare)
    float*  prep_f = nullptr;           // [kPrepF][p]
    double* mt2_tab = nullptr;          // sampler II, <= 3 traits: [2^t * (t(t+1)/2 + 1)][p] state tables
    float*  tsec = nullptr;             // Rule T (section_solve): the section inverses of the current sweep, [sections][(64 t)^2]
    size_t  tsec_cap = 0;               // ... capacity in floats
    unsigned long long* xch = nullptr;  // Rule T: [kMaxT][256] {value, tag} words: the sampler workgroup's hand-over to the helper workgroup
    int     xch_epoch = 0;              // ... grows by 8 per launch
    float*  Xout = nullptr;             // output (EBV) rows: [p][ld_out] fp32, Mi.output_genotypes (tools4genotypes.jl:290-296)
    int64_t n_out = 0, ld_out = 0;
    float*  var_vec = nullptr;
    float*  var_mat = nullptr;          // p x t x t per-marker effect covariances (multi-trait BayesA/B), uploaded per sweep
    float*  ginv_mat = nullptr;         // their inverses (k_prepare)
    bool    var_mat_resident = false;   // var_mat holds this chain's per-marker covariances (uploaded or drawn on the device)
    double* pi_vec = nullptr;
    double* pi_mat = nullptr;
    double* lpr_mat = nullptr;          // p x 2^t marker-specific multi-trait log priors
    bool    lpr_active = false;         // ... in use by the current sweep
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    int timing_stride = 0;
    double last_events = -1.0;          // effect changes of the previous sweep (-1: none yet)
    unsigned long long last_counters[kNCounters] = {};      // the sampler's diagnostics counters of the previous sweep
    double event_overhead_ms = 0.0;     // mean HIP-event interval around an empty launch (calibration)
    std::vector<hipEvent_t> kev;        // pairs of events around sampled k_update_partial launches
    // marker-shard reconcile (jwas_hip_comm_init / jwas_hip_sweep_sharded): RCCL communicator on this context's device
    int32_t* cmp_idx = nullptr;         // [p] + [1] compacted nonzero effects of one trait (k_compact_alpha)
    float* cmp_val = nullptr;           // [p]
    void* comm = nullptr;               // ncclComm_t
    int comm_rank = 0, comm_world = 1;
    float* r_snap = nullptr;            // [kMaxT][ld] residual snapshot of the running sweep
    double* shard_buf = nullptr;        // [kMaxT*ld + kShardStats] delta r (fp64) + packed marker statistics: ONE all-reduce
    // exact ROW shards (jwas_hip_comm_row_shards): this context holds a slice of the individuals and ALL markers; x'x, the
    // Grams and every block's partial RHS are summed over the ranks, the sampler then runs replicated
    bool row_mode = false;
    int loop_slot = -1;                 // >= 0: loopback transport (ranks = contexts of one process on different host threads)
    double* row_buf = nullptr;          // [32] small exchanges
    // Float64 mode (runMCMC(double_precision=true); csrc/f64_path.hpp): its own storage / state, created by jwas_hip_set_precision
    struct F64 {
        double* X = nullptr;                // [p][ld]
        double* r = nullptr;                // [kMaxT][ld]
        double* xpx = nullptr;              // [p]
        double* gram = nullptr;             // [nblocks][bs][bs]
        double *alpha = nullptr, *beta = nullptr;      // [t][p]
        void* delta = nullptr;              // double [t][p], or int32 [p] (BayesR classes)
        double *mean_a = nullptr, *mean_a2 = nullptr, *mean_d = nullptr;
        double* partials = nullptr;         // [kMaxT][nslices][bstride]  (independent blocks: one such set per block)
        size_t partials_cap = 0;            // ... in doubles
        jw64::Events64* ev = nullptr;       // [2]  (independent blocks: ev_all, one per block)
        jw64::Events64* ev_all = nullptr;
        int64_t ev_all_cap = 0;
        jw64::Params64* dparams = nullptr;
        double* var_vec = nullptr;          // [p] BayesB
        double* w = nullptr;                // [ld] residual weights R^-1 (pad rows 0; ones when unweighted)
        std::vector<int64_t> starts;        // block starts (nblocks + 1 entries, 0-based): uniform or explicit partition
        int bstride = 0;                    // largest block of the partition, rounded up to a multiple of 8
        bool explicit_part = false;
    };
    F64* f64 = nullptr;
};

static constexpr int kStatGrid = 128;
static int row_allreduce(jwas_hip_ctx* c, void* dev, size_t count, bool f64);      // exact row shards: sum over the ranks
static thread_local std::string g_create_error;

static int fail(jwas_hip_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail((ctx), JWAS_HIP_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                       \
    } while (0)

#define NEED(ctx, cond, code, ...)                                                                 \
    do { if (!(cond)) return fail((ctx), (code), __VA_ARGS__); } while (0)

static int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// t x t inverse: double Gauss-Jordan with partial pivoting, rounded to float (stands in for Julia's
// inv(::Matrix{Float32}), MTBayesABC.jl:66-67).  Same operation sequence as the oracle's.
static int inv_small(const float* A, int t, float* Ainv)
{
    double M[kMaxT][2 * kMaxT];
    for (int i = 0; i < t; ++i)
        for (int j = 0; j < t; ++j) { M[i][j] = A[i * t + j]; M[i][t + j] = (i == j); }
    for (int c = 0; c < t; ++c) {
        int piv = c;
        for (int i = c + 1; i < t; ++i) if (std::fabs(M[i][c]) > std::fabs(M[piv][c])) piv = i;
        if (M[piv][c] == 0.0) return -1;
        if (piv != c) for (int j = 0; j < 2 * t; ++j) { double tmp = M[c][j]; M[c][j] = M[piv][j]; M[piv][j] = tmp; }
        const double d = M[c][c];
        for (int j = 0; j < 2 * t; ++j) M[c][j] /= d;
        for (int i = 0; i < t; ++i) if (i != c) {
            const double f = M[i][c];
            if (f != 0.0) for (int j = 0; j < 2 * t; ++j) M[i][j] -= f * M[c][j];
        }
    }
    for (int i = 0; i < t; ++i) for (int j = 0; j < t; ++j) Ainv[i * t + j] = (float)M[i][t + j];
    return 0;
}

// Run f(cols) with the accessor of the context's storage (columns from j_off on).
template <class F>
static auto with_cols(jwas_hip_ctx* c, int64_t j_off, F&& f)
{
    if (c->packed) return f(PackedCols{c->Q + j_off * (c->ld >> 2), c->ld, c->qmean + j_off, c->n, c->centered, c->w, (int32_t)c->weighted});
    return f(DenseCols{c->X + j_off * c->ld, c->ld, c->w, (int32_t)c->weighted});
}
#define HAVE_STORAGE(c) ((c)->X != nullptr || (c)->Q != nullptr)
#define IS_F64(c) ((c)->f64 != nullptr)
#define NOT_F64(c, what) NEED(c, !IS_F64(c), JWAS_HIP_EUNSUP, "%s is not available in a Float64 context (double_precision=true)", what)
#define ONLY_F64(c) NEED(c, IS_F64(c), JWAS_HIP_ESTATE, "this entry point needs a Float64 context (jwas_hip_set_precision(ctx, 64))")
static int f64_setup_blocks(jwas_hip_ctx* c, int32_t bs);
static int f64_setup_blocks_explicit(jwas_hip_ctx* c, const int64_t* starts, int64_t nblocks);
static int f64_set_weights(jwas_hip_ctx* c, const float* rinv32, const double* rinv64);
static int f64_init_state(jwas_hip_ctx* c, int32_t method, int32_t nt);


extern "C" {

const char* jwas_hip_last_error(const jwas_hip_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int jwas_hip_create(int device, jwas_hip_ctx** out)
{
    if (!out) return fail(nullptr, JWAS_HIP_EINVAL, "jwas_hip_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, JWAS_HIP_EHIP, "no HIP device available (%s); the HIP path has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, JWAS_HIP_EINVAL, "device %d out of range [0,%d)", device, ndev);
    if (device >= 64) return fail(nullptr, JWAS_HIP_EUNSUP, "device ids above 63 are not supported (per-device kernel attributes are tracked in 64-bit masks)");
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, JWAS_HIP_EHIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(nullptr, JWAS_HIP_EHIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, JWAS_HIP_EUNSUP, "device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    jwas_hip_ctx* c = new jwas_hip_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(nullptr, JWAS_HIP_EHIP, "hipStreamCreate failed"); }
    c->own_stream = true;
    if (hipEventCreate(&c->ev_start) != hipSuccess || hipEventCreate(&c->ev_stop) != hipSuccess) { delete c; return fail(nullptr, JWAS_HIP_EHIP, "hipEventCreate failed"); }
    *out = c;
    return JWAS_HIP_OK;
}

static void free_state(jwas_hip_ctx* c)
{
    (void)hipFree(c->alpha); (void)hipFree(c->beta); (void)hipFree(c->delta);
    (void)hipFree(c->mean_a); (void)hipFree(c->mean_a2); (void)hipFree(c->mean_d);
    (void)hipFree(c->prep_d); (void)hipFree(c->prep_f); c->prep_d = nullptr; c->prep_f = nullptr;
    (void)hipFree(c->mt2_tab); c->mt2_tab = nullptr;
    (void)hipFree(c->tsec); c->tsec = nullptr; c->tsec_cap = 0;
    (void)hipFree(c->xch); c->xch = nullptr;
    (void)hipFree(c->cmp_idx); (void)hipFree(c->cmp_val); c->cmp_idx = nullptr; c->cmp_val = nullptr;
    c->alpha = c->beta = nullptr; c->delta = nullptr; c->mean_a = c->mean_a2 = c->mean_d = nullptr;
    (void)hipFree(c->var_mat); (void)hipFree(c->ginv_mat); c->var_mat = nullptr; c->ginv_mat = nullptr; c->var_mat_resident = false;
}

static void free_blocks(jwas_hip_ctx* c)
{
    (void)hipFree(c->xpx);
    for (auto& b : c->sets) {
        (void)hipFree(b.gram); (void)hipFree(b.cross); (void)hipFree(b.corr); (void)hipFree(b.partials);
        (void)hipFree(b.gcross[0]); (void)hipFree(b.gcross[1]); (void)hipFree(b.gcbuf); (void)hipFree(b.gidx); (void)hipFree(b.gdelta); (void)hipFree(b.gpp);
    }
    c->sets.clear();
    c->xpx = c->gram = c->cross = c->corr = nullptr; c->partials = nullptr;
    c->block_size = 0; c->nblocks = 0;
    c->starts.clear(); (void)hipFree(c->d_starts); c->d_starts = nullptr;
    (void)hipFree(c->ipartials); (void)hipFree(c->ev_all); (void)hipFree(c->ev_offs); (void)hipFree(c->idx_all); (void)hipFree(c->delta_all);
    c->ipartials = nullptr; c->ev_all = nullptr; c->ev_offs = nullptr; c->idx_all = nullptr; c->delta_all = nullptr; c->ind_traits = 0;
}

static void free_storage(jwas_hip_ctx* c)
{
    (void)hipFree(c->X); (void)hipFree(c->r); (void)hipFree(c->Q); (void)hipFree(c->qmean); (void)hipFree(c->w);
    c->X = c->r = nullptr; c->Q = nullptr; c->qmean = nullptr; c->packed = false; c->w = nullptr; c->weighted = false;
    (void)hipFree(c->ev); (void)hipFree(c->dparams); (void)hipFree(c->counters); (void)hipFree(c->fin_out); (void)hipFree(c->stat_out); (void)hipFree(c->sync_cnt);
    c->sync_cnt = nullptr; c->ev = nullptr; c->dparams = nullptr; c->counters = nullptr; c->fin_out = c->stat_out = nullptr;
    if (c->host_buf) (void)hipHostFree(c->host_buf);
    c->host_buf = nullptr;
    (void)hipFree(c->var_vec); (void)hipFree(c->pi_vec); (void)hipFree(c->pi_mat); (void)hipFree(c->lpr_mat);
    (void)hipFree(c->var_mat); (void)hipFree(c->ginv_mat); c->var_mat = c->ginv_mat = nullptr;
    c->var_vec = nullptr; c->pi_vec = c->pi_mat = c->lpr_mat = nullptr;
    (void)hipFree(c->Xout); c->Xout = nullptr; c->n_out = c->ld_out = 0;
}

void jwas_hip_destroy(jwas_hip_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)jwas_hip_comm_destroy(c);
    free_state(c); free_blocks(c); free_storage(c);
    if (c->f64) {
        auto* F = c->f64;
        for (void* q : {(void*)F->X, (void*)F->r, (void*)F->xpx, (void*)F->gram, (void*)F->alpha, (void*)F->beta, F->delta, (void*)F->mean_a,
                        (void*)F->mean_a2, (void*)F->mean_d, (void*)F->partials, (void*)F->ev, (void*)F->dparams, (void*)F->var_vec, (void*)F->w, (void*)F->ev_all}) (void)hipFree(q);
        delete F;
        c->f64 = nullptr;
    }
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    for (hipEvent_t e : c->kev) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int jwas_hip_set_stream(jwas_hip_ctx* c, void* s)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)s;
    c->own_stream = false;
    return JWAS_HIP_OK;
}

int jwas_hip_device_info(jwas_hip_ctx* c, int* n_cu, int64_t* total, int64_t* free_b)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
    size_t f = 0, t = 0;
    HIPCHK(c, hipMemGetInfo(&f, &t));
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (total) *total = (int64_t)t;
    if (free_b) *free_b = (int64_t)f;
    return JWAS_HIP_OK;
}

int64_t jwas_hip_estimate_bytes(int64_t n, int64_t p, int32_t ntraits, int32_t block_size)
{
    // HBM analogue of estimate_marker_memory (tools4genotypes.jl:99-235): X + Grams + x'x + state.
    const int64_t ld = round_up(n, kSliceRows);
    int64_t bytes = 4 * ld * p;                                  // X
    bytes += 2 * 4 * (int64_t)block_size * p;                    // Grams + cross-Grams (p/b blocks of b*b each)
    bytes += 4 * p;                                              // x'x
    bytes += (int64_t)ntraits * p * 4 * 6;                       // alpha, beta, delta, 3 running means
    bytes += (int64_t)kMaxT * ld * 4;                            // residuals
    bytes += (int64_t)block_size * (ld / kSliceRows) * ntraits * 8;   // slice partials
    if (ntraits >= 2 && ntraits <= 3 && block_size == 256)       // Rule T (section_solve): the per-sweep section inverses, (64 t)^2 floats per 64 markers
        bytes += (p / 256) * 4 * (int64_t)(64 * ntraits) * (64 * ntraits) * 4;
    return bytes;
}

int64_t jwas_hip_estimate_bytes_storage(int64_t n, int64_t p, int32_t ntraits, int32_t block_size, int32_t storage)
{
    const int64_t ld = round_up(n, kSliceRows);
    int64_t bytes = jwas_hip_estimate_bytes(n, p, ntraits, block_size);
    if (storage == JWAS_HIP_STORAGE_PACKED2BIT) bytes += (ld >> 2) * p + 4 * p - 4 * ld * p;   // payload + means instead of fp32 X
    return bytes;
}

static int alloc_storage(jwas_hip_ctx* c, int64_t n, int64_t p, bool packed = false)
{
    NEED(c, n > 0 && p > 0, JWAS_HIP_EINVAL, "genotype matrix must be non-empty (n=%lld, p=%lld)", (long long)n, (long long)p);
    NEED(c, p < (1ll << 31), JWAS_HIP_EUNSUP, "p=%lld exceeds the 2^31 marker limit of one context", (long long)p);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->comm || c->loop_slot >= 0) (void)jwas_hip_comm_destroy(c);     // (its buffers are sized by the matrix)
    free_state(c); free_blocks(c); free_storage(c);
    c->method = -1; c->block_size = 0; c->nblocks = 0;
    c->n = n; c->p = p; c->ld = round_up(n, kSliceRows);
    c->nslices = (int)(c->ld / kSliceRows);
    c->upd_nslices = packed ? (int)((c->ld + kWideRows - 1) / kWideRows) : c->nslices;
    const int usl = c->upd_nslices;
    // Update-role geometry.  A workgroup is one row group (spg slices of 256 rows, one wave each, spg <= 8) x one column
    // group.  The step kernel's dynamic LDS (sized for the sampler role) allows one workgroup per CU and the quiet-XCD
    // placement leaves every 8th CU idle, so 224 of the 256 CUs stream; more workgroups than that run as a second round
    // and cost up to 2x (measured: 5.5-6.1 TB/s with <= 224 workgroups, 3.8-4.1 TB/s with 235-245).  Every column group
    // of a row group re-applies the previous block's changes to its copy of the residual slice, so the number of column
    // groups is capped.  Short matrices have few row groups and need many column groups to put enough workgroups on the
    // chip (per sweep at p = 102 400, sparse steady state, 1024-marker blocks, cap 8 -> 16 -> 32: n = 5 000: 2.60 -> 1.92
    // -> 1.80 ms; 10 000: 2.67 -> 1.96 -> 1.87; 20 000: 2.84 -> 2.14 -> 2.22; 30 000: 2.99 -> 2.61).
    // Among spg = 4..8 the geometry with the most streaming waves (slices x column groups) wins, ties go to the one that
    // spreads them over more CUs: at n = 50 000 (196 slices) spg 8 gives 25 x 8 = 200 workgroups, spg 7 gives 28 x 8 =
    // 224 with the same 1568 waves -- 35.1 instead of 36.1 us per 1024-marker launch (48.0 vs 46.4-46.8 iterations/s, twice);
    // n = 280 000: spg 5, 219 instead of 137 workgroups, 18.4 instead of 19.8 ms per 102 400-marker sweep; n = 100 000: neutral.
    {
        const char* e_spg = std::getenv("JWAS_HIP_SPG");                 // (experiments)
        const char* e_cap = std::getenv("JWAS_HIP_MAX_NCG");
        const char* e_bud = std::getenv("JWAS_HIP_WG_BUDGET");           // (experiments) streaming workgroups per launch: 224 of the 256 CUs
        const int budget = e_bud ? std::max(8, std::min(255, std::atoi(e_bud))) : 224;
        int best_spg = kRowGroupSlices, best_ncg = 1, best_nrg = (usl + kRowGroupSlices - 1) / kRowGroupSlices;
        long best_waves = -1, best_wgs = -1;
        for (int spg = kRowGroupSlices; spg >= 4; --spg) {
            if (e_spg && spg != std::max(1, std::min(kRowGroupSlices, std::atoi(e_spg)))) continue;
            const int nrg = (usl + spg - 1) / spg;
            int ncg = budget / nrg; if (ncg < 1) ncg = 1;
            const int cap = e_cap ? std::atoi(e_cap) : (nrg < 8 ? 32 : 16);
            if (ncg > cap) ncg = cap;
            const long waves = (long)usl * ncg, wgs = (long)nrg * ncg;
            // (ties are only broken for tall matrices, where the launch is bound by the update role; shorter ones are
            // bound by the sampler and measured neutral to slightly worse with 6-7 slices per group)
            if (waves > best_waves || (waves == best_waves && wgs > best_wgs && wgs <= budget && usl >= 128)) {
                best_waves = waves; best_wgs = wgs; best_spg = spg; best_ncg = ncg; best_nrg = nrg;
            }
        }
        c->spg = best_spg; c->nrg = best_nrg; c->ncg = best_ncg;
    }
    size_t fb = 0, tb = 0;
    HIPCHK(c, hipMemGetInfo(&fb, &tb));
    // (packed: + 1 KB -- the last 1024-row slice of the LAST column may reach past the column's end; what it reads there meets r = 0)
    const size_t need = packed ? (size_t)(c->ld >> 2) * p + 1024 : (size_t)4 * c->ld * p;
    NEED(c, need < fb, JWAS_HIP_ENOMEM, "genotype matrix needs %.2f GB but only %.2f GB of HBM is free", need / 1e9, fb / 1e9);
    c->packed = packed;
    if (packed) {
        HIPCHK(c, hipMalloc(&c->Q, need));
        HIPCHK(c, hipMalloc(&c->qmean, sizeof(float) * p));
    } else HIPCHK(c, hipMalloc(&c->X, need));
    {   // unit residual weights by default
        std::vector<float> ones((size_t)c->ld, 0.f);
        std::fill(ones.begin(), ones.begin() + n, 1.f);
        HIPCHK(c, hipMalloc(&c->w, sizeof(float) * c->ld));
        HIPCHK(c, hipMemcpy(c->w, ones.data(), sizeof(float) * c->ld, hipMemcpyHostToDevice));
        c->weighted = false;
    }
    HIPCHK(c, hipMalloc(&c->r, sizeof(float) * 2 * kMaxT * c->ld));
    HIPCHK(c, hipMemsetAsync(c->r, 0, sizeof(float) * 2 * kMaxT * c->ld, c->stream));
    HIPCHK(c, hipMalloc(&c->ev, sizeof(Events) * 2));
    HIPCHK(c, hipMemsetAsync(c->ev, 0, sizeof(Events) * 2, c->stream));
    HIPCHK(c, hipMalloc(&c->dparams, sizeof(DevParams)));
    HIPCHK(c, hipMalloc(&c->counters, sizeof(unsigned long long) * kNCounters));
    HIPCHK(c, hipMalloc(&c->fin_out, sizeof(double) * c->nslices * (kMaxT * kMaxT + kMaxT)));
    HIPCHK(c, hipMalloc(&c->sync_cnt, sizeof(int) * 2 * c->nrg));
    HIPCHK(c, hipMemsetAsync(c->sync_cnt, 0, sizeof(int) * 2 * c->nrg, c->stream));
    HIPCHK(c, hipMalloc(&c->stat_out, sizeof(double) * kStatGrid * kNStat));
    HIPCHK(c, hipHostMalloc(&c->host_buf, sizeof(double) * ((size_t)c->nslices * (kMaxT * kMaxT + kMaxT) + kStatGrid * kNStat + 64)));
    return JWAS_HIP_OK;
}

int jwas_hip_load_dense_f32(jwas_hip_ctx* c, const float* Xh, int64_t n, int64_t p, int64_t ld_host)
{
    if (c) NOT_F64(c, "jwas_hip_load_dense_f32 (use jwas_hip_load_dense_f64)");
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, Xh, JWAS_HIP_EINVAL, "X_host is NULL");
    NEED(c, ld_host >= n, JWAS_HIP_EINVAL, "ld_host (%lld) must be >= n (%lld)", (long long)ld_host, (long long)n);
    int rc = alloc_storage(c, n, p);
    if (rc) return rc;
    if (c->ld != n) HIPCHK(c, hipMemsetAsync(c->X, 0, (size_t)4 * c->ld * p, c->stream));
    HIPCHK(c, hipMemcpy2DAsync(c->X, (size_t)4 * c->ld, Xh, (size_t)4 * ld_host, (size_t)4 * n, (size_t)p,
                               hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

int jwas_hip_alloc_dense_f32(jwas_hip_ctx* c, int64_t n, int64_t p)
{
    if (c) NOT_F64(c, "jwas_hip_alloc_dense_f32");
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    return alloc_storage(c, n, p);
}

int jwas_hip_alloc_packed2bit(jwas_hip_ctx* c, int64_t n, int64_t p, int32_t centered)
{
    if (c) NOT_F64(c, "2-bit packed storage");
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    int rc = alloc_storage(c, n, p, true);
    if (rc) return rc;
    c->centered = centered ? 1 : 0;
    return JWAS_HIP_OK;
}

int jwas_hip_load_packed2bit(jwas_hip_ctx* c, const uint8_t* payload, int64_t n, int64_t p, int64_t stride_bytes,
                             const float* means, int32_t centered)
{
    if (c) NOT_F64(c, "2-bit packed storage");
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, payload && means, JWAS_HIP_EINVAL, "payload / means is NULL");
    NEED(c, stride_bytes >= (n + 3) / 4, JWAS_HIP_EINVAL, "stride_bytes (%lld) must be >= cld(n,4) = %lld", (long long)stride_bytes, (long long)((n + 3) / 4));
    int rc = alloc_storage(c, n, p, true);
    if (rc) return rc;
    c->centered = centered ? 1 : 0;
    const size_t sb = (size_t)(c->ld >> 2), src = (size_t)((n + 3) / 4);
    HIPCHK(c, hipMemsetAsync(c->Q, 0, sb * p, c->stream));
    HIPCHK(c, hipMemcpy2DAsync(c->Q, sb, payload, (size_t)stride_bytes, src, (size_t)p, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->qmean, means, sizeof(float) * p, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

// load_streaming_backend (streaming_genotypes.jl:884-971): manifest = tab-separated key/value lines
static bool read_manifest(const std::string& path, std::vector<std::pair<std::string, std::string>>* kv)
{
    FILE* f = std::fopen(path.c_str(), "r");
    if (!f) return false;
    char line[8192];
    while (std::fgets(line, sizeof line, f)) {
        std::string s(line);
        while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
        const size_t tab = s.find('\t');
        if (tab != std::string::npos) kv->emplace_back(s.substr(0, tab), s.substr(tab + 1));
    }
    std::fclose(f);
    return true;
}

int jwas_hip_load_jgb2(jwas_hip_ctx* c, const char* path)
{
    if (c) NOT_F64(c, "2-bit packed storage");
    NEED(c, c && path, JWAS_HIP_EINVAL, "NULL argument");
    std::string prefix(path);                                   // _resolve_streaming_prefix (:97-105)
    for (const char* ext : {".meta", ".jgb2"}) {
        const size_t L = std::strlen(ext);
        if (prefix.size() >= L && prefix.compare(prefix.size() - L, L, ext) == 0) { prefix.resize(prefix.size() - L); break; }
    }
    std::vector<std::pair<std::string, std::string>> kv;
    NEED(c, read_manifest(prefix + ".meta", &kv), JWAS_HIP_EINVAL, "Streaming manifest is not found: %s.meta", prefix.c_str());
    auto get = [&](const char* k) -> std::string { for (auto& e : kv) if (e.first == k) return e.second; return std::string(); };
    const std::string sn = get("nObs"), sp = get("nMarkers"), ss = get("stride_bytes"), sc = get("centered");
    NEED(c, !sn.empty() && !sp.empty() && !ss.empty() && !sc.empty(), JWAS_HIP_EINVAL, "Streaming manifest %s.meta lacks nObs / nMarkers / stride_bytes / centered", prefix.c_str());
    const int64_t n = std::atoll(sn.c_str()), p = std::atoll(sp.c_str()), stride = std::atoll(ss.c_str());
    const int centered = std::atoi(sc.c_str()) == 1;
    NEED(c, n > 0 && p > 0 && stride >= (n + 3) / 4, JWAS_HIP_EINVAL, "Streaming manifest %s.meta is inconsistent", prefix.c_str());
    // recorded paths may be stale if the files were moved: fall back to <prefix>.jgb2 / <prefix>.mean.f32
    auto open_first = [&](const std::string& a, const std::string& b) -> FILE* {
        FILE* f = a.empty() ? nullptr : std::fopen(a.c_str(), "rb");
        return f ? f : std::fopen(b.c_str(), "rb");
    };
    FILE* fd = open_first(get("data_path"), prefix + ".jgb2");
    NEED(c, fd, JWAS_HIP_EINVAL, "Packed genotype file is not found for %s", prefix.c_str());
    std::fseek(fd, 0, SEEK_END);
    const int64_t fsz = (int64_t)std::ftell(fd);
    std::fseek(fd, 0, SEEK_SET);
    if (fsz != p * stride) { std::fclose(fd); return fail(c, JWAS_HIP_EINVAL, "Packed genotype file size does not match metadata for %s", prefix.c_str()); }
    FILE* fm = open_first(get("mean_path"), prefix + ".mean.f32");
    if (!fm) { std::fclose(fd); return fail(c, JWAS_HIP_EINVAL, "marker mean sidecar is not found for %s", prefix.c_str()); }
    std::vector<float> means((size_t)p);
    const size_t got = std::fread(means.data(), sizeof(float), (size_t)p, fm);
    std::fclose(fm);
    if ((int64_t)got != p) { std::fclose(fd); return fail(c, JWAS_HIP_EINVAL, "marker mean sidecar of %s is truncated", prefix.c_str()); }
    int rc = alloc_storage(c, n, p, true);
    if (rc) { std::fclose(fd); return rc; }
    c->centered = centered;
    const size_t sb = (size_t)(c->ld >> 2);
    hipError_t e = hipMemsetAsync(c->Q, 0, sb * p, c->stream);
    // stream the file through a pinned staging buffer, <= 64 MB of markers at a time
    const int64_t chunk = std::max<int64_t>(1, (64ll << 20) / stride);
    void* stage = nullptr;
    if (e == hipSuccess) e = hipHostMalloc(&stage, (size_t)(chunk * stride));
    for (int64_t j0 = 0; e == hipSuccess && j0 < p; j0 += chunk) {
        const int64_t m = std::min(chunk, p - j0);
        if ((int64_t)std::fread(stage, (size_t)stride, (size_t)m, fd) != m) { e = hipErrorUnknown; break; }
        e = hipMemcpy2DAsync(c->Q + j0 * sb, sb, stage, (size_t)stride, (size_t)((n + 3) / 4), (size_t)m, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    std::fclose(fd);
    if (stage) (void)hipHostFree(stage);
    if (e == hipSuccess) e = hipMemcpyAsync(c->qmean, means.data(), sizeof(float) * p, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, JWAS_HIP_EHIP, "jwas_hip_load_jgb2: reading / uploading %s.jgb2 failed (%s)", prefix.c_str(), hipGetErrorString(e));
    return JWAS_HIP_OK;
}

int jwas_hip_set_weights(jwas_hip_ctx* c, const float* rinv)
{
    if (c && IS_F64(c)) return f64_set_weights(c, rinv, nullptr);
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, HAVE_STORAGE(c), JWAS_HIP_ESTATE, "no genotype matrix loaded");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<float> wv((size_t)c->ld, 0.f);
    bool unit = true;
    for (int64_t i = 0; i < c->n; ++i) {
        const float v = rinv ? rinv[i] : 1.f;
        NEED(c, std::isfinite(v) && v > 0.f, JWAS_HIP_EINVAL, "residual weights must be positive and finite (row %lld: %g)", (long long)i, (double)v);
        wv[(size_t)i] = v;
        unit = unit && (v == 1.f);                              // is_unit_weights (tools4genotypes.jl:43-51)
    }
    HIPCHK(c, hipMemcpy(c->w, wv.data(), sizeof(float) * c->ld, hipMemcpyHostToDevice));
    c->weighted = !unit;
    free_blocks(c);                                             // x'R^-1 x and the Grams depend on the weights
    return JWAS_HIP_OK;
}

int jwas_hip_set_weights_f64(jwas_hip_ctx* c, const double* rinv)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, IS_F64(c), JWAS_HIP_ESTATE, "jwas_hip_set_weights_f64 needs a Float64 context (jwas_hip_set_precision(ctx, 64)); a Float32 context takes jwas_hip_set_weights");
    return f64_set_weights(c, nullptr, rinv);
}

int jwas_hip_storage_info(jwas_hip_ctx* c, int32_t* kind, int64_t* n, int64_t* p, int64_t* bytes)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, HAVE_STORAGE(c), JWAS_HIP_ESTATE, "no genotype matrix loaded");
    if (kind) *kind = c->packed ? JWAS_HIP_STORAGE_PACKED2BIT : JWAS_HIP_STORAGE_DENSE_F32;
    if (n) *n = c->n;
    if (p) *p = c->p;
    if (bytes) *bytes = c->packed ? (c->ld >> 2) * c->p : 4 * c->ld * c->p;
    return JWAS_HIP_OK;
}

int jwas_hip_dense_layout(jwas_hip_ctx* c, int64_t* n, int64_t* p, int64_t* ld, void** Xd)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, HAVE_STORAGE(c), JWAS_HIP_ESTATE, "no genotype matrix loaded");
    NEED(c, !c->packed, JWAS_HIP_EUNSUP, "the context holds 2-bit packed genotypes (see jwas_hip_storage_info)");
    if (n) *n = c->n;
    if (p) *p = c->p;
    if (ld) *ld = c->ld;
    if (Xd) *Xd = c->X;
    return JWAS_HIP_OK;
}

int jwas_hip_get_columns(jwas_hip_ctx* c, int64_t j0, int64_t count, float* out)
{
    NEED(c, c && out, JWAS_HIP_EINVAL, "NULL argument");
    NEED(c, HAVE_STORAGE(c), JWAS_HIP_ESTATE, "no genotype matrix loaded");
    NEED(c, j0 >= 0 && count >= 0 && j0 + count <= c->p, JWAS_HIP_EINVAL, "column range [%lld,%lld) outside [0,%lld)",
         (long long)j0, (long long)(j0 + count), (long long)c->p);
    HIPCHK(c, hipSetDevice(c->device));
    if (count == 0) return JWAS_HIP_OK;
    if (!c->packed) {
        HIPCHK(c, hipMemcpy2DAsync(out, (size_t)4 * c->n, c->X + j0 * c->ld, (size_t)4 * c->ld, (size_t)4 * c->n, (size_t)count,
                                   hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return JWAS_HIP_OK;
    }
    // packed storage: decode on the device (decode_marker!, streaming_genotypes.jl:978-1002), <= 4096 columns at a time
    const int64_t chunk = 4096;
    float* tmp = nullptr;
    HIPCHK(c, hipMalloc(&tmp, sizeof(float) * (size_t)c->n * (size_t)std::min(chunk, count)));
    hipError_t e = hipSuccess;
    for (int64_t k0 = 0; e == hipSuccess && k0 < count; k0 += chunk) {
        const int64_t m = std::min(chunk, count - k0);
        hipLaunchKernelGGL((k_get_columns<PackedCols>), dim3((unsigned)m), dim3(256), 0, c->stream,
                           PackedCols{c->Q, c->ld, c->qmean, c->n, c->centered, c->w, (int32_t)c->weighted}, j0 + k0, c->n, tmp);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(out + k0 * c->n, tmp, sizeof(float) * (size_t)c->n * m, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(c, JWAS_HIP_EHIP, "jwas_hip_get_columns: %s", hipGetErrorString(e));
    return JWAS_HIP_OK;
}

int jwas_hip_set_columns(jwas_hip_ctx* c, int64_t j0, int64_t count, const float* in, int64_t ld_host)
{
    NEED(c, c && in, JWAS_HIP_EINVAL, "NULL argument");
    NEED(c, c->X != nullptr, JWAS_HIP_ESTATE, "allocate a dense matrix first (jwas_hip_alloc_dense_f32)");
    NEED(c, j0 >= 0 && count >= 0 && j0 + count <= c->p, JWAS_HIP_EINVAL, "column range [%lld,%lld) outside [0,%lld)",
         (long long)j0, (long long)(j0 + count), (long long)c->p);
    NEED(c, ld_host >= c->n, JWAS_HIP_EINVAL, "ld_host (%lld) must be >= n (%lld)", (long long)ld_host, (long long)c->n);
    HIPCHK(c, hipSetDevice(c->device));
    if (count == 0) return JWAS_HIP_OK;
    if (c->ld != c->n)          // pad rows of these columns
        HIPCHK(c, hipMemset2DAsync(c->X + j0 * c->ld + c->n, (size_t)4 * c->ld, 0, (size_t)4 * (c->ld - c->n), (size_t)count, c->stream));
    HIPCHK(c, hipMemcpy2DAsync(c->X + j0 * c->ld, (size_t)4 * c->ld, in, (size_t)4 * ld_host, (size_t)4 * c->n, (size_t)count,
                               hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_blocks(c);                                             // x'x and the Grams depend on the matrix
    return JWAS_HIP_OK;
}

int jwas_hip_synth_genotypes(jwas_hip_ctx* c, uint64_t seed, int32_t kind, int32_t center, int64_t marker_offset)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, HAVE_STORAGE(c), JWAS_HIP_ESTATE, "allocate the matrix first (jwas_hip_alloc_dense_f32 / jwas_hip_alloc_packed2bit)");
    NEED(c, kind == 0 || kind == 1, JWAS_HIP_EINVAL, "kind must be 0 (0/1/2 genotypes) or 1 (uniform)");
    HIPCHK(c, hipSetDevice(c->device));
    NEED(c, marker_offset >= 0 && marker_offset + c->p < (1ll << 32), JWAS_HIP_EINVAL, "marker_offset out of range");
    if (c->packed) {
        NEED(c, kind == 0, JWAS_HIP_EUNSUP, "2-bit packed storage holds 0/1/2 genotypes only (kind 0)");
        c->centered = center ? 1 : 0;
        hipLaunchKernelGGL(k_synth_packed, dim3((unsigned)c->p), dim3(256), 0, c->stream, c->Q, c->qmean, c->n, c->ld,
                           (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)marker_offset);
    } else
    hipLaunchKernelGGL(k_synth, dim3((unsigned)c->p), dim3(256), 0, c->stream, c->X, c->n, c->ld,
                       (uint32_t)seed, (uint32_t)(seed >> 32), (int)kind, (int)center, (uint32_t)marker_offset, (int64_t)0);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

int jwas_hip_synth_single_step(jwas_hip_ctx* c, uint64_t seed, int64_t n_genotyped, int32_t center, int64_t marker_offset)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, c->X != nullptr, JWAS_HIP_ESTATE, "allocate a dense matrix first (jwas_hip_alloc_dense_f32): imputed genotypes are real-valued");
    NEED(c, n_genotyped >= 1 && n_genotyped <= c->n, JWAS_HIP_EINVAL, "n_genotyped must lie in [1, n] (got %lld)", (long long)n_genotyped);
    NEED(c, marker_offset >= 0 && marker_offset + c->p < (1ll << 32), JWAS_HIP_EINVAL, "marker_offset out of range");
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_synth, dim3((unsigned)c->p), dim3(256), 0, c->stream, c->X, c->n, c->ld,
                       (uint32_t)seed, (uint32_t)(seed >> 32), 2, (int)center, (uint32_t)marker_offset, n_genotyped);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

// ---- precompute ---------------------------------------------------------------------------------
// first marker / size of block k of the current partition (uniform blocks, or the explicit starts of
// jwas_hip_setup_blocks_explicit)
static inline int64_t blk_j0(const jwas_hip_ctx* c, int64_t k) { return c->starts.empty() ? k * c->block_size : c->starts[(size_t)k]; }
static inline int blk_b(const jwas_hip_ctx* c, int64_t k)
{
    if (!c->starts.empty()) return (int)(c->starts[(size_t)k + 1] - c->starts[(size_t)k]);
    const int64_t j0 = k * c->block_size;
    return (int)((j0 + c->block_size <= c->p) ? c->block_size : c->p - j0);
}

static void select_set(jwas_hip_ctx* c, size_t i)
{
    const auto& b = c->sets[i];
    c->set_index = (int)i;
    c->block_size = b.bs; c->nblocks = b.nblocks;
    c->gram = b.gram; c->cross = b.cross; c->corr = b.corr; c->partials = b.partials;
    // independent-mode buffers are sized by the block configuration: rebuild them on next use
    (void)hipFree(c->ipartials); (void)hipFree(c->ev_all); (void)hipFree(c->ev_offs); (void)hipFree(c->idx_all); (void)hipFree(c->delta_all);
    c->ipartials = nullptr; c->ev_all = nullptr; c->ev_offs = nullptr; c->idx_all = nullptr; c->delta_all = nullptr; c->ind_traits = 0;
}

// Grams and cross-Grams of one block size (x'x must exist); c->starts non-empty: of the explicit partition (bs >= every block).
static int build_block_set(jwas_hip_ctx* c, int32_t bs, int32_t gram_mode)
{
    jwas_hip_ctx::BlockSet B{};
    B.bs = bs;
    B.nblocks = c->starts.empty() ? (c->p + bs - 1) / bs : (int64_t)c->starts.size() - 1;
    HIPCHK(c, hipMalloc(&B.gram, sizeof(float) * (size_t)B.nblocks * bs * bs));
    HIPCHK(c, hipMalloc(&B.cross, sizeof(float) * (size_t)B.nblocks * bs * bs));
    HIPCHK(c, hipMalloc(&B.corr, sizeof(float) * 2 * kMaxT * (size_t)bs));
    HIPCHK(c, hipMalloc(&B.partials, sizeof(double) * 2 * (size_t)bs * c->nrg * kMaxT));   // ping-pong
#ifdef JWAS_HIP_DEV_KNOBS
    if (std::getenv("JWAS_HIP_DEBUG_POISON")) {          // development builds: NaN-fill what the setup kernels do not write
        HIPCHK(c, hipMemset(B.gram, 0xFF, sizeof(float) * (size_t)B.nblocks * bs * bs));
        HIPCHK(c, hipMemset(B.cross, 0xFF, sizeof(float) * (size_t)B.nblocks * bs * bs));
        HIPCHK(c, hipMemset(B.partials, 0xFF, sizeof(double) * 2 * (size_t)bs * c->nrg * kMaxT));
    }
#endif
    c->sets.push_back(B);
    // Gram launches are chunked over blocks so grid.y stays below 65536
    const int64_t ychunk = 32768;
    for (int64_t y0 = 0; y0 < B.nblocks; y0 += ychunk) {
        const int64_t ny = (B.nblocks - y0 < ychunk) ? B.nblocks - y0 : ychunk;
        float* Gc = B.gram + y0 * (int64_t)bs * bs;
        const int64_t pc = c->p - y0 * bs;
        float* Cc = B.cross + y0 * (int64_t)bs * bs;
        const int64_t nyc = (y0 + ny < B.nblocks) ? ny : ny - 1;      // cross blocks y0+1 .. (last block has none after it)
        const int64_t* ds = c->d_starts;                  // explicit starts: absolute marker indices, one launch (<= 32768 blocks)
        with_cols(c, ds ? 0 : y0 * bs, [&](auto Xc) {
            using CX = decltype(Xc);
            if (gram_mode == JWAS_HIP_GRAM_F64) {
                hipLaunchKernelGGL((k_gram_f64<CX>), dim3(bs, (unsigned)ny), dim3(256), 0, c->stream, Xc, pc, (int)bs, Gc, ds);
                if (nyc > 0) hipLaunchKernelGGL((k_cross_f64<CX>), dim3(bs, (unsigned)nyc), dim3(256), 0, c->stream, Xc, pc, (int)bs, Cc, ds);
            } else {
                const int nt = bs / 64;
                hipLaunchKernelGGL((k_gram_mfma<CX>), dim3(nt * (nt + 1) / 2, (unsigned)ny), dim3(256), 0, c->stream, Xc, pc, (int)bs, Gc, 0, ds);
                if (nyc > 0) hipLaunchKernelGGL((k_gram_mfma<CX>), dim3(nt * nt, (unsigned)nyc), dim3(256), 0, c->stream, Xc, pc, (int)bs, Cc, 1, ds);
            }
            return 0;
        });
        HIPCHK(c, hipGetLastError());
    }
    if (c->row_mode) {      // exact row shards: X_b'X_b and the cross-Grams are sums over the ranks' individuals (fp32, once)
        int rc = row_allreduce(c, B.gram, (size_t)B.nblocks * bs * bs, false); if (rc) return rc;
        rc = row_allreduce(c, B.cross, (size_t)B.nblocks * bs * bs, false); if (rc) return rc;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

static int check_block_args(jwas_hip_ctx* c, int32_t bs, int32_t gram_mode)
{
    NEED(c, HAVE_STORAGE(c), JWAS_HIP_ESTATE, "no genotype matrix loaded");
    NEED(c, bs == 64 || bs == 128 || bs == 256 || bs == 512 || bs == 1024, JWAS_HIP_EINVAL, "block_size must be 64, 128, 256, 512 or 1024 (got %d)", bs);
    NEED(c, gram_mode == JWAS_HIP_GRAM_F64 || gram_mode == JWAS_HIP_GRAM_MFMA, JWAS_HIP_EINVAL, "unknown gram_mode %d", gram_mode);
    return JWAS_HIP_OK;
}

int jwas_hip_setup_blocks(jwas_hip_ctx* c, int32_t bs, int32_t gram_mode)
{
    if (c && IS_F64(c)) return f64_setup_blocks(c, bs);
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    int rc = check_block_args(c, bs, gram_mode);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_blocks(c);
    HIPCHK(c, hipMalloc(&c->xpx, sizeof(float) * c->p));
    with_cols(c, 0, [&](auto cx) {
        hipLaunchKernelGGL((k_xpx<decltype(cx)>), dim3((unsigned)c->p), dim3(256), 0, c->stream, cx, c->xpx);
        return 0;
    });
    HIPCHK(c, hipGetLastError());
    if (c->row_mode) { rc = row_allreduce(c, c->xpx, (size_t)c->p, false); if (rc) return rc; }     // x'x over all individuals
    rc = build_block_set(c, bs, gram_mode);
    if (rc) return rc;
    select_set(c, 0);
    return JWAS_HIP_OK;
}

int jwas_hip_setup_blocks_explicit(jwas_hip_ctx* c, const int64_t* starts, int64_t nblocks, int32_t gram_mode)
{
    if (c && IS_F64(c)) return f64_setup_blocks_explicit(c, starts, nblocks);
    NEED(c, c && starts, JWAS_HIP_EINVAL, "NULL argument");
    NEED(c, HAVE_STORAGE(c), JWAS_HIP_ESTATE, "no genotype matrix loaded");
    NEED(c, gram_mode == JWAS_HIP_GRAM_F64 || gram_mode == JWAS_HIP_GRAM_MFMA, JWAS_HIP_EINVAL, "unknown gram_mode %d", gram_mode);
    NEED(c, nblocks >= 1 && nblocks <= 32768, JWAS_HIP_EUNSUP, "explicit block partitions hold 1..32768 blocks (got %lld)", (long long)nblocks);
    NEED(c, starts[0] == 0, JWAS_HIP_EINVAL, "block starts must begin with marker 0");
    int64_t bmax = 0;
    for (int64_t k = 0; k < nblocks; ++k) {
        const int64_t e = (k + 1 < nblocks) ? starts[k + 1] : c->p;
        NEED(c, e > starts[k] && e <= c->p, JWAS_HIP_EINVAL, "block starts must be sorted, unique and below the number of markers");
        bmax = std::max(bmax, e - starts[k]);
    }
    NEED(c, bmax <= 1024, JWAS_HIP_EUNSUP, "a block holds at most 1024 markers (got %lld)", (long long)bmax);
    int32_t bs = 64;
    while (bs < bmax) bs *= 2;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_blocks(c);
    c->starts.assign(starts, starts + nblocks);
    c->starts.push_back(c->p);
    HIPCHK(c, hipMalloc(&c->d_starts, sizeof(int64_t) * c->starts.size()));
    HIPCHK(c, hipMemcpy(c->d_starts, c->starts.data(), sizeof(int64_t) * c->starts.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc(&c->xpx, sizeof(float) * c->p));
    with_cols(c, 0, [&](auto cx) {
        hipLaunchKernelGGL((k_xpx<decltype(cx)>), dim3((unsigned)c->p), dim3(256), 0, c->stream, cx, c->xpx);
        return 0;
    });
    HIPCHK(c, hipGetLastError());
    if (c->row_mode) { int rr = row_allreduce(c, c->xpx, (size_t)c->p, false); if (rr) return rr; }
    int rc = build_block_set(c, bs, gram_mode);
    if (rc) return rc;
    select_set(c, 0);
    return JWAS_HIP_OK;
}

int jwas_hip_add_block_size(jwas_hip_ctx* c, int32_t bs, int32_t gram_mode)
{
    if (c) NOT_F64(c, "a second block size");
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    int rc = check_block_args(c, bs, gram_mode);
    if (rc) return rc;
    NEED(c, !c->sets.empty(), JWAS_HIP_ESTATE, "jwas_hip_setup_blocks has not been called");
    NEED(c, c->starts.empty(), JWAS_HIP_EUNSUP, "a second block size cannot be added to an explicit block partition");
    for (auto& b : c->sets) NEED(c, b.bs != bs, JWAS_HIP_EINVAL, "block size %d is already resident", bs);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return build_block_set(c, bs, gram_mode);
}

int jwas_hip_select_block_size(jwas_hip_ctx* c, int32_t bs)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < c->sets.size(); ++i)
        if (c->sets[i].bs == bs) { if (c->block_size != bs) select_set(c, i); return JWAS_HIP_OK; }
    return fail(c, JWAS_HIP_EINVAL, "block size %d is not resident (jwas_hip_setup_blocks / jwas_hip_add_block_size)", bs);
}

static void free_groups(jwas_hip_ctx::BlockSet& B)
{
    (void)hipFree(B.gcross[0]); (void)hipFree(B.gcross[1]); (void)hipFree(B.gcbuf); (void)hipFree(B.gidx); (void)hipFree(B.gdelta); (void)hipFree(B.gpp);
    B.gcross[0] = B.gcross[1] = B.gcbuf = B.gdelta = nullptr; B.gidx = nullptr; B.gpp = nullptr; B.gm = 0;
}

// Grouped launches for the SELECTED block size: the cross-Grams of consecutive pairs (and, m = 4, fours) of blocks, the
// correction buffers and the merged change lists (sweep.hpp, k_group_step).  m = 0 frees them.
int jwas_hip_setup_groups(jwas_hip_ctx* c, int32_t m, int32_t gram_mode)
{
    if (c) NOT_F64(c, "grouped launches");
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, !c->sets.empty(), JWAS_HIP_ESTATE, "jwas_hip_setup_blocks has not been called");
    NEED(c, m == 0 || m == 2 || m == 4, JWAS_HIP_EINVAL, "blocks per launch must be 0 (off), 2 or 4 (got %d)", m);
    NEED(c, gram_mode == JWAS_HIP_GRAM_F64 || gram_mode == JWAS_HIP_GRAM_MFMA, JWAS_HIP_EINVAL, "unknown gram_mode %d", gram_mode);
    NEED(c, c->starts.empty(), JWAS_HIP_EUNSUP, "grouped launches need uniform blocks (not an explicit block partition)");
    NEED(c, !c->row_mode, JWAS_HIP_EUNSUP, "grouped launches are not available on row shards");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    auto& B = c->sets[(size_t)c->set_index];
    free_groups(B);
    if (m == 0) return JWAS_HIP_OK;
    const int bs = B.bs;
    NEED(c, m * bs <= 4096, JWAS_HIP_EUNSUP, "a launch streams at most 4096 markers (got %d blocks of %d)", m, bs);
    for (int lvl = 0; lvl < (m == 4 ? 2 : 1); ++lvl) {
        const int gb = (2 << lvl) * bs;                                  // markers per pair / per four
        const int64_t ngr = (c->p + gb - 1) / gb;
        // (m = 4 reads the pair -> pair cross-Grams INSIDE a four only: the odd pairs; the even ones are part of level 1)
        const bool odd_only = (m == 4 && lvl == 0);
        const int64_t nlaunch = odd_only ? ngr / 2 : ngr - 1;            // cross blocks 1, 3, 5, ... < ngr  /  1 .. ngr - 1
        // (the odd pairs are stored compactly -- pair 2q + 1 at q: half the bytes of the pair level at 4 blocks per launch)
        HIPCHK(c, hipMalloc(&B.gcross[lvl], sizeof(float) * (size_t)(odd_only ? std::max<int64_t>(nlaunch, 1) : ngr) * gb * gb));
        if (nlaunch > 0) {
            NEED(c, nlaunch <= 65535, JWAS_HIP_EUNSUP, "grouped launches: too many groups (%lld)", (long long)ngr);
            with_cols(c, 0, [&](auto Xc) {
                using CX = decltype(Xc);
                if (gram_mode == JWAS_HIP_GRAM_F64)
                    hipLaunchKernelGGL((k_cross_f64<CX>), dim3(gb, (unsigned)nlaunch), dim3(256), 0, c->stream, Xc, c->p, gb, B.gcross[lvl], (const int64_t*)nullptr, odd_only ? 1 : 0);
                else {
                    const int nt = gb / 128;
                    hipLaunchKernelGGL((k_cross_mfma128<CX>), dim3(nt * nt, (unsigned)nlaunch), dim3(256), 0, c->stream, Xc, c->p, gb, B.gcross[lvl], odd_only ? 2 : 1);
                }
                return 0;
            });
            HIPCHK(c, hipGetLastError());
        }
    }
    const size_t ncb = (size_t)2 * m * bs + 5 * (size_t)bs;
    HIPCHK(c, hipMalloc(&B.gcbuf, sizeof(float) * ncb));
    HIPCHK(c, hipMemsetAsync(B.gcbuf, 0, sizeof(float) * ncb, c->stream));
    HIPCHK(c, hipMalloc(&B.gidx, sizeof(int32_t) * 2 * (size_t)m * bs));
    HIPCHK(c, hipMalloc(&B.gdelta, sizeof(float) * 2 * (size_t)m * bs));
    HIPCHK(c, hipMemsetAsync(B.gidx, 0, sizeof(int32_t) * 2 * (size_t)m * bs, c->stream));
    HIPCHK(c, hipMemsetAsync(B.gdelta, 0, sizeof(float) * 2 * (size_t)m * bs, c->stream));
    {   // hand-over words of the ping-pong samplers: cW [2][bs] | counts [8] | cP part [2 bs] | cP [2 bs] | cG relay [3][m bs]
        const size_t nw = 6 * (size_t)bs + 8 + 3 * (size_t)m * bs;
        HIPCHK(c, hipMalloc(&B.gpp, sizeof(unsigned long long) * nw));
        HIPCHK(c, hipMemsetAsync(B.gpp, 0, sizeof(unsigned long long) * nw, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    B.gm = m;
    return JWAS_HIP_OK;
}

int jwas_hip_num_blocks(jwas_hip_ctx* c, int64_t* nb, int32_t* bs)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, c->block_size, JWAS_HIP_ESTATE, "jwas_hip_setup_blocks has not been called");
    if (nb) *nb = c->nblocks;
    if (bs) *bs = c->block_size;
    return JWAS_HIP_OK;
}

int jwas_hip_get_xpx(jwas_hip_ctx* c, float* out)
{
    if (c) NOT_F64(c, "jwas_hip_get_xpx (use jwas_hip_get_xpx_f64)");
    NEED(c, c && out, JWAS_HIP_EINVAL, "NULL argument");
    NEED(c, c->xpx, JWAS_HIP_ESTATE, "jwas_hip_setup_blocks has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->xpx, sizeof(float) * c->p, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

int jwas_hip_set_xpx(jwas_hip_ctx* c, const float* in)
{
    NEED(c, c && in, JWAS_HIP_EINVAL, "NULL argument");
    NEED(c, c->xpx, JWAS_HIP_ESTATE, "jwas_hip_setup_blocks has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->xpx, in, sizeof(float) * c->p, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

static int block_dims(jwas_hip_ctx* c, int64_t blk, int* b)
{
    NEED(c, c->gram, JWAS_HIP_ESTATE, "jwas_hip_setup_blocks has not been called");
    NEED(c, blk >= 0 && blk < c->nblocks, JWAS_HIP_EINVAL, "block %lld outside [0,%lld)", (long long)blk, (long long)c->nblocks);
    *b = blk_b(c, blk);
    return JWAS_HIP_OK;
}

int jwas_hip_get_gram(jwas_hip_ctx* c, int64_t blk, float* out)
{
    NEED(c, c && out, JWAS_HIP_EINVAL, "NULL argument");
    int b = 0, rc = block_dims(c, blk, &b);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->gram + blk * (int64_t)c->block_size * c->block_size, sizeof(float) * b * b, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

int jwas_hip_update_geometry(jwas_hip_ctx* c, int32_t* spg, int32_t* nrg, int32_t* ncg)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, HAVE_STORAGE(c), JWAS_HIP_ESTATE, "no genotype matrix loaded");
    if (spg) *spg = c->spg;
    if (nrg) *nrg = c->nrg;
    if (ncg) *ncg = c->ncg;
    return JWAS_HIP_OK;
}

int jwas_hip_set_cross_gram(jwas_hip_ctx* c, int64_t blk, const float* in)
{
    NEED(c, c && in, JWAS_HIP_EINVAL, "NULL argument");
    int b = 0, rc = block_dims(c, blk, &b);
    if (rc) return rc;
    NEED(c, blk >= 1, JWAS_HIP_EINVAL, "block 0 has no predecessor");
    const int bp = blk_b(c, blk - 1);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->cross + blk * (int64_t)c->block_size * c->block_size, in, sizeof(float) * (size_t)bp * b, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

int jwas_hip_set_gram(jwas_hip_ctx* c, int64_t blk, const float* in)
{
    NEED(c, c && in, JWAS_HIP_EINVAL, "NULL argument");
    int b = 0, rc = block_dims(c, blk, &b);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->gram + blk * (int64_t)c->block_size * c->block_size, in, sizeof(float) * b * b, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

// ---- chain state ----------------------------------------------------------------------------------
int jwas_hip_init_state(jwas_hip_ctx* c, int32_t method, int32_t nt)
{
    if (c && IS_F64(c)) return f64_init_state(c, method, nt);
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, HAVE_STORAGE(c), JWAS_HIP_ESTATE, "no genotype matrix loaded");
    NEED(c, method >= JWAS_HIP_BAYESC && method <= JWAS_HIP_MEGABAYESB, JWAS_HIP_EINVAL, "unknown method %d", method);
    if (method >= JWAS_HIP_MTBAYESC1) NEED(c, nt >= 2 && nt <= kMaxT, JWAS_HIP_EUNSUP, "multi-trait samplers support 2..%d traits (got %d)", kMaxT, nt);
    else NEED(c, nt == 1, JWAS_HIP_EINVAL, "single-trait method requires ntraits == 1 (got %d)", nt);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_state(c);
    c->method = method; c->ntraits = nt;
    const size_t fb = sizeof(float) * (size_t)nt * c->p;
    HIPCHK(c, hipMalloc(&c->alpha, fb));
    HIPCHK(c, hipMalloc(&c->beta, fb));
    HIPCHK(c, hipMalloc(&c->delta, fb));          // int32 and float are both 4 bytes
    HIPCHK(c, hipMalloc(&c->mean_a, fb));
    HIPCHK(c, hipMalloc(&c->mean_a2, fb));
    HIPCHK(c, hipMalloc(&c->mean_d, fb));
    HIPCHK(c, hipMalloc(&c->prep_d, sizeof(double) * kPrepD * (size_t)c->p));
    HIPCHK(c, hipMalloc(&c->prep_f, sizeof(float) * kPrepF * (size_t)c->p));
    if (method == JWAS_HIP_MTBAYESC2 && nt <= 3)
        HIPCHK(c, hipMalloc(&c->mt2_tab, sizeof(double) * (size_t)((1 << nt) * (nt * (nt + 1) / 2 + 1)) * (size_t)c->p));
    HIPCHK(c, hipMemsetAsync(c->alpha, 0, fb, c->stream));
    HIPCHK(c, hipMemsetAsync(c->beta, 0, fb, c->stream));
    HIPCHK(c, hipMemsetAsync(c->delta, 0, fb, c->stream));
    HIPCHK(c, hipMemsetAsync(c->mean_a, 0, fb, c->stream));
    HIPCHK(c, hipMemsetAsync(c->mean_a2, 0, fb, c->stream));
    HIPCHK(c, hipMemsetAsync(c->mean_d, 0, fb, c->stream));
    HIPCHK(c, hipMemsetAsync(c->r, 0, sizeof(float) * 2 * kMaxT * c->ld, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

#define NEED_TRAIT(c, trait)                                                                       \
    NEED(c, c->method >= 0, JWAS_HIP_ESTATE, "jwas_hip_init_state has not been called");           \
    NEED(c, trait >= 0 && trait < c->ntraits, JWAS_HIP_EINVAL, "trait %d outside [0,%d)", trait, c->ntraits)

int jwas_hip_set_state(jwas_hip_ctx* c, int32_t trait, const float* a, const float* b, const void* d)
{
    if (c) NOT_F64(c, "jwas_hip_set_state (use jwas_hip_set_state_f64)");
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED_TRAIT(c, trait);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nb = sizeof(float) * c->p, off = (size_t)trait * c->p;
    if (a) HIPCHK(c, hipMemcpyAsync(c->alpha + off, a, nb, hipMemcpyHostToDevice, c->stream));
    if (b) HIPCHK(c, hipMemcpyAsync(c->beta + off, b, nb, hipMemcpyHostToDevice, c->stream));
    if (d) HIPCHK(c, hipMemcpyAsync((float*)c->delta + off, d, nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

int jwas_hip_get_state(jwas_hip_ctx* c, int32_t trait, float* a, float* b, void* d)
{
    if (c) NOT_F64(c, "jwas_hip_get_state (use jwas_hip_get_state_f64)");
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED_TRAIT(c, trait);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nb = sizeof(float) * c->p, off = (size_t)trait * c->p;
    if (a) HIPCHK(c, hipMemcpyAsync(a, c->alpha + off, nb, hipMemcpyDeviceToHost, c->stream));
    if (b) HIPCHK(c, hipMemcpyAsync(b, c->beta + off, nb, hipMemcpyDeviceToHost, c->stream));
    if (d) HIPCHK(c, hipMemcpyAsync(d, (float*)c->delta + off, nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

int jwas_hip_set_residual(jwas_hip_ctx* c, int32_t trait, const float* rh)
{
    if (c) NOT_F64(c, "jwas_hip_set_residual (use jwas_hip_set_residual_f64)");
    NEED(c, c && rh, JWAS_HIP_EINVAL, "NULL argument");
    NEED_TRAIT(c, trait);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->r + (size_t)trait * c->ld, rh, sizeof(float) * c->n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

int jwas_hip_get_residual(jwas_hip_ctx* c, int32_t trait, float* rh)
{
    if (c) NOT_F64(c, "jwas_hip_get_residual (use jwas_hip_get_residual_f64)");
    NEED(c, c && rh, JWAS_HIP_EINVAL, "NULL argument");
    NEED_TRAIT(c, trait);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(rh, c->r + (size_t)trait * c->ld, sizeof(float) * c->n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

int jwas_hip_residual_dev(jwas_hip_ctx* c, void** rdev, int64_t* ld)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, c->r, JWAS_HIP_ESTATE, "no genotype matrix loaded");
    if (rdev) *rdev = c->r;
    if (ld) *ld = c->ld;
    return JWAS_HIP_OK;
}

int jwas_hip_residual_to_dev(jwas_hip_ctx* c, int32_t trait, void* dst)
{
    NEED(c, c && dst, JWAS_HIP_EINVAL, "NULL argument");
    NEED_TRAIT(c, trait);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst, c->r + (size_t)trait * c->ld, sizeof(float) * c->n, hipMemcpyDeviceToDevice, c->stream));
    return JWAS_HIP_OK;
}

int jwas_hip_residual_from_dev(jwas_hip_ctx* c, int32_t trait, const void* src)
{
    NEED(c, c && src, JWAS_HIP_EINVAL, "NULL argument");
    NEED_TRAIT(c, trait);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->r + (size_t)trait * c->ld, src, sizeof(float) * c->n, hipMemcpyDeviceToDevice, c->stream));
    return JWAS_HIP_OK;
}

// r_k[i] = fl32(fl64(r_k[i]) + shift) for the n individuals (pad rows stay zero): the residual correction of a location
// parameter whose design column is all ones -- the intercept step of the host's Gibbs pass (solver.jl:143-162) without a
// host copy of the residual.
__global__ __launch_bounds__(256) void k_residual_add_scalar(float* __restrict__ r, int64_t n, double shift)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) r[i] = (float)((double)r[i] + shift);
}

int jwas_hip_residual_add_scalar(jwas_hip_ctx* c, int32_t trait, double shift)
{
    if (c) NOT_F64(c, "jwas_hip_residual_add_scalar");
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED_TRAIT(c, trait);
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_residual_add_scalar, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream,
                       c->r + (size_t)trait * c->ld, c->n, shift);
    HIPCHK(c, hipGetLastError());
    return JWAS_HIP_OK;
}

int jwas_hip_residual_sub_xalpha(jwas_hip_ctx* c, int32_t trait)
{
    if (c && IS_F64(c)) {
        NEED_TRAIT(c, trait);
        HIPCHK(c, hipSetDevice(c->device));
        double* rk = c->f64->r + (size_t)trait * c->ld;
        hipLaunchKernelGGL(jw64::k64_mul_alpha, dim3((unsigned)c->nslices), dim3(256), 0, c->stream, c->f64->X, c->ld, c->p,
                           c->f64->alpha + (size_t)trait * c->p, rk, -1.0, (const double*)rk);
        HIPCHK(c, hipGetLastError());
        return JWAS_HIP_OK;
    }
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED_TRAIT(c, trait);
    HIPCHK(c, hipSetDevice(c->device));
    with_cols(c, 0, [&](auto cx) {
        hipLaunchKernelGGL((k_sub_xalpha<decltype(cx)>), dim3(c->nslices), dim3(256), 0, c->stream, cx, c->p,
                           c->alpha + (size_t)trait * c->p, c->r + (size_t)trait * c->ld);
        return 0;
    });
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

// The nonzero effects of trait `trait` as device lists (marker order), compacted on the device; *nnz = -1: dense enough
// that the plain loop over all markers is the better kernel.  The lists live in the context (not to be freed).
static hipError_t compact_alpha(jwas_hip_ctx* c, int32_t trait, int* nnz)
{
    hipError_t e = hipSuccess;
    if (!c->cmp_idx) {
        e = hipMalloc(&c->cmp_idx, sizeof(int32_t) * ((size_t)c->p + 1));
        if (e == hipSuccess) e = hipMalloc(&c->cmp_val, sizeof(float) * (size_t)c->p);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_compact_alpha, dim3(1), dim3(1024), 0, c->stream, c->p, c->alpha + (size_t)trait * c->p,
                       c->cmp_idx, c->cmp_val, c->cmp_idx + c->p);
    e = hipGetLastError();
    int32_t cnt = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&cnt, c->cmp_idx + c->p, sizeof cnt, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    *nnz = (int)cnt;
    return e;
}
static hipError_t sparse_alpha(jwas_hip_ctx* c, int32_t trait, int32_t** d_idx, float** d_val, int* nnz)
{
    hipError_t e = compact_alpha(c, trait, nnz);
    *d_idx = c->cmp_idx; *d_val = c->cmp_val;
    if (e == hipSuccess && (int64_t)*nnz * 4 > c->p) *nnz = -1;          // dense
    return e;
}

int jwas_hip_mul_alpha(jwas_hip_ctx* c, int32_t trait, float* out)
{
    if (c) NOT_F64(c, "jwas_hip_mul_alpha (use jwas_hip_mul_alpha_f64)");
    NEED(c, c && out, JWAS_HIP_EINVAL, "NULL argument");
    NEED_TRAIT(c, trait);
    HIPCHK(c, hipSetDevice(c->device));
    float* tmp = nullptr;
    HIPCHK(c, hipMalloc(&tmp, sizeof(float) * c->ld));
    int32_t* d_idx = nullptr; float* d_val = nullptr; int nnz = -1;
    hipError_t e0 = sparse_alpha(c, trait, &d_idx, &d_val, &nnz);
    if (e0 != hipSuccess) { (void)hipFree(tmp); return fail(c, JWAS_HIP_EHIP, "jwas_hip_mul_alpha: %s", hipGetErrorString(e0)); }
    with_cols(c, 0, [&](auto cx) {
        if (nnz >= 0)
            hipLaunchKernelGGL((k_mul_alpha_list<decltype(cx)>), dim3(c->nslices), dim3(256), 0, c->stream, cx, nnz, d_idx, d_val, tmp);
        else
            hipLaunchKernelGGL((k_mul_alpha<decltype(cx)>), dim3(c->nslices), dim3(256), 0, c->stream, cx, c->p,
                               c->alpha + (size_t)trait * c->p, tmp);
        return 0;
    });
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, tmp, sizeof(float) * c->n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(c, JWAS_HIP_EHIP, "jwas_hip_mul_alpha: %s", hipGetErrorString(e));
    return JWAS_HIP_OK;
}

// Output rows: the reference keeps Mi.output_genotypes = Z_out * genotypes for mme.output_ID (all genotyped individuals by
// default, input_data_validation.jl:150-154; tools4genotypes.jl:290-296) next to the training rows and forms
// EBV = output_genotypes * alpha for every saved sample (output.jl:281-306).
// One saved marker-effect sample as a sparse record (output.jl:443-526 writes the dense text row; with a sparse prior a
// sample has a few hundred nonzero effects among 600 000).  idx / val: caller arrays of `capacity` entries, marker order.
int jwas_hip_get_alpha_sparse(jwas_hip_ctx* c, int32_t trait, int64_t capacity, int32_t* idx, float* val, int64_t* nnz_out)
{
    NEED(c, c && nnz_out, JWAS_HIP_EINVAL, "NULL argument");
    NEED_TRAIT(c, trait);
    HIPCHK(c, hipSetDevice(c->device));
    int nnz = 0;
    HIPCHK(c, compact_alpha(c, trait, &nnz));
    *nnz_out = nnz;
    NEED(c, nnz <= capacity, JWAS_HIP_EINVAL, "%d nonzero effects do not fit the caller's %lld entries", nnz, (long long)capacity);
    if (nnz > 0) {
        NEED(c, idx && val, JWAS_HIP_EINVAL, "idx / val is NULL");
        HIPCHK(c, hipMemcpyAsync(idx, c->cmp_idx, sizeof(int32_t) * nnz, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(val, c->cmp_val, sizeof(float) * nnz, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return JWAS_HIP_OK;
}

int jwas_hip_load_output_dense_f32(jwas_hip_ctx* c, const float* Xh, int64_t n_out, int64_t p, int64_t ld_host)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, Xh, JWAS_HIP_EINVAL, "X_out is NULL");
    NEED(c, HAVE_STORAGE(c), JWAS_HIP_ESTATE, "no genotype matrix loaded");
    NEED(c, n_out >= 1, JWAS_HIP_EINVAL, "n_out must be >= 1 (got %lld)", (long long)n_out);
    NEED(c, p == c->p, JWAS_HIP_EINVAL, "output genotypes have %lld markers, the training matrix %lld", (long long)p, (long long)c->p);
    NEED(c, ld_host >= n_out, JWAS_HIP_EINVAL, "ld_host (%lld) must be >= n_out (%lld)", (long long)ld_host, (long long)n_out);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->Xout); c->Xout = nullptr; c->n_out = c->ld_out = 0;
    const int64_t ld = (n_out + kSliceRows - 1) / kSliceRows * kSliceRows;
    HIPCHK(c, hipMalloc(&c->Xout, (size_t)4 * ld * p));
    c->n_out = n_out; c->ld_out = ld;
    if (ld != n_out) HIPCHK(c, hipMemsetAsync(c->Xout, 0, (size_t)4 * ld * p, c->stream));
    HIPCHK(c, hipMemcpy2DAsync(c->Xout, (size_t)4 * ld, Xh, (size_t)4 * ld_host, (size_t)4 * n_out, (size_t)p,
                               hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

int jwas_hip_mul_alpha_output(jwas_hip_ctx* c, int32_t trait, float* out)
{
    NEED(c, c && out, JWAS_HIP_EINVAL, "NULL argument");
    NEED_TRAIT(c, trait);
    NEED(c, c->Xout, JWAS_HIP_ESTATE, "jwas_hip_load_output_dense_f32 has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    float* tmp = nullptr;
    HIPCHK(c, hipMalloc(&tmp, sizeof(float) * c->ld_out));
    DenseCols cx{c->Xout, c->ld_out, nullptr, 0};
    int32_t* d_idx = nullptr; float* d_val = nullptr; int nnz = -1;
    hipError_t e0 = sparse_alpha(c, trait, &d_idx, &d_val, &nnz);
    if (e0 != hipSuccess) { (void)hipFree(tmp); return fail(c, JWAS_HIP_EHIP, "jwas_hip_mul_alpha_output: %s", hipGetErrorString(e0)); }
    if (nnz >= 0)
        hipLaunchKernelGGL((k_mul_alpha_list<DenseCols>), dim3((unsigned)(c->ld_out / kSliceRows)), dim3(256), 0, c->stream, cx, nnz, d_idx, d_val, tmp);
    else
        hipLaunchKernelGGL((k_mul_alpha<DenseCols>), dim3((unsigned)(c->ld_out / kSliceRows)), dim3(256), 0, c->stream, cx, c->p,
                           c->alpha + (size_t)trait * c->p, tmp);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, tmp, sizeof(float) * c->n_out, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(c, JWAS_HIP_EHIP, "jwas_hip_mul_alpha_output: %s", hipGetErrorString(e));
    return JWAS_HIP_OK;
}

// Window genomic variances of one saved marker-effect sample: the O(samples x n x p) inner loop of the reference's
// window-based GWAS (src/3.GWAS/src/GWAS.jl:152-165: genVar = var(X*alpha), var_w = var(X[:, w] * alpha[w])) as one
// launch over the nonzero effects only.  The caller derives var = (ss - sum^2/n) / (n - 1).
static int window_sums_impl(jwas_hip_ctx* c, int32_t use_output_rows, int32_t nwin, const int32_t* wptr, const int32_t* idx,
                            const float* val, const float* val2, double* const* outs /* 2 or 5 arrays of nwin */)
{
    const int nv = val2 ? 5 : 2;
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, wptr, JWAS_HIP_EINVAL, "NULL argument");
    for (int v = 0; v < nv; ++v) NEED(c, outs[v], JWAS_HIP_EINVAL, "NULL output array");
    NEED(c, nwin >= 1, JWAS_HIP_EINVAL, "nwin must be >= 1 (got %d)", nwin);
    NEED(c, HAVE_STORAGE(c), JWAS_HIP_ESTATE, "no genotype matrix loaded");
    NEED(c, !use_output_rows || c->Xout, JWAS_HIP_ESTATE, "jwas_hip_load_output_dense_f32 has not been called");
    NEED(c, wptr[0] == 0, JWAS_HIP_EINVAL, "wptr[0] must be 0");
    for (int w = 0; w < nwin; ++w) NEED(c, wptr[w + 1] >= wptr[w], JWAS_HIP_EINVAL, "wptr must be non-decreasing");
    const int64_t nnz = wptr[nwin];
    NEED(c, nnz == 0 || (idx && val), JWAS_HIP_EINVAL, "idx / val is NULL");
    for (int64_t e = 0; e < nnz; ++e) NEED(c, idx[e] >= 0 && idx[e] < c->p, JWAS_HIP_EINVAL, "marker index %d out of range", idx[e]);
    HIPCHK(c, hipSetDevice(c->device));
    const int nsl = (int)(use_output_rows ? c->ld_out / kSliceRows : c->nslices);
    const size_t ne = (size_t)(nnz > 0 ? nnz : 1);
    int32_t *d_wptr = nullptr, *d_idx = nullptr; float *d_val = nullptr, *d_val2 = nullptr; double *d_part = nullptr, *d_out = nullptr;
    hipError_t e = hipMalloc(&d_wptr, sizeof(int32_t) * (nwin + 1));
    if (e == hipSuccess) e = hipMalloc(&d_idx, sizeof(int32_t) * ne);
    if (e == hipSuccess) e = hipMalloc(&d_val, sizeof(float) * ne);
    if (e == hipSuccess && val2) e = hipMalloc(&d_val2, sizeof(float) * ne);
    if (e == hipSuccess) e = hipMalloc(&d_part, sizeof(double) * nv * (size_t)nwin * nsl);
    if (e == hipSuccess) e = hipMalloc(&d_out, sizeof(double) * nv * (size_t)nwin);
    if (e == hipSuccess) e = hipMemcpyAsync(d_wptr, wptr, sizeof(int32_t) * (nwin + 1), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && nnz) e = hipMemcpyAsync(d_idx, idx, sizeof(int32_t) * nnz, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && nnz) e = hipMemcpyAsync(d_val, val, sizeof(float) * nnz, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && nnz && val2) e = hipMemcpyAsync(d_val2, val2, sizeof(float) * nnz, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        auto launch = [&](auto cx) {
            if (val2) hipLaunchKernelGGL((k_window_partial<decltype(cx), 5>), dim3((unsigned)nsl), dim3(256), 0, c->stream, cx, nwin, d_wptr, d_idx, d_val, d_val2, d_part);
            else      hipLaunchKernelGGL((k_window_partial<decltype(cx), 2>), dim3((unsigned)nsl), dim3(256), 0, c->stream, cx, nwin, d_wptr, d_idx, d_val, d_val2, d_part);
            return 0;
        };
        if (use_output_rows) launch(DenseCols{c->Xout, c->ld_out, nullptr, 0});
        else with_cols(c, 0, launch);
        hipLaunchKernelGGL(k_window_reduce, dim3((unsigned)((nwin + 255) / 256)), dim3(256), 0, c->stream, nwin, nsl, nv, d_part, d_out);
        e = hipGetLastError();
    }
    for (int v = 0; v < nv && e == hipSuccess; ++v)
        e = hipMemcpyAsync(outs[v], d_out + (size_t)v * nwin, sizeof(double) * nwin, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_wptr); (void)hipFree(d_idx); (void)hipFree(d_val); (void)hipFree(d_val2); (void)hipFree(d_part); (void)hipFree(d_out);
    if (e != hipSuccess) return fail(c, JWAS_HIP_EHIP, "jwas_hip_window_sums: %s", hipGetErrorString(e));
    return JWAS_HIP_OK;
}

int jwas_hip_window_sums(jwas_hip_ctx* c, int32_t use_output_rows, int32_t nwin, const int32_t* wptr, const int32_t* idx,
                         const float* val, double* out_sum, double* out_ss)
{
    double* outs[2] = {out_sum, out_ss};
    return window_sums_impl(c, use_output_rows, nwin, wptr, idx, val, nullptr, outs);
}

int jwas_hip_window_sums2(jwas_hip_ctx* c, int32_t use_output_rows, int32_t nwin, const int32_t* wptr, const int32_t* idx,
                          const float* val1, const float* val2, double* out_sum1, double* out_ss1, double* out_sum2,
                          double* out_ss2, double* out_cross)
{
    NEED(c, c && val2, JWAS_HIP_EINVAL, "NULL argument");
    double* outs[5] = {out_sum1, out_ss1, out_sum2, out_ss2, out_cross};
    return window_sums_impl(c, use_output_rows, nwin, wptr, idx, val1, val2, outs);
}

}  // extern "C" (templates need C++ linkage)

// ---- the sweep --------------------------------------------------------------------------------------
// The step kernel's instantiations live in one translation unit per sampler family (step_launch.hpp).
static inline bool bs_is_dense_big(int bs) { return bs == 256 || bs == 512; }

static StepLaunch step_launch_of(const jwas_hip_ctx* c)
{
    StepLaunch L;
    L.device = c->device; L.block_size = c->block_size; L.nrg = c->nrg;
    L.nblocks = c->nblocks; L.p = c->p; L.stream = c->stream;
    L.d_starts = (const int64_t*)c->d_starts; L.ev_all = c->ev_all;
    L.packed = c->packed;
    L.dc = DenseCols{c->X, c->ld, c->w, (int32_t)c->weighted};
    L.pc = PackedCols{c->Q, c->ld, c->qmean, c->n, c->centered, c->w, (int32_t)c->weighted};
    return L;
}

static hipError_t launch_step_any(jwas_hip_ctx* c, const UpdateArgs& U, const SamplerArgs& S, int do_sample, bool dense)
{
    const StepLaunch L = step_launch_of(c);
    switch (c->method) {
        case JWAS_HIP_BAYESC: case JWAS_HIP_BAYESB: case JWAS_HIP_BAYESR: return launch_step_st(L, c->method, U, S, do_sample, dense);
        case JWAS_HIP_MTBAYESC2: case JWAS_HIP_MTBAYESB2: return launch_step_mt2(L, c->method, c->ntraits, U, S, do_sample, dense);
        case JWAS_HIP_MEGABAYESC: case JWAS_HIP_MEGABAYESB: return launch_step_mega(L, c->method, c->ntraits, U, S, do_sample, dense);
        case JWAS_HIP_MTBAYESB1: return launch_step_mtb1(L, c->ntraits, U, S, do_sample, dense);
        default: return launch_step_mtc1(L, c->ntraits, U, S, do_sample, dense);
    }
}

static int upload_vec(jwas_hip_ctx* c, void** dev, const void* host, size_t bytes)
{
    if (!*dev) HIPCHK(c, hipMalloc(dev, bytes));
    HIPCHK(c, hipMemcpyAsync(*dev, host, bytes, hipMemcpyHostToDevice, c->stream));
    return JWAS_HIP_OK;
}

// Independent-block sweep (BayesABC_block_independent!, BayesABC.jl:190-255): all block RHS from the residual
// snapshot (one pass over X), all blocks sampled concurrently, change lists compacted in (block, marker) order;
// the caller's k_finish applies them to the residual.
static int sweep_independent(jwas_hip_ctx* c, EventList* out, bool dense, int dense_big_off, int compact_off, int nreps)
{
    const int t = c->ntraits, bs = c->block_size;
    const int64_t nb = c->nblocks;
    NEED(c, nb <= 65535, JWAS_HIP_EUNSUP, "independent blocks: at most 65535 blocks (got %lld)", (long long)nb);
    NEED(c, !c->row_mode, JWAS_HIP_EUNSUP, "independent blocks are not available on row shards");
    const int64_t pstride = (int64_t)t * c->nrg * bs;
    if (!c->ev_all || c->ind_traits != t) {
        (void)hipFree(c->ipartials); (void)hipFree(c->ev_all); (void)hipFree(c->ev_offs); (void)hipFree(c->idx_all); (void)hipFree(c->delta_all);
        c->ipartials = nullptr; c->ev_all = nullptr; c->ev_offs = nullptr; c->idx_all = nullptr; c->delta_all = nullptr;
        HIPCHK(c, hipMalloc(&c->ipartials, sizeof(double) * (size_t)nb * pstride));
        HIPCHK(c, hipMalloc(&c->ev_all, sizeof(Events) * (size_t)nb));
        HIPCHK(c, hipMalloc(&c->ev_offs, sizeof(int32_t) * (size_t)(nb + 1)));
        HIPCHK(c, hipMalloc(&c->idx_all, sizeof(int32_t) * (size_t)c->p));
        HIPCHK(c, hipMalloc(&c->delta_all, sizeof(float) * (size_t)kMaxT * c->p));
        c->ind_traits = t;
    }
    HIPCHK(c, hipMemsetAsync(c->corr, 0, sizeof(float) * 2 * kMaxT * (size_t)bs, c->stream));   // corr_in = 0 for every block
    UpdateArgs U;
    std::memset(&U, 0, sizeof U);
    U.r_in = c->r; U.r_out = nullptr;
    U.ev = &c->ev[0];                               // count zeroed by the caller: nothing to apply
    U.nslices = c->upd_nslices; U.nrg = c->nrg; U.ncg = c->ncg; U.spg = c->spg;
    U.partials = c->ipartials; U.bstride = bs;
    SamplerArgs S;
    std::memset(&S, 0, sizeof S);
    S.P = c->dparams; S.partials = c->ipartials; S.nrg = c->nrg; S.bstride = bs;
    S.p = c->p; S.bsz = bs; S.xpx = c->xpx; S.gram = c->gram;
    S.cross_next = c->gram; S.b_next = 0; S.corr_in = c->corr; S.corr_out = c->corr + (size_t)kMaxT * bs;
    S.prep_d = c->prep_d; S.prep_f = c->prep_f; S.mt2_tab = c->mt2_tab; S.lpr_mat = c->lpr_active ? c->lpr_mat : nullptr;
    S.ginv_mat = has_marker_cov(c->method) ? c->ginv_mat : nullptr;
    S.alpha = c->alpha; S.beta = c->beta; S.delta = c->delta;
    S.counters = c->counters;
    S.dense_big_off = dense_big_off;
    S.compact_off = compact_off; S.nreps = nreps;
    hipError_t e;
    const StepLaunch L = step_launch_of(c);
    switch (c->method) {
        case JWAS_HIP_BAYESC: case JWAS_HIP_BAYESB: case JWAS_HIP_BAYESR: e = launch_indep_st(L, c->method, U, S, pstride, dense); break;
        case JWAS_HIP_MTBAYESC2: e = launch_indep_mt2(L, t, U, S, pstride); break;
        case JWAS_HIP_MEGABAYESC: e = launch_indep_mega(L, t, U, S, pstride); break;
        default: e = launch_indep_mtc1(L, t, U, S, pstride);
    }
    HIPCHK(c, e);
    hipLaunchKernelGGL(k_indep_scan, dim3(1), dim3(1024), 0, c->stream, c->ev_all, (int)nb, c->ev_offs, c->ev_offs + nb);
    hipLaunchKernelGGL(k_indep_gather, dim3((unsigned)nb), dim3(256), 0, c->stream, c->ev_all, c->ev_offs, t, c->idx_all, c->delta_all, c->p);
    HIPCHK(c, hipGetLastError());
    *out = EventList{c->ev_offs + nb, c->idx_all, c->delta_all, c->p};
    return JWAS_HIP_OK;
}

// ---- marker-shard reconcile across GPUs (BayesABC.jl:205-253 with one "block" per GPU) -----------
// RCCL is bound lazily (dlopen) so that a single-GPU host never needs it.
#include <dlfcn.h>
namespace {
constexpr int kShardStats = kNStat + 2;           // packed marker statistics + number of effect changes
typedef struct { char internal[128]; } jw_nccl_id;                 // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(jw_nccl_id*) = nullptr;
    int (*CommInitRank)(void**, int, jw_nccl_id, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
    std::string err;
    bool load()
    {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {               // an RCCL the process already holds (e.g. torch's)
            lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (lib) break;
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            if (lib) break;
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!lib) { err = std::string("cannot load librccl.so: ") + dlerror(); return false; }
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(lib, "ncclCommCount"));
        CommUserRank = reinterpret_cast<decltype(CommUserRank)>(dlsym(lib, "ncclCommUserRank"));
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce || !GetErrorString || !CommCount || !CommUserRank) { err = "librccl.so lacks the NCCL entry points"; lib = nullptr; return false; }
        return true;
    }
};
Rccl g_rccl;
constexpr int kNcclFloat64 = 8, kNcclFloat32 = 7, kNcclSum = 0;      // ncclDataType_t, ncclRedOp_t::ncclSum (rccl.h)

// Loopback transport (tests, single-GPU development): the "ranks" are contexts of ONE process driven by different host
// threads; an all-reduce goes through host memory, summed in rank order.  Same call sites as the RCCL one.
struct Loopback {
    std::mutex m;
    std::condition_variable cv;
    int world = 0, arrived = 0;
    uint64_t gen = 0;
    std::vector<std::vector<double>> part;      // per rank
    std::vector<double> sum;
};
Loopback g_loop[4];
}  // namespace

// Sum `count` elements (fp64 or fp32) over the ranks, in place, on the context's stream (RCCL) or through the host
// (loopback).  Every rank receives the same bits.
static int row_allreduce(jwas_hip_ctx* c, void* dev, size_t count, bool f64)
{
    if (c->loop_slot >= 0) {
        Loopback& L = g_loop[c->loop_slot];
        std::vector<double> mine(count);
        if (f64) { HIPCHK(c, hipMemcpyAsync(mine.data(), dev, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
        else {
            std::vector<float> tmp(count);
            HIPCHK(c, hipMemcpyAsync(tmp.data(), dev, sizeof(float) * count, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            for (size_t i = 0; i < count; ++i) mine[i] = tmp[i];
        }
        std::vector<double> total;
        {
            std::unique_lock<std::mutex> lk(L.m);
            const uint64_t g0 = L.gen;
            L.part[(size_t)c->comm_rank] = std::move(mine);
            if (++L.arrived == L.world) {
                L.sum.assign(count, 0.0);
                for (int r = 0; r < L.world; ++r) {
                    if (L.part[(size_t)r].size() != count) { L.arrived = 0; ++L.gen; L.cv.notify_all(); return fail(c, JWAS_HIP_ESTATE, "loopback all-reduce: the ranks disagree on the element count"); }
                    if (f64) for (size_t i = 0; i < count; ++i) L.sum[i] += L.part[(size_t)r][i];
                    else for (size_t i = 0; i < count; ++i) L.sum[i] = (double)((float)L.sum[i] + (float)L.part[(size_t)r][i]);      // fp32 sums, rank order
                }
                L.arrived = 0; ++L.gen;
                L.cv.notify_all();
            } else if (!L.cv.wait_for(lk, std::chrono::seconds(60), [&] { return L.gen != g0; }))
                return fail(c, JWAS_HIP_ESTATE, "loopback all-reduce: the other rank did not arrive within 60 s");
            total = L.sum;
        }
        if (total.size() != count) return fail(c, JWAS_HIP_ESTATE, "loopback all-reduce: the ranks disagree on the element count");
        if (f64) HIPCHK(c, hipMemcpy(dev, total.data(), sizeof(double) * count, hipMemcpyHostToDevice));
        else {
            std::vector<float> tmp(count);
            for (size_t i = 0; i < count; ++i) tmp[i] = (float)total[i];
            HIPCHK(c, hipMemcpy(dev, tmp.data(), sizeof(float) * count, hipMemcpyHostToDevice));
        }
        return JWAS_HIP_OK;
    }
    NEED(c, c->comm, JWAS_HIP_ESTATE, "no communicator attached");
    const int r = g_rccl.AllReduce(dev, dev, count, f64 ? kNcclFloat64 : kNcclFloat32, kNcclSum, c->comm, c->stream);
    NEED(c, r == 0, JWAS_HIP_EHIP, "ncclAllReduce: %s", g_rccl.GetErrorString(r));
    return JWAS_HIP_OK;
}

// buf[k*ld + i] = fl64(r_local) - fl64(r_snapshot) for the t residual vectors; the last workgroup adds this rank's marker
// statistics (the k_marker_stats partials summed in fixed order) and its number of effect changes behind them.
__global__ __launch_bounds__(256) void k_shard_pack(int t, int64_t ld, const float* __restrict__ r_loc, const float* __restrict__ r_snap,
                                                    const double* __restrict__ stat_out, int nstatgrid,
                                                    const unsigned long long* __restrict__ counters, double* __restrict__ buf)
{
    const int64_t total = (int64_t)t * ld;
    if (blockIdx.x == gridDim.x - 1) {
        for (int v = threadIdx.x; v < kNStat; v += 256) {
            // (the additions in ascending order, as the host sums them; 16 independent loads in flight instead of one --
            // 30 us of dependent round trips per sweep at one rank's share of config 2, profiles/r04_rank_share_*)
            double sum = 0.0;
            for (int g0 = 0; g0 < nstatgrid; g0 += 16) {
                double q[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) q[u] = stat_out[(int64_t)(g0 + u < nstatgrid ? g0 + u : nstatgrid - 1) * kNStat + v];
#pragma unroll
                for (int u = 0; u < 16; ++u) if (g0 + u < nstatgrid) sum += q[u];
            }
            buf[total + v] = sum;
        }
        if (threadIdx.x == 0) { buf[total + kNStat] = (double)counters[0]; buf[total + kNStat + 1] = 0.0; }
        return;
    }
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) buf[i] = (double)r_loc[i] - (double)r_snap[i];
}

// r = fl32(r_snapshot + sum over ranks of delta r)  (BayesABC.jl:251-253) and the residual statistics of the reconciled
// residual (the k_finish reductions).  grid = nslices, block = 256: one row per thread.
template <int NT>
__global__ __launch_bounds__(256) void k_shard_apply(const float* __restrict__ w, int64_t ld, const float* __restrict__ r_snap,
                                                     const double* __restrict__ buf, float* __restrict__ r_out, double* __restrict__ out)
{
    __shared__ double red[4 * (NT * NT + NT)];
    const int64_t row = (int64_t)blockIdx.x * kSliceRows + threadIdx.x;
    float rv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        rv[t] = (float)((double)r_snap[t * ld + row] + buf[t * ld + row]);
        r_out[t * ld + row] = rv[t];
    }
    const double wr = (double)w[row];
    double v[NT * NT + NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
#pragma unroll
        for (int c = 0; c < NT; ++c) v[a * NT + c] = ((double)rv[a] * (double)rv[c]) * wr;
        v[NT * NT + a] = (double)rv[a] * wr;
    }
    block_sum<NT * NT + NT>(v, red, 4);
    if (threadIdx.x == 0)
#pragma unroll
        for (int i = 0; i < NT * NT + NT; ++i) out[(int64_t)blockIdx.x * (NT * NT + NT) + i] = v[i];
}

extern "C" {

int jwas_hip_set_kernel_timing(jwas_hip_ctx* c, int32_t stride)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, stride >= 0, JWAS_HIP_EINVAL, "stride must be >= 0");
    c->timing_stride = stride;
    if (stride > 0 && c->event_overhead_ms == 0.0) {
        // An event pair around one launch also counts the dispatch gap.  Measure that gap with an empty
        // kernel launched the same way (after a kernel on the same stream), so callers can report the
        // kernel's own duration; rocprofv3's per-kernel average is the cross-check.
        HIPCHK(c, hipSetDevice(c->device));
        const int reps = 64;
        std::vector<hipEvent_t> ev(2 * reps);
        for (auto& e : ev) HIPCHK(c, hipEventCreate(&e));
        for (int i = 0; i < reps; ++i) {       // enqueued back to back, like the sweep's launches; one sync at the end
            hipLaunchKernelGGL(k_null, dim3(256), dim3(64), 0, c->stream);      // predecessor keeps the queue busy
            HIPCHK(c, hipEventRecord(ev[2 * i], c->stream));
            hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, c->stream);
            HIPCHK(c, hipEventRecord(ev[2 * i + 1], c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        double tot = 0.0;
        for (int i = 8; i < reps; ++i) { float ms = 0.f; HIPCHK(c, hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); tot += ms; }
        c->event_overhead_ms = tot / (reps - 8);
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    return JWAS_HIP_OK;
}

}  // extern "C"

// Everything of a sweep up to the marker statistics, enqueued on the context's stream (no host synchronisation).
static int sweep_enqueue(jwas_hip_ctx* c, const jwas_sweep_params* P, size_t* ntimed_out, double* timed_bytes_out)
{
    NEED(c, c && P, JWAS_HIP_EINVAL, "NULL argument");
    NEED(c, c->method >= 0, JWAS_HIP_ESTATE, "jwas_hip_init_state has not been called");
    NEED(c, c->block_size, JWAS_HIP_ESTATE, "jwas_hip_setup_blocks has not been called");
    NEED(c, P->method == c->method && P->ntraits == c->ntraits, JWAS_HIP_EINVAL,
         "sweep method/ntraits (%d/%d) differ from init_state (%d/%d)", P->method, P->ntraits, c->method, c->ntraits);
    const int t = c->ntraits;
    HIPCHK(c, hipSetDevice(c->device));

    DevParams D;
    c->lpr_active = is_mt_method(c->method) && !is_mega(c->method) && !has_marker_cov(c->method) && P->log_prior_states_matrix != nullptr;
    std::memset(&D, 0, sizeof D);
    D.method = c->method; D.ntraits = t; D.nreps = P->nreps;
    D.iter = P->iteration; D.seed_lo = (uint32_t)P->seed; D.seed_hi = (uint32_t)(P->seed >> 32); D.marker0 = P->marker_offset;
    for (int i = 0; i < t * t; ++i) { D.vare[i] = P->vare[i]; D.var_effect[i] = P->var_effect[i]; }
    const bool marker_cov = has_marker_cov(c->method);
    if (marker_cov) {
        // multi-trait BayesA/B: one effect covariance per marker (locus_effect_variances, MTBayesABC.jl:66; with
        // constraint = true its diagonal, BayesABC.jl:5); inverted on the device by k_prepare (the constrained form keeps the
        // variances), parked in LDS beside the marker's draws
        NEED(c, P->var_effect_matrix || c->var_mat_resident, JWAS_HIP_EINVAL, "multi-trait BayesA/B needs per-marker effect covariances (var_effect_matrix, or jwas_hip_sample_marker_covariances)");
        NEED(c, !P->independent_blocks, JWAS_HIP_EUNSUP, "independent_blocks is not available with per-marker effect covariances");
        NEED(c, !P->log_prior_states_matrix, JWAS_HIP_EUNSUP, "marker-specific joint priors are not available with per-marker effect covariances");
        NEED(c, c->block_size * t <= 2048, JWAS_HIP_EUNSUP, "per-marker effect covariances need block_size * ntraits <= 2048 (got %d x %d)", c->block_size, t);
        const size_t mb = sizeof(float) * (size_t)t * t * c->p;
        if (P->var_effect_matrix) { int rc = upload_vec(c, (void**)&c->var_mat, P->var_effect_matrix, mb); if (rc) return rc; c->var_mat_resident = true; }
        if (!c->ginv_mat) HIPCHK(c, hipMalloc(&c->ginv_mat, mb));
        D.var_mat = c->var_mat; D.ginv_mat = c->ginv_mat;
        for (int i = 0; i < t * t; ++i) D.Ginv[i] = (i / t == i % t) ? 1.f : 0.f;       // (unused)
    }
    if (is_mt_method(c->method) && !is_mega(c->method)) {
        NEED(c, inv_small(P->vare, t, D.Rinv) == 0, JWAS_HIP_EINVAL, "residual covariance matrix is singular");
        if (!marker_cov)
            NEED(c, inv_small(P->var_effect, t, D.Ginv) == 0, JWAS_HIP_EINVAL, "marker effect covariance matrix is singular");
        bool any_finite = false;
        for (int i = 0; i < (1 << t); ++i) { D.log_prior[i] = P->log_prior_states[i]; any_finite = any_finite || std::isfinite(D.log_prior[i]); }
        if ((c->method == JWAS_HIP_MTBAYESC2 || c->method == JWAS_HIP_MTBAYESB2) && !P->log_prior_states_matrix)      // MTBayesABC.jl:190
            NEED(c, any_finite, JWAS_HIP_EINVAL, "All MTBayesABC sampler II state probabilities are zero or invalid.");
        if (P->log_prior_states_matrix) {          // MarkerSpecificPiPrior (MTBayesABC.jl:22-47)
            NEED(c, t == 2, JWAS_HIP_EUNSUP, "marker-specific joint priors support 2 traits (got %d)", t);
            NEED(c, c->block_size <= 512, JWAS_HIP_EUNSUP, "marker-specific joint priors need a block size <= 512 (got %d)", c->block_size);
            int rc = upload_vec(c, (void**)&c->lpr_mat, P->log_prior_states_matrix, sizeof(double) * (size_t)(1 << t) * c->p);
            if (rc) return rc;
        }
    } else if (is_mega(c->method)) {
        // megaBayesABC! (BayesABC.jl:1-8): trait k uses vare[k,k], var_effect[k,k] (BayesA/B: the marker's own) and its own pi (pi_classes[k])
        for (int k = 0; k < t; ++k) {
            NEED(c, P->vare[k * t + k] > 0.f, JWAS_HIP_EINVAL, "residual variance must be positive");
            NEED(c, marker_cov || P->var_effect[k * t + k] > 0.f, JWAS_HIP_EINVAL, "marker effect variance must be positive");
            NEED(c, P->pi_classes[k] >= 0.0 && P->pi_classes[k] <= 1.0, JWAS_HIP_EINVAL, "pi must lie in [0,1]");
            D.pi4[k] = P->pi_classes[k];
        }
    } else {
        NEED(c, P->vare[0] > 0.f, JWAS_HIP_EINVAL, "residual variance must be positive");
    }
    if (c->method == JWAS_HIP_BAYESR) {
        // bayesr_validate_priors / sigmaSq check (BayesR.jl:9-20,50)
        NEED(c, P->var_effect[0] > 0.f, JWAS_HIP_EINVAL, "BayesR sigmaSq must be positive.");
        if (!P->pi_matrix) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) { NEED(c, P->pi_classes[k] >= 0.0, JWAS_HIP_EINVAL, "BayesR pi entries must be nonnegative."); s += P->pi_classes[k]; }
            NEED(c, std::fabs(s - 1.0) <= 1e-8, JWAS_HIP_EINVAL, "BayesR pi must sum to 1.");
        }
        for (int k = 0; k < 4; ++k) { D.pi4[k] = P->pi_classes[k]; D.gamma[k] = P->gamma[k]; }
        if (P->pi_matrix) { int rc = upload_vec(c, (void**)&c->pi_mat, P->pi_matrix, sizeof(double) * 4 * c->p); if (rc) return rc; D.pi_mat = c->pi_mat; }
    } else if (c->method == JWAS_HIP_BAYESC || c->method == JWAS_HIP_BAYESB) {
        D.pi = P->pi;
        if (P->pi_vec) { int rc = upload_vec(c, (void**)&c->pi_vec, P->pi_vec, sizeof(double) * c->p); if (rc) return rc; D.pi_vec = c->pi_vec; }
        if (c->method == JWAS_HIP_BAYESB) {
            NEED(c, P->var_effect_vec, JWAS_HIP_EINVAL, "BayesB needs per-marker effect variances (var_effect_vec)");
            int rc = upload_vec(c, (void**)&c->var_vec, P->var_effect_vec, sizeof(float) * c->p); if (rc) return rc;
            D.var_vec = c->var_vec;
        } else NEED(c, P->var_effect[0] > 0.f, JWAS_HIP_EINVAL, "marker effect variance must be positive");
    }
    HIPCHK(c, hipMemcpyAsync(c->dparams, &D, sizeof D, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(&c->ev[0].count, 0, sizeof(int32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->counters, 0, sizeof(unsigned long long) * kNCounters, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_start, c->stream));

    {   // per-sweep marker constants (draws, prior logs, lhs terms) for all p markers in parallel
        const dim3 pg((unsigned)((c->p + 255) / 256)), pb(256);
        switch (c->method) {
            case JWAS_HIP_BAYESC: hipLaunchKernelGGL((k_prepare<kBayesC, 1>), pg, pb, 0, c->stream, c->dparams, c->p, c->xpx, c->alpha, c->prep_d, c->prep_f); break;
            case JWAS_HIP_BAYESB: hipLaunchKernelGGL((k_prepare<kBayesB, 1>), pg, pb, 0, c->stream, c->dparams, c->p, c->xpx, c->alpha, c->prep_d, c->prep_f); break;
            case JWAS_HIP_BAYESR: hipLaunchKernelGGL((k_prepare<kBayesR, 1>), pg, pb, 0, c->stream, c->dparams, c->p, c->xpx, c->alpha, c->prep_d, c->prep_f); break;
#define JW_MT_PREP(M)                                                                                                                        \
                if (t == 2) hipLaunchKernelGGL((k_prepare<M, 2>), pg, pb, 0, c->stream, c->dparams, c->p, c->xpx, c->alpha, c->prep_d, c->prep_f);          \
                else if (t == 3) hipLaunchKernelGGL((k_prepare<M, 3>), pg, pb, 0, c->stream, c->dparams, c->p, c->xpx, c->alpha, c->prep_d, c->prep_f);     \
                else hipLaunchKernelGGL((k_prepare<M, 4>), pg, pb, 0, c->stream, c->dparams, c->p, c->xpx, c->alpha, c->prep_d, c->prep_f);                 \
                break;
            case JWAS_HIP_MTBAYESC2:
                if (t == 2) hipLaunchKernelGGL((k_prepare_mt2<2>), pg, pb, 0, c->stream, c->dparams, c->p, c->xpx, c->mt2_tab);
                else if (t == 3) hipLaunchKernelGGL((k_prepare_mt2<3>), pg, pb, 0, c->stream, c->dparams, c->p, c->xpx, c->mt2_tab);
                JW_MT_PREP(kMTBayesC2)
            case JWAS_HIP_MEGABAYESC: JW_MT_PREP(kMegaBayesC)
            case JWAS_HIP_MTBAYESB1: JW_MT_PREP(kMTBayesB1)
            case JWAS_HIP_MTBAYESB2: JW_MT_PREP(kMTBayesB2)
            case JWAS_HIP_MEGABAYESB: JW_MT_PREP(kMegaBayesB)
            default: JW_MT_PREP(kMTBayesC1)
#undef JW_MT_PREP
        }
    }

    const int bs = c->block_size;
    const size_t rstride = (size_t)kMaxT * c->ld;
    const size_t pstride = (size_t)bs * c->nrg * kMaxT;
    size_t ntimed = 0;
    double timed_bytes = 0.0;
    const int64_t nb = c->nblocks;
    const bool independent = P->independent_blocks != 0;
    EventList ev_list{nullptr, nullptr, nullptr, 0};
    const float* r_last = c->r;
    // A uniform prior pi = 0 (single-trait BayesA/B/C: RR-BLUP, BayesA, BayesL, the reference's own benchmark setting): the
    // kernel instantiation whose sampler follows Rule D on every path and takes dense_big_st on full 256- / 512-marker blocks
    // (sweep.hpp).  Every marker of every block changes, so the update role shares the apply work (COOP) unless told not to.
    const bool dense_big = (c->method == JWAS_HIP_BAYESC || c->method == JWAS_HIP_BAYESB) && P->pi == 0.0 && P->pi_vec == nullptr;
    // multi-trait sweeps in which most markers changed last time (the reference's default prior: every marker in the model),
    // single pass over <= 128-marker blocks: the dense-walk-only instantiation of the sampler (sampler_role_mt<.., DW>: the same
    // chain, a fraction of the code).  JWAS_HIP_DENSE_MT=0|1 overrides (tests: both instantiations give the same bits).
    const char* edm = std::getenv("JWAS_HIP_DENSE_MT");
    // (the share of markers that changed in the previous sweep from which the dense walk of 256-marker blocks is taken: it costs one
    // step per marker IN the model -- markers outside it that stay there take no step since round 6 -- so it pays far below "most markers
    // change": config 4's chain at 22 000 changes per sweep, 19 ms on the walk against 42 ms through the speculative rounds)
    static const double dense_mt_fraction = std::getenv("JWAS_HIP_DENSE_MT_FRACTION") ? std::atof(std::getenv("JWAS_HIP_DENSE_MT_FRACTION")) : 0.1;
    // ... and full 256-marker blocks of sampler I with one shared covariance (dense_big_mt: decided per launch below)
    const bool dense_mt256 = c->block_size == 256 && (c->method == JWAS_HIP_MTBAYESC1 || c->method == JWAS_HIP_MTBAYESB1) && !P->log_prior_states_matrix;
    const bool dense_mt = is_mt_method(c->method) && !is_sampler2(c->method) && (c->block_size <= 128 || dense_mt256) && P->nreps == 1 && !P->independent_blocks &&
                          (edm ? std::atoi(edm) != 0 : c->last_events >= (dense_mt256 ? dense_mt_fraction : 0.6) * (double)c->p);
    const int dense_big_off = std::getenv("JWAS_HIP_DENSE_BIG_OFF") != nullptr ? 1 : 0;      // (tests: the same chain through the general path)
    // Rule T (jwas_sweep_params.section_solve): the dense chain of a 64-marker section as a mat-vec with the section's inverse,
    // formed here for all sections of the full blocks in parallel (sampler_mt.hpp).  Multi-trait sampler I with <= 3 traits on uniform
    // 256-marker blocks; every other sweep ignores the flag (4 traits: the sampler has no solve path, and a helper workgroup waiting
    // for sections nobody publishes would spin).
    const int64_t solve_blocks = (P->section_solve && t <= 3 && dense_mt256 && P->nreps == 1 && !independent && c->starts.empty() && !c->row_mode && !dense_big_off)
                                     ? c->p / 256 : 0;
    const size_t tsf = (size_t)(64 * t) * (size_t)(64 * t);
    // dense sweeps hand the next block's lookahead correction to a helper workgroup (corr_helper): multi-trait Rule T blocks, and
    // single-trait dense_big_st blocks (uniform pi = 0 on full 256- / 512-marker blocks; JWAS_HIP_CORR_HELPER=0: in the sampler)
    static const int corr_helper_on = std::getenv("JWAS_HIP_CORR_HELPER") ? std::atoi(std::getenv("JWAS_HIP_CORR_HELPER")) : 1;
    const bool st_helper = corr_helper_on && dense_big && P->nreps == 1 && !independent && !c->row_mode && !dense_big_off && (bs_is_dense_big(c->block_size));
    if ((solve_blocks > 0 || st_helper) && !c->xch) {
        HIPCHK(c, hipMalloc(&c->xch, sizeof(unsigned long long) * kMaxT * 512));
        HIPCHK(c, hipMemsetAsync(c->xch, 0, sizeof(unsigned long long) * kMaxT * 512, c->stream));
    }
    if (solve_blocks > 0) {
        const size_t need = (size_t)solve_blocks * 4 * tsf;
        if (c->tsec_cap < need) {
            (void)hipFree(c->tsec); c->tsec = nullptr; c->tsec_cap = 0;
            HIPCHK(c, hipMalloc(&c->tsec, sizeof(float) * need));
            c->tsec_cap = need;
        }
        const StepLaunch L = step_launch_of(c);
        if (c->method == JWAS_HIP_MTBAYESB1) HIPCHK(c, launch_section_inverse_mtb1(L, t, c->dparams, c->xpx, c->gram, c->ginv_mat, solve_blocks * 4, c->tsec));
        else HIPCHK(c, launch_section_inverse_mtc1(L, t, c->dparams, c->xpx, c->gram, solve_blocks * 4, c->tsec));
    }
    const int compact_off = std::getenv("JWAS_HIP_COMPACT_OFF") != nullptr ? std::atoi(std::getenv("JWAS_HIP_COMPACT_OFF")) : 0;          // (tests: the speculative rounds instead of the compact chain)
    // GROUPED LAUNCHES (jwas_sweep_params.group_launch; jwas_hip_setup_groups; sweep.hpp k_group_step): gm blocks per launch.
    // Single-trait single-pass sweeps on uniform blocks; every other sweep ignores the flag.  JWAS_HIP_GROUPS=0 switches it off.
    static const int groups_env = std::getenv("JWAS_HIP_GROUPS") ? std::atoi(std::getenv("JWAS_HIP_GROUPS")) : 1;
    const auto& SET = c->sets[(size_t)c->set_index];
    const bool grouped = P->group_launch != 0 && groups_env != 0 && SET.gm >= 2 && !is_mt_method(c->method) && P->nreps == 1 && !independent &&
                         !c->row_mode && c->starts.empty() && !dense_big;
    int64_t last_launch = nb;                                   // index of the sweep's last launch
    if (independent) {
        int rc = sweep_independent(c, &ev_list, dense_big, dense_big_off, compact_off, P->nreps);
        if (rc) return rc;
    } else if (grouped) {
        const int m = SET.gm;
        const int64_t gb = (int64_t)m * bs, ng = (nb + m - 1) / m;
        const size_t pstride_g = (size_t)gb * c->nrg;           // (fits the ping-pong buffers: gm <= kMaxT, one trait)
        HIPCHK(c, hipMemcpyAsync(c->r + rstride, c->r, sizeof(float) * (size_t)c->ld, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemsetAsync(&c->ev[1].count, 0, sizeof(int32_t), c->stream));
        HIPCHK(c, hipMemsetAsync(SET.gcbuf, 0, sizeof(float) * ((size_t)2 * gb + 5 * (size_t)bs), c->stream));   // group 0 has no predecessor
        // cooperative apply of the merged list (update_role, COOP: the column groups of a row group share the rows instead of every one
        // of them re-reading the changed columns; taken per launch from 32 changes on) in the high-turnover sweeps -- the ones that run
        // the ping-pong samplers: config 3 25.0 -> 24.3 ms, fixed pi 25.6 -> 25.1 ms per sweep.  Same bits (the chain per row is the
        // list's order either way; bounded wait with the redundant apply as its fall-back).  JWAS_HIP_GROUP_COOP=0|1 overrides.
        static const int gcoop_env = std::getenv("JWAS_HIP_GROUP_COOP") ? std::atoi(std::getenv("JWAS_HIP_GROUP_COOP")) : -1;
        const bool gcoop = c->sync_cnt != nullptr && !c->packed &&
                           (gcoop_env >= 0 ? gcoop_env != 0 : (bs <= 512 || c->last_events < 0 || c->last_events > 0.0125 * (double)c->p));
        if (gcoop) HIPCHK(c, hipMemsetAsync(c->sync_cnt, 0, sizeof(int) * 2 * c->nrg, c->stream));
        const int off_w = (int)(2 * gb), off_p = off_w + 2 * bs, off_z = off_p + 2 * bs;
        for (int64_t K = 0; K <= ng; ++K) {
            UpdateArgs U;
            std::memset(&U, 0, sizeof U);
            U.r_in = c->r + ((K + 1) & 1) * rstride; U.r_out = c->r + (K & 1) * rstride;
            U.ev = &c->ev[K & 1];
            const int32_t* uidx = SET.gidx + (K & 1) * gb;
            const float* udel = SET.gdelta + (K & 1) * gb;
            U.j0 = (K < ng) ? K * gb : 0;
            U.b = (K < ng) ? (int)std::min<int64_t>(gb, c->p - U.j0) : 0;
            U.nslices = c->upd_nslices; U.nrg = c->nrg; U.spg = c->spg;
            U.ncg = (U.b > 0 && c->ncg > U.b) ? U.b : c->ncg;
            U.partials = c->partials + (K & 1) * pstride_g; U.bstride = (int)gb;
            U.dbg = c->counters;
            U.sync_now = gcoop ? c->sync_cnt + (K & 1) * c->nrg : nullptr;
            U.sync_next = gcoop ? c->sync_cnt + ((K + 1) & 1) * c->nrg : nullptr;
            {
                static const int qx = std::getenv("JWAS_HIP_QUIET_XCD") ? std::atoi(std::getenv("JWAS_HIP_QUIET_XCD")) : -1;
                U.quiet_xcd = qx >= 0 ? qx : (c->last_events < 0 || c->last_events > 0.0125 * (double)c->p ? 1 : 0);
            }
            GroupSamplers SS;
            GroupArgs G;
            std::memset(&SS, 0, sizeof SS);
            std::memset(&G, 0, sizeof G);
            G.m = m;
            // PING-PONG samplers (2 blocks per launch; sweep.hpp): while the sampler chain is the critical path (many changes per
            // sweep: the regime in which ids = 0 mod 8 do no update work anyway) the pair's second block is sampled by workgroup 8, its
            // front running beside the first block's chain.  Same chain, same bits; JWAS_HIP_PINGPONG=0|1 overrides (1: also in the
            // steady state, with the quiet XCD forced on).
            static const int pp_env = std::getenv("JWAS_HIP_PINGPONG") ? std::atoi(std::getenv("JWAS_HIP_PINGPONG")) : -1;
            // (blocks of <= 512 markers are only ever grouped for the sampler-bound sweeps: there the mode is on whatever the last sweep's
            // count was -- the host keeps those sweeps on 512-marker pairs down to 0.9 % turnover, and without it a pair costs 46.9 instead
            // of 40.2 us at 6 300 changes per sweep)
            const bool pp_want = SET.gpp != nullptr && (pp_env >= 0 ? pp_env != 0 : (U.quiet_xcd != 0 || bs <= 512));
            G.pp_kernel = (pp_want || gcoop) ? 1 : 0;            // (constant over the sweep's launches)
            SS.a[0].bsz = bs;
            if (K >= 1) {
                const int64_t gs = K - 1, first = gs * m;
                const int64_t bb = (int64_t)bs * bs;
                float* cb = SET.gcbuf;
                const int off_g_in = (int)((gs & 1) * gb), off_g_out = (int)(((gs + 1) & 1) * gb);
                G.ns = (int)std::min<int64_t>(m, nb - first);
                G.j0 = first * bs;
                if (gs + 1 < ng) {
                    G.cross_grp = SET.gcross[m == 4 ? 1 : 0] + (size_t)(gs + 1) * gb * gb;
                    G.bn_grp = (int)std::min<int64_t>(gb, c->p - (gs + 1) * gb);
                }
                if (m == 4 && G.ns > 2) {
                    G.cross_pair = SET.gcross[0] + (size_t)gs * (2 * (size_t)bs) * (2 * (size_t)bs);      // (pair 2 gs + 1, stored at gs)
                    G.bn_pair = (int)std::min<int64_t>(2 * (int64_t)bs, c->p - (first + 2) * bs);
                }
                G.corr_g_out = cb + off_g_out; G.corr_p = cb + off_p;
                G.ev_idx = SET.gidx + (gs & 1) * gb; G.ev_delta = SET.gdelta + (gs & 1) * gb;
                unsigned pp_tag = 0;
                // hand-over words (SET.gpp): cW [2][bs] | counts [8] | cP part [2 bs] | cP [2 bs] | cG relay [3][m bs]
                unsigned long long* const pw_cw = SET.gpp, * const pw_cnt = pw_cw + 2 * (size_t)bs, * const pw_ph = pw_cnt + 8,
                                  * const pw_cp = pw_ph + 2 * (size_t)bs, * const pw_h = pw_cp + 2 * (size_t)bs;
                if (pp_want && G.ns >= 2) {
                    G.pp = 1; U.quiet_xcd = 1;
                    c->pp_epoch = (c->pp_epoch + 1) & 0x7fffffffu; if (c->pp_epoch == 0) c->pp_epoch = 1;
                    pp_tag = c->pp_epoch;
                    G.pp_ph = pw_ph; G.pp_cp = pw_cp; G.pp_h = pw_h;
                }
                for (int s2 = 0; s2 < G.ns; ++s2) {
                    SamplerArgs& S = SS.a[s2];
                    const int64_t i = first + s2;
                    S.P = c->dparams;
                    S.partials = c->partials + (gs & 1) * pstride_g + (size_t)s2 * bs; S.nrg = c->nrg; S.bstride = (int)gb;
                    S.j0 = i * bs; S.b = blk_b(c, i); S.p = c->p; S.bsz = bs;
                    S.xpx = c->xpx;
                    S.gram = c->gram + i * bb;
                    const bool inner = !(s2 & 1) && s2 + 1 < G.ns;      // the first block of a PAIR: its sampler forms the cW of the second
                    S.b_next = inner ? blk_b(c, i + 1) : 0;
                    S.cross_next = c->cross + (inner ? i + 1 : i) * bb;
                    S.gram_next = c->gram + (inner ? i + 1 : i) * bb;   // (L2 prefetch only, and only when b_next > 0)
                    S.compact_off = compact_off; S.nreps = P->nreps;
                    S.corr_in = cb + ((s2 & 1) ? off_w : off_z);                          // cW: the pair's first block (or +0)
                    S.corr_out = cb + off_w;
                    S.corr_in2 = cb + off_g_in + (size_t)s2 * bs;                         // cG: the previous group
                    S.corr_in3 = cb + ((s2 >= 2) ? off_p + (s2 - 2) * bs : off_z);        // cP: the group's first pair (or +0)
                    S.prep_d = c->prep_d; S.prep_f = c->prep_f;
                    S.alpha = c->alpha; S.beta = c->beta; S.delta = c->delta;
                    S.ev_out = &c->ev[gs & 1];
                    S.ev_idx = SET.gidx + (gs & 1) * gb; S.ev_delta = SET.gdelta + (gs & 1) * gb;
                    S.counters = c->counters;
                    if (G.pp) {
                        S.pp_tag = pp_tag;
                        S.pp_cw_in = (s2 & 1) ? pw_cw + (size_t)(s2 >> 1) * bs : nullptr;
                        S.pp_cw_out = inner ? pw_cw + (size_t)(s2 >> 1) * bs : nullptr;
                        S.pp_cp_in = (m == 4 && s2 >= 2) ? pw_cp + (size_t)(s2 - 2) * bs : nullptr;
                        S.pp_cnt_in = s2 > 0 ? pw_cnt + (s2 - 1) : nullptr;
                        S.pp_cnt_out = s2 + 1 < G.ns ? pw_cnt + s2 : nullptr;
                    }
                }
            }
            const bool timed = c->timing_stride > 0 && K < ng && (K % c->timing_stride) == 0;
            if (timed) {
                while (c->kev.size() < 2 * (ntimed + 1)) { hipEvent_t e; HIPCHK(c, hipEventCreate(&e)); c->kev.push_back(e); }
                HIPCHK(c, hipEventRecord(c->kev[2 * ntimed], c->stream));
            }
            HIPCHK(c, launch_group_st(step_launch_of(c), c->method, U, uidx, udel, SS, G));
            if (timed) {
                HIPCHK(c, hipEventRecord(c->kev[2 * ntimed + 1], c->stream));
                ++ntimed;
                timed_bytes += 4.0 * (double)c->n * (double)U.b;
            }
        }
        const int64_t par = (ng - 1) & 1;
        ev_list = EventList{&c->ev[par].count, SET.gidx + par * gb, SET.gdelta + par * gb, gb};
        last_launch = ng;
    } else {
    // one-block lookahead pipeline (sweep.hpp): launch k = sampler(block k-1) || update/partial(block k)
    HIPCHK(c, hipMemcpyAsync(c->r + rstride, c->r, sizeof(float) * (size_t)t * c->ld, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(&c->ev[1].count, 0, sizeof(int32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->sync_cnt, 0, sizeof(int) * 2 * c->nrg, c->stream));      // (launch parity restarts with the sweep)
    HIPCHK(c, hipMemsetAsync(c->corr, 0, sizeof(float) * 2 * kMaxT * (size_t)bs, c->stream));   // block 0 has no predecessor
    // dense sweeps (a quarter of the markers or more changed in the previous one; JWAS_HIP_COOP_APPLY=0|1 overrides):
    // the update role's column groups share the work of applying a block's changes (update_role).  The knob is read
    // once per sweep (the tests switch it between sweeps of one process).
    const char* efc = std::getenv("JWAS_HIP_COOP_APPLY");
    const int fc = efc ? std::atoi(efc) : -1;
    const bool coop = c->sync_cnt != nullptr && (fc >= 0 ? fc != 0 : (dense_big || c->last_events > 0.25 * (double)c->p));
    for (int64_t k = 0; k <= nb; ++k) {
        UpdateArgs U;
        U.r_in = c->r + ((k + 1) & 1) * rstride; U.r_out = c->r + (k & 1) * rstride;
        U.ev = &c->ev[k & 1];
        U.j0 = (k < nb) ? blk_j0(c, k) : 0;
        U.b = (k < nb) ? blk_b(c, k) : 0;
        U.nslices = c->upd_nslices; U.nrg = c->nrg; U.spg = c->spg;
        U.ncg = (U.b > 0 && c->ncg > U.b) ? U.b : c->ncg;
        U.partials = c->partials + (k & 1) * pstride; U.bstride = bs;
        U.dbg = c->counters;
        U.sync_now = coop ? c->sync_cnt + (k & 1) * c->nrg : nullptr;
        U.sync_next = coop ? c->sync_cnt + ((k + 1) & 1) * c->nrg : nullptr;
        {   // Placement heuristic (speed only): keep XCD 0 free of streaming traffic for the sampler while the sampler chain
            // is the critical path (many changes per sweep); in the steady state the sampler has slack and all 8 XCDs
            // stream (+4-5 % bandwidth).  Decided from the previous sweep's change count; JWAS_HIP_QUIET_XCD=0|1 overrides.
            static const int qx = std::getenv("JWAS_HIP_QUIET_XCD") ? std::atoi(std::getenv("JWAS_HIP_QUIET_XCD")) : -1;
            U.quiet_xcd = qx >= 0 ? qx : (c->last_events < 0 || c->last_events > 0.0125 * (double)c->p ? 1 : 0);
        }
#ifdef JWAS_HIP_DEV_KNOBS
        { static const int thr = std::getenv("JWAS_HIP_DEBUG_THROTTLE") ? std::atoi(std::getenv("JWAS_HIP_DEBUG_THROTTLE")) : 0; U.dbg_throttle = thr; }
#else
        U.dbg_throttle = 0;
#endif
        SamplerArgs S;
        std::memset(&S, 0, sizeof S);
        const int64_t sb = k - 1;
        if (sb >= 0) {
            S.P = c->dparams;
            S.partials = c->partials + (sb & 1) * pstride; S.nrg = c->nrg; S.bstride = bs;
            S.j0 = blk_j0(c, sb); S.b = blk_b(c, sb); S.p = c->p;
            S.bsz = bs;
            S.xpx = c->xpx;
            S.gram = c->gram + sb * (int64_t)bs * bs;
            // the correction of block sb was written by the sampler of block sb-1 (launch k-1) into corr[sb&1];
            // this sampler writes the one of block sb+1 into corr[(sb+1)&1]
            S.b_next = (sb + 1 < nb) ? blk_b(c, sb + 1) : 0;
            S.cross_next = c->cross + (sb + 1 < nb ? sb + 1 : sb) * (int64_t)bs * bs;
            S.gram_next = (sb + 1 < nb) ? c->gram + (sb + 1) * (int64_t)bs * bs : nullptr;
            S.cross_after = (sb + 2 < nb) ? c->cross + (sb + 2) * (int64_t)bs * bs : nullptr;
            S.dense_big_off = dense_big_off;
            S.compact_off = compact_off; S.nreps = P->nreps;
            S.tsec = (sb < solve_blocks) ? c->tsec + (size_t)sb * 4 * tsf : nullptr;
            S.tsec_next = (sb + 1 < solve_blocks) ? c->tsec + (size_t)(sb + 1) * 4 * tsf : nullptr;
            S.tsec_lines = (int)(4 * tsf / 32);
            if (S.tsec != nullptr || (st_helper && S.b == bs && S.b_next > 0)) {       // (the helper workgroup lives on the quiet XCD: ids = 0 mod 8 do no streaming)
                S.xch = c->xch; c->xch_epoch = (c->xch_epoch + 8) & 0x3fffffff; S.xch_epoch = c->xch_epoch;
                U.quiet_xcd = 1;
            }
            S.lines_after = (sb + 2 < nb) ? (int)(((int64_t)blk_b(c, sb + 1) * blk_b(c, sb + 2) + 31) / 32) : 0;
            S.corr_in = c->corr + (sb & 1) * (size_t)kMaxT * bs;
            S.corr_out = c->corr + ((sb + 1) & 1) * (size_t)kMaxT * bs;
            S.prep_d = c->prep_d; S.prep_f = c->prep_f; S.mt2_tab = c->mt2_tab; S.lpr_mat = c->lpr_active ? c->lpr_mat : nullptr;
            S.ginv_mat = has_marker_cov(c->method) ? c->ginv_mat : nullptr;
            S.alpha = c->alpha; S.beta = c->beta; S.delta = c->delta;
            S.ev_out = &c->ev[(k - 1) & 1];
            S.counters = c->counters;
        }
        const bool timed = c->timing_stride > 0 && k < nb && (k % c->timing_stride) == 0;
        if (timed) {
            while (c->kev.size() < 2 * (ntimed + 1)) { hipEvent_t e; HIPCHK(c, hipEventCreate(&e)); c->kev.push_back(e); }
            HIPCHK(c, hipEventRecord(c->kev[2 * ntimed], c->stream));
        }
        {
            // (256-marker multi-trait blocks: the dense-walk-only instantiation serves FULL blocks; a ragged last one goes through
            // the general instantiation)
            bool dmt = dense_mt;
            if (dmt && c->block_size == 256 && sb >= 0) dmt = !dense_big_off && S.b == 256;
            if (sb >= 0 && S.tsec != nullptr) dmt = true;          // Rule T lives in dense_big_mt: the dense-walk-only instantiation, whatever the last sweep did
            HIPCHK(c, launch_step_any(c, U, S, sb >= 0, dense_big || dmt));
        }
        if (c->row_mode && U.b > 0) {          // the block's partial RHS summed over the ranks' individuals, before its sampler runs
            int rc = row_allreduce(c, U.partials, (size_t)t * c->nrg * bs, true);
            if (rc) return rc;
        }
        if (timed) {
            HIPCHK(c, hipEventRecord(c->kev[2 * ntimed + 1], c->stream));
            ++ntimed;
            timed_bytes += 4.0 * (double)c->n * (double)U.b;
        }
    }
    }   // lookahead pipeline
    if (!independent) {
        if (!grouped) ev_list = event_list(&c->ev[(nb - 1) & 1]);
        r_last = c->r + (last_launch & 1) * rstride;          // r(nb-2), written by the last step
    }
    const int nfin = t * t + t;
    with_cols(c, 0, [&](auto cx) {
    using CX = decltype(cx);
    switch (t) {   // apply the last block's changes; the finished residual always lands in buffer 0
        case 1: hipLaunchKernelGGL((k_finish<1, CX>), dim3(c->nslices), dim3(256), 0, c->stream, cx, r_last, c->r, ev_list, c->fin_out); break;
        case 2: hipLaunchKernelGGL((k_finish<2, CX>), dim3(c->nslices), dim3(256), 0, c->stream, cx, r_last, c->r, ev_list, c->fin_out); break;
        case 3: hipLaunchKernelGGL((k_finish<3, CX>), dim3(c->nslices), dim3(256), 0, c->stream, cx, r_last, c->r, ev_list, c->fin_out); break;
        default: hipLaunchKernelGGL((k_finish<4, CX>), dim3(c->nslices), dim3(256), 0, c->stream, cx, r_last, c->r, ev_list, c->fin_out);
    }
    return 0;
    });
    const double* gamma_dev = reinterpret_cast<const double*>(reinterpret_cast<const char*>(c->dparams) + offsetof(DevParams, gamma));
    switch (t) {
        case 1: hipLaunchKernelGGL((k_marker_stats<1>), dim3(kStatGrid), dim3(256), 0, c->stream, c->method, c->p, c->alpha, c->beta, c->delta, gamma_dev, c->stat_out); break;
        case 2: hipLaunchKernelGGL((k_marker_stats<2>), dim3(kStatGrid), dim3(256), 0, c->stream, c->method, c->p, c->alpha, c->beta, c->delta, gamma_dev, c->stat_out); break;
        case 3: hipLaunchKernelGGL((k_marker_stats<3>), dim3(kStatGrid), dim3(256), 0, c->stream, c->method, c->p, c->alpha, c->beta, c->delta, gamma_dev, c->stat_out); break;
        default: hipLaunchKernelGGL((k_marker_stats<4>), dim3(kStatGrid), dim3(256), 0, c->stream, c->method, c->p, c->alpha, c->beta, c->delta, gamma_dev, c->stat_out);
    }
    HIPCHK(c, hipGetLastError());
    *ntimed_out = ntimed;
    *timed_bytes_out = timed_bytes;
    return JWAS_HIP_OK;
}

// Copies the reductions back and fills the statistics.  packed != NULL: the marker statistics come from the all-reduced
// shard buffer (kShardStats doubles, device) instead of this context's own partials.
static int sweep_collect(jwas_hip_ctx* c, jwas_sweep_stats* S, size_t ntimed, double timed_bytes, const double* packed_dev)
{
    const int t = c->ntraits;
    const int nfin = t * t + t;
    HIPCHK(c, hipEventRecord(c->ev_stop, c->stream));

    double* h_fin = c->host_buf;
    double* h_stat = h_fin + (size_t)c->nslices * nfin;
    unsigned long long* h_cnt = reinterpret_cast<unsigned long long*>(h_stat + (size_t)kStatGrid * kNStat);
    HIPCHK(c, hipMemcpyAsync(h_fin, c->fin_out, sizeof(double) * c->nslices * nfin, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_stat, c->stat_out, sizeof(double) * kStatGrid * kNStat, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_cnt, c->counters, sizeof(unsigned long long) * kNCounters, hipMemcpyDeviceToHost, c->stream));
    if (packed_dev)       // (reuses the head of the statistics staging area: row 0 = the all-rank sums)
        HIPCHK(c, hipMemcpyAsync(h_stat, packed_dev, sizeof(double) * kShardStats, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));

    std::memset(S, 0, sizeof *S);
    for (int s = 0; s < c->nslices; ++s) {
        const double* f = h_fin + (size_t)s * nfin;
        for (int a = 0; a < t; ++a) {
            for (int b2 = 0; b2 < t; ++b2) S->resid_ss[a * t + b2] += f[a * t + b2];
            S->resid_sum[a] += f[t * t + a];
        }
    }
    for (int g = 0; g < (packed_dev ? 1 : kStatGrid); ++g) {
        const double* v = h_stat + (size_t)g * kNStat;
        for (int a = 0; a < t; ++a) S->sum_delta[a] += v[a];
        for (int i = 0; i < t * t; ++i) { S->alpha_ss[i] += v[4 + i]; S->beta_ss[i] += v[20 + i]; }
        for (int k = 0; k < 4; ++k) S->class_counts[k] += v[36 + k];
        S->bayesr_ssq += v[40]; S->bayesr_nnz += v[41];
        for (int q = 0; q < (1 << t) && q < kMaxStates; ++q) S->state_counts[q] += v[42 + q];
    }
    if (c->row_mode) {       // r'r and sum(r) are sums over the ranks' individuals (the marker statistics are replicated)
        double h[kMaxT * kMaxT + kMaxT];
        for (int i = 0; i < t * t; ++i) h[i] = S->resid_ss[i];
        for (int a = 0; a < t; ++a) h[t * t + a] = S->resid_sum[a];
        HIPCHK(c, hipMemcpy(c->row_buf, h, sizeof(double) * (size_t)nfin, hipMemcpyHostToDevice));
        int rc = row_allreduce(c, c->row_buf, (size_t)nfin, true);
        if (rc) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(h, c->row_buf, sizeof(double) * (size_t)nfin, hipMemcpyDeviceToHost));
        for (int i = 0; i < t * t; ++i) S->resid_ss[i] = h[i];
        for (int a = 0; a < t; ++a) S->resid_sum[a] = h[t * t + a];
    }
    S->n_events = packed_dev ? h_stat[kNStat] : (double)h_cnt[0];
    c->last_events = (double)h_cnt[0];
    for (int i = 0; i < kNCounters; ++i) c->last_counters[i] = h_cnt[i];
    NEED(c, h_cnt[kPpTimeoutCounter] == 0, JWAS_HIP_EHIP, "grouped launches: %llu hand-over words between the two sampler workgroups never arrived (results of this sweep are invalid)",
         h_cnt[kPpTimeoutCounter]);
    if (std::getenv("JWAS_HIP_DEBUG_PHASES"))
        std::fprintf(stderr, "[jwas_hip] blocks=%lld events=%llu unstaged=%llu cycles: front=%llu cand=%llu stage=%llu serial=%llu write=%llu corr=%llu rounds=%llu slow_rounds=%llu stage: assign=%llu issue=%llu wait=%llu update wg0: share=%llu wait=%llu rest=%llu compact: blocks=%llu fallback=%llu walk=%llu verify=%llu role=%llu tailwait=%llu xwrite=%llu xchain=%llu (section_solve: blocks = sections solved, fallback = fallen back, walk..xwrite = cycles of y | mat-vec | combine | verify+apply | tail) group: cP=%llu tail=%llu workgroups=%llu last_workgroup=%llu late_new_candidates=%llu mt_taken_over=%llu mt_helped=%llu\n",
                     (long long)c->nblocks, h_cnt[0], h_cnt[1], h_cnt[2], h_cnt[3], h_cnt[4], h_cnt[5], h_cnt[6], h_cnt[9], h_cnt[7], h_cnt[8], h_cnt[10], h_cnt[11], h_cnt[12], h_cnt[13], h_cnt[14], h_cnt[15], h_cnt[16], h_cnt[17], h_cnt[18], h_cnt[19], h_cnt[20], h_cnt[21], h_cnt[22], h_cnt[23], h_cnt[25], h_cnt[26], h_cnt[27], h_cnt[28], h_cnt[29], h_cnt[30], h_cnt[31]);
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev_start, c->ev_stop));
    S->sweep_ms = ms;
    for (size_t i = 0; i < ntimed; ++i) {
        float kms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&kms, c->kev[2 * i], c->kev[2 * i + 1]));
        S->update_kernel_ms += kms;
    }
    S->update_kernel_samples = (double)ntimed;
    S->update_kernel_bytes = timed_bytes;
    S->event_overhead_ms = c->event_overhead_ms;
    return JWAS_HIP_OK;
}


// =====================================================================================================================
// Float64 mode: runMCMC(double_precision=true) (JWAS.jl:349-366, readgenotypes.jl:298,345).  Kernels: csrc/f64_path.hpp.
// A context becomes a Float64 context with jwas_hip_set_precision(ctx, 64) BEFORE genotypes are loaded; it then takes the
// *_f64 data entry points (double host arrays) and the shared control entry points (setup_blocks, init_state, sweep,
// accumulate, residual_sub_xalpha, num_blocks); everything else of the Float32 surface (packed storage, weights, explicit
// partitions, output rows, window sums, shards) answers "not available in a Float64 context".
// =====================================================================================================================

static void f64_free_state(jwas_hip_ctx* c)
{
    auto* F = c->f64;
    for (void* q : {(void*)F->alpha, (void*)F->beta, F->delta, (void*)F->mean_a, (void*)F->mean_a2, (void*)F->mean_d, (void*)F->var_vec}) (void)hipFree(q);
    F->alpha = F->beta = F->mean_a = F->mean_a2 = F->mean_d = F->var_vec = nullptr; F->delta = nullptr;
}

// x'R^-1 x, the Grams X_b'R^-1 X_b of the partition in F->starts (block k at k * bstride^2, row stride = the block's size) and
// the partial-sum buffer.
static int f64_build_blocks(jwas_hip_ctx* c)
{
    auto* F = c->f64;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(F->xpx); (void)hipFree(F->gram); (void)hipFree(F->partials); F->xpx = F->gram = F->partials = nullptr; F->partials_cap = 0;
    const int64_t nb = (int64_t)F->starts.size() - 1;
    int mx = 0;
    for (int64_t k = 0; k < nb; ++k) mx = std::max<int>(mx, (int)(F->starts[(size_t)k + 1] - F->starts[(size_t)k]));
    F->bstride = (mx + 7) / 8 * 8;
    c->nblocks = nb;
    HIPCHK(c, hipMalloc(&F->xpx, sizeof(double) * c->p));
    HIPCHK(c, hipMalloc(&F->gram, sizeof(double) * (size_t)nb * F->bstride * F->bstride));
    F->partials_cap = (size_t)kMaxT * c->nslices * F->bstride;
    HIPCHK(c, hipMalloc(&F->partials, sizeof(double) * F->partials_cap));
    hipLaunchKernelGGL(jw64::k64_xpx, dim3((unsigned)c->p), dim3(256), 0, c->stream, F->X, c->ld, F->w, F->xpx);
    for (int64_t k = 0; k < nb; ++k) {
        const int64_t j0 = F->starts[(size_t)k];
        const int b = (int)(F->starts[(size_t)k + 1] - j0);
        hipLaunchKernelGGL(jw64::k64_gram, dim3((unsigned)b), dim3(256), 0, c->stream, F->X, c->ld, F->w, j0, b, F->gram + k * (int64_t)F->bstride * F->bstride);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

static int f64_setup_blocks(jwas_hip_ctx* c, int32_t bs)
{
    auto* F = c->f64;
    NEED(c, F->X, JWAS_HIP_ESTATE, "no genotype matrix loaded");
    NEED(c, bs >= 1 && bs <= jw64::kMaxBlock64, JWAS_HIP_EINVAL, "Float64 contexts run blocks of 1 to %d markers (got %d)", jw64::kMaxBlock64, bs);
    F->starts.clear();
    for (int64_t j = 0; j < c->p; j += bs) F->starts.push_back(j);
    F->starts.push_back(c->p);
    F->explicit_part = false;
    c->block_size = bs;
    return f64_build_blocks(c);
}

static int f64_setup_blocks_explicit(jwas_hip_ctx* c, const int64_t* starts, int64_t nblocks)
{
    auto* F = c->f64;
    NEED(c, F->X, JWAS_HIP_ESTATE, "no genotype matrix loaded");
    NEED(c, starts && nblocks >= 1, JWAS_HIP_EINVAL, "block starts: NULL or empty");
    NEED(c, starts[0] == 0, JWAS_HIP_EINVAL, "block starts must begin with marker 0");
    int mx = 0;
    for (int64_t k = 0; k < nblocks; ++k) {
        const int64_t hi = (k + 1 < nblocks) ? starts[k + 1] : c->p;
        NEED(c, hi > starts[k] && hi <= c->p, JWAS_HIP_EINVAL, "block starts must be sorted, unique and within the markers (block %lld)", (long long)k);
        NEED(c, hi - starts[k] <= jw64::kMaxBlock64, JWAS_HIP_EINVAL, "block %lld holds %lld markers; blocks hold at most %d markers", (long long)k, (long long)(hi - starts[k]), jw64::kMaxBlock64);
        mx = std::max<int>(mx, (int)(hi - starts[k]));
    }
    F->starts.assign(starts, starts + nblocks);
    F->starts.push_back(c->p);
    F->explicit_part = true;
    c->block_size = mx;
    return f64_build_blocks(c);
}

static int f64_init_state(jwas_hip_ctx* c, int32_t method, int32_t nt)
{
    auto* F = c->f64;
    NEED(c, F->X, JWAS_HIP_ESTATE, "no genotype matrix loaded");
    NEED(c, method == JWAS_HIP_BAYESC || method == JWAS_HIP_BAYESB || method == JWAS_HIP_BAYESR || method == JWAS_HIP_MTBAYESC1, JWAS_HIP_EUNSUP,
         "Float64 contexts run single-trait BayesA/B/C, BayesR and multi-trait sampler I (got method %d)", method);
    if (method == JWAS_HIP_MTBAYESC1) NEED(c, nt >= 2 && nt <= kMaxT, JWAS_HIP_EUNSUP, "multi-trait samplers support 2..%d traits (got %d)", kMaxT, nt);
    else NEED(c, nt == 1, JWAS_HIP_EINVAL, "single-trait method requires ntraits == 1 (got %d)", nt);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    f64_free_state(c);
    c->method = method; c->ntraits = nt;
    const size_t db = sizeof(double) * (size_t)nt * c->p;
    const size_t delb = method == JWAS_HIP_BAYESR ? sizeof(int32_t) * (size_t)c->p : db;
    HIPCHK(c, hipMalloc(&F->alpha, db)); HIPCHK(c, hipMalloc(&F->beta, db)); HIPCHK(c, hipMalloc(&F->delta, delb));
    HIPCHK(c, hipMalloc(&F->mean_a, db)); HIPCHK(c, hipMalloc(&F->mean_a2, db)); HIPCHK(c, hipMalloc(&F->mean_d, db));
    for (void* q : {(void*)F->alpha, (void*)F->beta, (void*)F->mean_a, (void*)F->mean_a2, (void*)F->mean_d}) HIPCHK(c, hipMemsetAsync(q, 0, db, c->stream));
    HIPCHK(c, hipMemsetAsync(F->delta, 0, delb, c->stream));
    HIPCHK(c, hipMemsetAsync(F->r, 0, sizeof(double) * 2 * kMaxT * c->ld, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

// Residual weights of a Float64 context: taken as the Float32 values the C ABI carries, widened (the reference casts whatever
// it is given, JWAS.jl:349-366); x'R^-1 x and the Grams depend on them, so the blocks are rebuilt.
static int f64_set_weights(jwas_hip_ctx* c, const float* rinv32, const double* rinv64)
{
    auto* F = c->f64;
    NEED(c, F->X, JWAS_HIP_ESTATE, "no genotype matrix loaded");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<double> wv((size_t)c->ld, 0.0);
    for (int64_t i = 0; i < c->n; ++i) {
        const double v = rinv64 ? rinv64[i] : rinv32 ? (double)rinv32[i] : 1.0;      // (Float64 weights as they are: build_MME.jl:310)
        NEED(c, std::isfinite(v) && v > 0.0, JWAS_HIP_EINVAL, "residual weights must be positive and finite (row %lld: %g)", (long long)i, v);
        wv[(size_t)i] = v;
    }
    HIPCHK(c, hipMemcpy(F->w, wv.data(), sizeof(double) * (size_t)c->ld, hipMemcpyHostToDevice));
    if (!F->starts.empty()) return f64_build_blocks(c);
    return JWAS_HIP_OK;
}

// One block: its partial sums (r_in -> r_out with the previous block's changes; r_out = NULL, ev_prev = NULL: none), then its
// sampler.  part: the partial-sum buffer of this block.
template <int NT>
static void f64_launch_block(jwas_hip_ctx* c, const double* r_in, double* r_out, const jw64::Events64* ev_prev, int64_t k,
                             double* part, jw64::Events64* ev_out)
{
    auto* F = c->f64;
    const int64_t j0 = F->starts[(size_t)k];
    const int b = (int)(F->starts[(size_t)k + 1] - j0);
    const int bsz = F->bstride;
    // column groups: enough workgroups to fill the chip (one slice = 256 rows), whole batches of 8 columns each
    int ncg = (int)std::max<int64_t>(1, std::min<int64_t>((b + 7) / 8, (512 + c->nslices - 1) / c->nslices));
    const int cpg = (((b + ncg - 1) / ncg) + 7) / 8 * 8;
    ncg = (b + cpg - 1) / cpg;
    hipLaunchKernelGGL((jw64::k64_update_partial<NT>), dim3((unsigned)c->nslices, (unsigned)ncg), dim3(256), 0, c->stream, F->X, c->ld, F->w,
                       r_in, r_out, ev_prev, j0, b, cpg, part, bsz);
    // the block's Gram in LDS only when it FITS beside the per-marker arrays (their stride is the partition's largest block:
    // b = 128 next to a 512-marker block with two traits would ask for 168 KB); otherwise the rows come from L2
    const bool glds = b <= jw64::kGramLds64 && jw64::Smem64(b, bsz, NT, true).bytes <= 160 * 1024;
    const jw64::Smem64 SM(b, bsz, NT, glds);
    const double* G = F->gram + k * (int64_t)bsz * bsz;
#define JW64_SAMPLE(M)                                                                                                                  \
    do {                                                                                                                                \
        if (glds) hipLaunchKernelGGL((jw64::k64_sample<M, NT, true>), dim3(1), dim3(256), SM.bytes, c->stream, F->dparams, G, part, c->nslices, bsz, j0, b, c->p, \
                                     F->xpx, F->alpha, F->beta, F->delta, ev_out, c->counters);                                       \
        else hipLaunchKernelGGL((jw64::k64_sample<M, NT, false>), dim3(1), dim3(256), SM.bytes, c->stream, F->dparams, G, part, c->nslices, bsz, j0, b, c->p, \
                                F->xpx, F->alpha, F->beta, F->delta, ev_out, c->counters);                                            \
    } while (0)
    if constexpr (NT == 1) {
        if (c->method == JWAS_HIP_BAYESC) JW64_SAMPLE(kBayesC);
        else if (c->method == JWAS_HIP_BAYESB) JW64_SAMPLE(kBayesB);
        else JW64_SAMPLE(kBayesR);
    } else JW64_SAMPLE(kMTBayesC1);
#undef JW64_SAMPLE
}

template <int NT>
static hipError_t f64_set_lds_attr()
{
    hipError_t e = hipSuccess;
#define JW64_ATTR(M)                                                                                                                                   \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&jw64::k64_sample<M, NT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
    if (e != hipSuccess) return e;                                                                                                                     \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&jw64::k64_sample<M, NT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    if (e != hipSuccess) return e;
    if constexpr (NT == 1) { JW64_ATTR(kBayesC) JW64_ATTR(kBayesB) JW64_ATTR(kBayesR) }
    else { JW64_ATTR(kMTBayesC1) }
#undef JW64_ATTR
    return e;
}

static int f64_sweep(jwas_hip_ctx* c, const jwas_sweep_params* P, jwas_sweep_stats* S)
{
    auto* F = c->f64;
    NEED(c, c->method >= 0, JWAS_HIP_ESTATE, "jwas_hip_init_state has not been called");
    NEED(c, c->block_size && F->gram, JWAS_HIP_ESTATE, "jwas_hip_setup_blocks has not been called");
    NEED(c, P->method == c->method && P->ntraits == c->ntraits, JWAS_HIP_EINVAL, "sweep method/ntraits (%d/%d) differ from init_state (%d/%d)", P->method, P->ntraits, c->method, c->ntraits);
    NEED(c, !P->log_prior_states_matrix && !P->var_effect_matrix, JWAS_HIP_EUNSUP, "marker-specific multi-trait priors / covariances are not available in a Float64 context");
    const int t = c->ntraits;
    NEED(c, F->bstride * t <= 2048, JWAS_HIP_EUNSUP, "Float64 contexts need block size x traits <= 2048 (got %d x %d)", F->bstride, t);
    HIPCHK(c, hipSetDevice(c->device));
    {   // > 64 KB of dynamic LDS (the block's Gram in doubles)
        static std::atomic<unsigned long long> attr_set{0ull};
        const unsigned long long bit = 1ull << (c->device & 63);
        if (!(attr_set.load(std::memory_order_acquire) & bit)) {
            HIPCHK(c, f64_set_lds_attr<1>()); HIPCHK(c, f64_set_lds_attr<2>()); HIPCHK(c, f64_set_lds_attr<3>()); HIPCHK(c, f64_set_lds_attr<4>());
            attr_set.fetch_or(bit, std::memory_order_release);
        }
    }
    jw64::Params64 D;
    std::memset(&D, 0, sizeof D);
    D.method = c->method; D.ntraits = t; D.nreps = P->nreps;
    D.iter = P->iteration; D.seed_lo = (uint32_t)P->seed; D.seed_hi = (uint32_t)(P->seed >> 32); D.marker0 = P->marker_offset;
    for (int i = 0; i < t * t; ++i) { D.vare[i] = P->vare_f64[i]; D.var_effect[i] = P->var_effect_f64[i]; }
    if (c->method == JWAS_HIP_MTBAYESC1) {
        // inv(vare), inv(G) (MTBayesABC.jl:66-67) in double: Gauss-Jordan with partial pivoting (the oracle's operation order)
        auto inv_d = [&](const double* A, double* Ainv) -> int {
            double M[kMaxT][2 * kMaxT];
            for (int i = 0; i < t; ++i) for (int j = 0; j < t; ++j) { M[i][j] = A[i * t + j]; M[i][t + j] = (i == j); }
            for (int cc = 0; cc < t; ++cc) {
                int piv = cc;
                for (int i = cc + 1; i < t; ++i) if (std::fabs(M[i][cc]) > std::fabs(M[piv][cc])) piv = i;
                if (M[piv][cc] == 0.0) return -1;
                if (piv != cc) for (int j = 0; j < 2 * t; ++j) std::swap(M[cc][j], M[piv][j]);
                const double d = M[cc][cc];
                for (int j = 0; j < 2 * t; ++j) M[cc][j] /= d;
                for (int i = 0; i < t; ++i) if (i != cc) { const double f = M[i][cc]; if (f != 0.0) for (int j = 0; j < 2 * t; ++j) M[i][j] -= f * M[cc][j]; }
            }
            for (int i = 0; i < t; ++i) for (int j = 0; j < t; ++j) Ainv[i * t + j] = M[i][t + j];
            return 0;
        };
        NEED(c, inv_d(D.vare, D.Rinv) == 0, JWAS_HIP_EINVAL, "residual covariance matrix is singular");
        NEED(c, inv_d(D.var_effect, D.Ginv) == 0, JWAS_HIP_EINVAL, "marker effect covariance matrix is singular");
        for (int i = 0; i < (1 << t); ++i) D.log_prior[i] = P->log_prior_states[i];
    } else NEED(c, D.vare[0] > 0.0, JWAS_HIP_EINVAL, "residual variance must be positive");
    if (c->method == JWAS_HIP_BAYESR) {
        NEED(c, D.var_effect[0] > 0.0, JWAS_HIP_EINVAL, "BayesR sigmaSq must be positive.");
        if (!P->pi_matrix) {
            double sum = 0.0;
            for (int k = 0; k < 4; ++k) { NEED(c, P->pi_classes[k] >= 0.0, JWAS_HIP_EINVAL, "BayesR pi entries must be nonnegative."); sum += P->pi_classes[k]; }
            NEED(c, std::fabs(sum - 1.0) <= 1e-8, JWAS_HIP_EINVAL, "BayesR pi must sum to 1.");
        }
        for (int k = 0; k < 4; ++k) { D.pi4[k] = P->pi_classes[k]; D.gamma[k] = P->gamma[k]; }
        if (P->pi_matrix) { int rc = upload_vec(c, (void**)&c->pi_mat, P->pi_matrix, sizeof(double) * 4 * c->p); if (rc) return rc; D.pi_mat = c->pi_mat; }
    } else if (c->method == JWAS_HIP_BAYESC || c->method == JWAS_HIP_BAYESB) {
        D.pi = P->pi;
        if (P->pi_vec) { int rc = upload_vec(c, (void**)&c->pi_vec, P->pi_vec, sizeof(double) * c->p); if (rc) return rc; D.pi_vec = c->pi_vec; }
        if (c->method == JWAS_HIP_BAYESB) {
            NEED(c, P->var_effect_vec_f64, JWAS_HIP_EINVAL, "BayesB needs per-marker effect variances (var_effect_vec_f64)");
            int rc = upload_vec(c, (void**)&F->var_vec, P->var_effect_vec_f64, sizeof(double) * c->p); if (rc) return rc;
            D.var_vec = F->var_vec;
        } else NEED(c, D.var_effect[0] > 0.0, JWAS_HIP_EINVAL, "marker effect variance must be positive");
    }
    HIPCHK(c, hipMemcpyAsync(F->dparams, &D, sizeof D, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->counters, 0, sizeof(unsigned long long) * kNCounters, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_start, c->stream));
    const int64_t nb = c->nblocks;
    const size_t rbuf = (size_t)kMaxT * c->ld;
    auto rb = [&](int64_t parity) { return F->r + (size_t)(parity & 1) * rbuf; };
    const jw64::Events64* fin_lists = nullptr;
    int64_t fin_nlists = 0, fin_parity = 0;
    if (!P->independent_blocks) {
        // launch k reads r(k & 1), applies block k-1's changes and writes r((k + 1) & 1)
        for (int64_t k = 0; k < nb; ++k) {
            const jw64::Events64* prev = k > 0 ? &F->ev[(k - 1) & 1] : nullptr;
            switch (t) {
                case 1: f64_launch_block<1>(c, rb(k), rb(k + 1), prev, k, F->partials, &F->ev[k & 1]); break;
                case 2: f64_launch_block<2>(c, rb(k), rb(k + 1), prev, k, F->partials, &F->ev[k & 1]); break;
                case 3: f64_launch_block<3>(c, rb(k), rb(k + 1), prev, k, F->partials, &F->ev[k & 1]); break;
                default: f64_launch_block<4>(c, rb(k), rb(k + 1), prev, k, F->partials, &F->ev[k & 1]);
            }
        }
        fin_lists = &F->ev[(nb - 1) & 1]; fin_nlists = 1; fin_parity = nb;
    } else {
        // BayesABC_block_independent! (BayesABC.jl:190-255): every block from the SAME residual, reconciled afterwards
        if (F->ev_all_cap < nb) {
            (void)hipFree(F->ev_all); F->ev_all = nullptr; F->ev_all_cap = 0;
            HIPCHK(c, hipMalloc(&F->ev_all, sizeof(jw64::Events64) * (size_t)nb));
            F->ev_all_cap = nb;
        }
        const size_t one = (size_t)kMaxT * c->nslices * F->bstride;
        if (F->partials_cap < one * (size_t)nb) {
            (void)hipFree(F->partials); F->partials = nullptr; F->partials_cap = 0;
            HIPCHK(c, hipMalloc(&F->partials, sizeof(double) * one * (size_t)nb));
            F->partials_cap = one * (size_t)nb;
        }
        for (int64_t k = 0; k < nb; ++k) {
            double* part = F->partials + one * (size_t)k;
            switch (t) {
                case 1: f64_launch_block<1>(c, rb(0), nullptr, nullptr, k, part, F->ev_all + k); break;
                case 2: f64_launch_block<2>(c, rb(0), nullptr, nullptr, k, part, F->ev_all + k); break;
                case 3: f64_launch_block<3>(c, rb(0), nullptr, nullptr, k, part, F->ev_all + k); break;
                default: f64_launch_block<4>(c, rb(0), nullptr, nullptr, k, part, F->ev_all + k);
            }
        }
        fin_lists = F->ev_all; fin_nlists = nb; fin_parity = 0;
    }
    const double* gamma_dev = reinterpret_cast<const double*>(reinterpret_cast<const char*>(F->dparams) + offsetof(jw64::Params64, gamma));
    switch (t) {
        case 1: hipLaunchKernelGGL((jw64::k64_finish<1>), dim3(c->nslices), dim3(256), 0, c->stream, F->X, c->ld, c->n, F->w, rb(fin_parity), rb(0), fin_lists, fin_nlists, c->fin_out);
                hipLaunchKernelGGL((jw64::k64_marker_stats<1>), dim3(kStatGrid), dim3(256), 0, c->stream, c->method, c->p, F->alpha, F->beta, F->delta, gamma_dev, c->stat_out); break;
        case 2: hipLaunchKernelGGL((jw64::k64_finish<2>), dim3(c->nslices), dim3(256), 0, c->stream, F->X, c->ld, c->n, F->w, rb(fin_parity), rb(0), fin_lists, fin_nlists, c->fin_out);
                hipLaunchKernelGGL((jw64::k64_marker_stats<2>), dim3(kStatGrid), dim3(256), 0, c->stream, c->method, c->p, F->alpha, F->beta, F->delta, gamma_dev, c->stat_out); break;
        case 3: hipLaunchKernelGGL((jw64::k64_finish<3>), dim3(c->nslices), dim3(256), 0, c->stream, F->X, c->ld, c->n, F->w, rb(fin_parity), rb(0), fin_lists, fin_nlists, c->fin_out);
                hipLaunchKernelGGL((jw64::k64_marker_stats<3>), dim3(kStatGrid), dim3(256), 0, c->stream, c->method, c->p, F->alpha, F->beta, F->delta, gamma_dev, c->stat_out); break;
        default: hipLaunchKernelGGL((jw64::k64_finish<4>), dim3(c->nslices), dim3(256), 0, c->stream, F->X, c->ld, c->n, F->w, rb(fin_parity), rb(0), fin_lists, fin_nlists, c->fin_out);
                hipLaunchKernelGGL((jw64::k64_marker_stats<4>), dim3(kStatGrid), dim3(256), 0, c->stream, c->method, c->p, F->alpha, F->beta, F->delta, gamma_dev, c->stat_out);
    }
    HIPCHK(c, hipGetLastError());
    return sweep_collect(c, S, 0, 0.0, nullptr);
}

extern "C" {

int jwas_hip_set_precision(jwas_hip_ctx* c, int32_t bits)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, bits == 32 || bits == 64, JWAS_HIP_EINVAL, "precision must be 32 or 64 bits (got %d)", bits);
    NEED(c, !HAVE_STORAGE(c) && !(c->f64 && c->f64->X), JWAS_HIP_ESTATE, "choose the precision before genotypes are loaded");
    if (bits == 64 && !c->f64) c->f64 = new jwas_hip_ctx::F64();
    if (bits == 32 && c->f64) { delete c->f64; c->f64 = nullptr; }
    return JWAS_HIP_OK;
}

int jwas_hip_load_dense_f64(jwas_hip_ctx* c, const double* Xh, int64_t n, int64_t p, int64_t ld_host)
{
    NEED(c, c && Xh, JWAS_HIP_EINVAL, "NULL argument");
    ONLY_F64(c);
    NEED(c, n > 0 && p > 0, JWAS_HIP_EINVAL, "genotype matrix must be non-empty (n=%lld, p=%lld)", (long long)n, (long long)p);
    NEED(c, p < (1ll << 31), JWAS_HIP_EUNSUP, "p=%lld exceeds the 2^31 marker limit of one context", (long long)p);
    NEED(c, ld_host >= n, JWAS_HIP_EINVAL, "ld_host (%lld) < n (%lld)", (long long)ld_host, (long long)n);
    auto* F = c->f64;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    f64_free_state(c);
    for (void* q : {(void*)F->X, (void*)F->r, (void*)F->xpx, (void*)F->gram, (void*)F->partials, (void*)F->ev, (void*)F->dparams, (void*)F->w, (void*)F->ev_all}) (void)hipFree(q);
    F->X = F->r = F->xpx = F->gram = F->partials = F->w = nullptr; F->ev = F->ev_all = nullptr; F->dparams = nullptr;
    F->partials_cap = 0; F->ev_all_cap = 0; F->starts.clear(); F->bstride = 0;
    (void)hipFree(c->counters); (void)hipFree(c->fin_out); (void)hipFree(c->stat_out); if (c->host_buf) (void)hipHostFree(c->host_buf);
    c->counters = nullptr; c->fin_out = c->stat_out = nullptr; c->host_buf = nullptr;
    c->method = -1; c->ntraits = 0; c->block_size = 0; c->nblocks = 0;
    c->n = n; c->p = p; c->ld = round_up(n, kSliceRows); c->nslices = (int)(c->ld / kSliceRows);
    HIPCHK(c, hipMalloc(&F->X, sizeof(double) * (size_t)c->ld * p));
    HIPCHK(c, hipMalloc(&F->r, sizeof(double) * 2 * (size_t)kMaxT * c->ld));                // two buffers: a sweep's launches alternate
    HIPCHK(c, hipMalloc(&F->w, sizeof(double) * (size_t)c->ld));
    {
        std::vector<double> ones((size_t)c->ld, 0.0);
        for (int64_t i = 0; i < n; ++i) ones[(size_t)i] = 1.0;
        HIPCHK(c, hipMemcpy(F->w, ones.data(), sizeof(double) * (size_t)c->ld, hipMemcpyHostToDevice));
    }
    F->partials = nullptr; F->partials_cap = 0;
    HIPCHK(c, hipMalloc(&F->ev, sizeof(jw64::Events64) * 2));
    HIPCHK(c, hipMalloc(&F->dparams, sizeof(jw64::Params64)));
    HIPCHK(c, hipMalloc(&c->counters, sizeof(unsigned long long) * kNCounters));
    const int nfin = kMaxT * kMaxT + kMaxT;
    HIPCHK(c, hipMalloc(&c->fin_out, sizeof(double) * (size_t)c->nslices * nfin));
    HIPCHK(c, hipMalloc(&c->stat_out, sizeof(double) * (size_t)kStatGrid * kNStat));
    HIPCHK(c, hipHostMalloc(&c->host_buf, sizeof(double) * ((size_t)c->nslices * nfin + (size_t)kStatGrid * kNStat) + sizeof(unsigned long long) * kNCounters + 64));
    HIPCHK(c, hipMemsetAsync(F->X, 0, sizeof(double) * (size_t)c->ld * p, c->stream));       // pad rows zero
    HIPCHK(c, hipMemsetAsync(F->ev, 0, sizeof(jw64::Events64) * 2, c->stream));
    HIPCHK(c, hipMemcpy2DAsync(F->X, sizeof(double) * c->ld, Xh, sizeof(double) * ld_host, sizeof(double) * n, (size_t)p, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

int jwas_hip_get_xpx_f64(jwas_hip_ctx* c, double* out)
{
    NEED(c, c && out, JWAS_HIP_EINVAL, "NULL argument");
    ONLY_F64(c);
    NEED(c, c->f64->xpx, JWAS_HIP_ESTATE, "jwas_hip_setup_blocks has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->f64->xpx, sizeof(double) * c->p, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

int jwas_hip_set_state_f64(jwas_hip_ctx* c, int32_t trait, const double* a, const double* b, const void* d)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    ONLY_F64(c);
    NEED_TRAIT(c, trait);
    auto* F = c->f64;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nb = sizeof(double) * c->p, off = (size_t)trait * c->p;
    if (a) HIPCHK(c, hipMemcpyAsync(F->alpha + off, a, nb, hipMemcpyHostToDevice, c->stream));
    if (b) HIPCHK(c, hipMemcpyAsync(F->beta + off, b, nb, hipMemcpyHostToDevice, c->stream));
    if (d) {
        if (c->method == JWAS_HIP_BAYESR) HIPCHK(c, hipMemcpyAsync(F->delta, d, sizeof(int32_t) * c->p, hipMemcpyHostToDevice, c->stream));
        else HIPCHK(c, hipMemcpyAsync((double*)F->delta + off, d, nb, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

int jwas_hip_get_state_f64(jwas_hip_ctx* c, int32_t trait, double* a, double* b, void* d)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    ONLY_F64(c);
    NEED_TRAIT(c, trait);
    auto* F = c->f64;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nb = sizeof(double) * c->p, off = (size_t)trait * c->p;
    if (a) HIPCHK(c, hipMemcpyAsync(a, F->alpha + off, nb, hipMemcpyDeviceToHost, c->stream));
    if (b) HIPCHK(c, hipMemcpyAsync(b, F->beta + off, nb, hipMemcpyDeviceToHost, c->stream));
    if (d) {
        if (c->method == JWAS_HIP_BAYESR) HIPCHK(c, hipMemcpyAsync(d, F->delta, sizeof(int32_t) * c->p, hipMemcpyDeviceToHost, c->stream));
        else HIPCHK(c, hipMemcpyAsync(d, (double*)F->delta + off, nb, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

int jwas_hip_set_residual_f64(jwas_hip_ctx* c, int32_t trait, const double* rh)
{
    NEED(c, c && rh, JWAS_HIP_EINVAL, "NULL argument");
    ONLY_F64(c);
    NEED_TRAIT(c, trait);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->f64->r + (size_t)trait * c->ld, rh, sizeof(double) * c->n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

int jwas_hip_get_residual_f64(jwas_hip_ctx* c, int32_t trait, double* rh)
{
    NEED(c, c && rh, JWAS_HIP_EINVAL, "NULL argument");
    ONLY_F64(c);
    NEED_TRAIT(c, trait);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(rh, c->f64->r + (size_t)trait * c->ld, sizeof(double) * c->n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

int jwas_hip_mul_alpha_f64(jwas_hip_ctx* c, int32_t trait, double* out)
{
    NEED(c, c && out, JWAS_HIP_EINVAL, "NULL argument");
    ONLY_F64(c);
    NEED_TRAIT(c, trait);
    auto* F = c->f64;
    HIPCHK(c, hipSetDevice(c->device));
    double* tmp = nullptr;
    HIPCHK(c, hipMalloc(&tmp, sizeof(double) * c->ld));
    hipLaunchKernelGGL(jw64::k64_mul_alpha, dim3((unsigned)c->nslices), dim3(256), 0, c->stream, F->X, c->ld, c->p, F->alpha + (size_t)trait * c->p, tmp, 1.0,
                       (const double*)nullptr);
    hipError_t e = hipMemcpyAsync(out, tmp, sizeof(double) * c->n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    HIPCHK(c, e);
    return JWAS_HIP_OK;
}

int jwas_hip_get_posterior_f64(jwas_hip_ctx* c, int32_t trait, double* ma, double* ma2, double* md)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    ONLY_F64(c);
    NEED_TRAIT(c, trait);
    auto* F = c->f64;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nb = sizeof(double) * c->p, off = (size_t)trait * c->p;
    if (ma) HIPCHK(c, hipMemcpyAsync(ma, F->mean_a + off, nb, hipMemcpyDeviceToHost, c->stream));
    if (ma2) HIPCHK(c, hipMemcpyAsync(ma2, F->mean_a2 + off, nb, hipMemcpyDeviceToHost, c->stream));
    if (md) HIPCHK(c, hipMemcpyAsync(md, F->mean_d + off, nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

}  // extern "C"

extern "C" {

int jwas_hip_last_sweep_counters(jwas_hip_ctx* c, uint64_t* out, int32_t n)
{
    NEED(c, c && out && n >= 0, JWAS_HIP_EINVAL, "NULL argument");
    for (int i = 0; i < n; ++i) out[i] = i < kNCounters ? (uint64_t)c->last_counters[i] : 0u;
    return JWAS_HIP_OK;
}

int jwas_hip_sweep(jwas_hip_ctx* c, const jwas_sweep_params* P, jwas_sweep_stats* S)
{
    NEED(c, c && P && S, JWAS_HIP_EINVAL, "NULL argument");
    if (c->f64) return f64_sweep(c, P, S);
    size_t ntimed = 0;
    double timed_bytes = 0.0;
    int rc = sweep_enqueue(c, P, &ntimed, &timed_bytes);
    if (rc) return rc;
    rc = sweep_collect(c, S, ntimed, timed_bytes, nullptr);
    return rc;
}

// ---- marker shards over the GPUs of a node ---------------------------------------------------------------
int jwas_hip_comm_unique_id(void* id_out_128)
{
    if (!id_out_128) return fail(nullptr, JWAS_HIP_EINVAL, "jwas_hip_comm_unique_id: NULL argument");
    if (!g_rccl.load()) return fail(nullptr, JWAS_HIP_EUNSUP, "%s", g_rccl.err.c_str());
    jw_nccl_id id;
    const int r = g_rccl.GetUniqueId(&id);
    if (r != 0) return fail(nullptr, JWAS_HIP_EHIP, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
    std::memcpy(id_out_128, &id, sizeof id);
    return JWAS_HIP_OK;
}

int jwas_hip_comm_init(jwas_hip_ctx* c, const void* unique_id_128, int32_t rank, int32_t world)
{
    if (c) NOT_F64(c, "a shard communicator");
    NEED(c, c && unique_id_128, JWAS_HIP_EINVAL, "NULL argument");
    NEED(c, world >= 1 && rank >= 0 && rank < world, JWAS_HIP_EINVAL, "rank %d outside [0,%d)", rank, world);
    NEED(c, HAVE_STORAGE(c), JWAS_HIP_ESTATE, "load this rank's marker columns first");
    NEED(c, !c->comm, JWAS_HIP_ESTATE, "a communicator is already attached (jwas_hip_comm_destroy first)");
    NEED(c, g_rccl.load(), JWAS_HIP_EUNSUP, "%s", g_rccl.err.c_str());
    HIPCHK(c, hipSetDevice(c->device));
    jw_nccl_id id;
    std::memcpy(&id, unique_id_128, sizeof id);
    const int r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != 0) { c->comm = nullptr; return fail(c, JWAS_HIP_EHIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString(r)); }
    c->comm_rank = rank; c->comm_world = world;
    HIPCHK(c, hipMalloc(&c->r_snap, sizeof(float) * (size_t)kMaxT * c->ld));
    HIPCHK(c, hipMalloc(&c->shard_buf, sizeof(double) * ((size_t)kMaxT * c->ld + kShardStats)));
    return JWAS_HIP_OK;
}

// Rank and size of the attached communicator AS THE TRANSPORT REPORTS THEM (ncclCommUserRank / ncclCommCount; the loopback
// transport reports what it was created with): a host that prints "N GPUs" checks it against this, not against its own
// launch arguments.  No communicator: rank 0 of 1.
int jwas_hip_comm_info(jwas_hip_ctx* c, int32_t* rank, int32_t* world)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    int r = 0, w = 1;
    if (c->comm) {
        int e = g_rccl.CommCount(c->comm, &w);
        NEED(c, e == 0, JWAS_HIP_EHIP, "ncclCommCount: %s", g_rccl.GetErrorString(e));
        e = g_rccl.CommUserRank(c->comm, &r);
        NEED(c, e == 0, JWAS_HIP_EHIP, "ncclCommUserRank: %s", g_rccl.GetErrorString(e));
    } else if (c->loop_slot >= 0) { r = c->comm_rank; w = c->comm_world; }
    if (rank) *rank = r;
    if (world) *world = w;
    return JWAS_HIP_OK;
}

int jwas_hip_comm_destroy(jwas_hip_ctx* c)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    if (!c->comm && c->loop_slot < 0) return JWAS_HIP_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    c->comm = nullptr; c->comm_world = 1; c->comm_rank = 0;
    (void)hipFree(c->r_snap); (void)hipFree(c->shard_buf);
    c->r_snap = nullptr; c->shard_buf = nullptr;
    c->row_mode = false; c->loop_slot = -1;
    (void)hipFree(c->row_buf); c->row_buf = nullptr;
    return JWAS_HIP_OK;
}

int jwas_hip_comm_init_loopback(jwas_hip_ctx* c, int32_t slot, int32_t rank, int32_t world)
{
    if (c) NOT_F64(c, "a shard communicator");
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, slot >= 0 && slot < 4, JWAS_HIP_EINVAL, "loopback slot %d outside [0,4)", slot);
    NEED(c, world >= 1 && world <= 8 && rank >= 0 && rank < world, JWAS_HIP_EINVAL, "rank %d outside [0,%d)", rank, world);
    NEED(c, !c->comm && c->loop_slot < 0, JWAS_HIP_ESTATE, "a communicator is already attached (jwas_hip_comm_destroy first)");
    {
        std::lock_guard<std::mutex> lk(g_loop[slot].m);
        if (g_loop[slot].world != world) { g_loop[slot].world = world; g_loop[slot].part.assign((size_t)world, {}); g_loop[slot].arrived = 0; }
    }
    c->loop_slot = slot; c->comm_rank = rank; c->comm_world = world;
    if (HAVE_STORAGE(c)) {          // (marker shards through this transport: the reconcile's buffers, as jwas_hip_comm_init makes them)
        HIPCHK(c, hipSetDevice(c->device));
        if (!c->r_snap) HIPCHK(c, hipMalloc(&c->r_snap, sizeof(float) * (size_t)kMaxT * c->ld));
        if (!c->shard_buf) HIPCHK(c, hipMalloc(&c->shard_buf, sizeof(double) * ((size_t)kMaxT * c->ld + kShardStats)));
    }
    return JWAS_HIP_OK;
}

int jwas_hip_comm_row_shards(jwas_hip_ctx* c, int32_t enable)
{
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, c->comm || c->loop_slot >= 0, JWAS_HIP_ESTATE, "attach a communicator first (jwas_hip_comm_init)");
    NEED(c, c->sets.empty(), JWAS_HIP_ESTATE, "switch the sharding mode before jwas_hip_setup_blocks (x'x and the Grams are summed over the ranks there)");
    HIPCHK(c, hipSetDevice(c->device));
    if (!enable) { c->row_mode = false; return JWAS_HIP_OK; }
    if (!c->row_buf) HIPCHK(c, hipMalloc(&c->row_buf, sizeof(double) * 32));
    // every rank must run the same update-role geometry (the partial RHS are summed row group by row group) and hold the
    // same markers: world * sum(x^2) == (sum x)^2  <=>  all x equal
    double h[6] = {(double)c->nrg, (double)c->nrg * c->nrg, (double)c->p, (double)c->p * (double)c->p, (double)c->packed, 1.0};
    HIPCHK(c, hipMemcpy(c->row_buf, h, sizeof h, hipMemcpyHostToDevice));
    int rc = row_allreduce(c, c->row_buf, 6, true);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(h, c->row_buf, sizeof h, hipMemcpyDeviceToHost));
    const double w = h[5];
    NEED(c, (int)w == c->comm_world, JWAS_HIP_ESTATE, "row shards: %d of %d ranks answered", (int)w, c->comm_world);
    NEED(c, w * h[1] == h[0] * h[0], JWAS_HIP_EINVAL, "row shards need the same number of row groups on every rank (this rank: %d; pad the shorter shards with zero rows)", c->nrg);
    NEED(c, w * h[3] == h[2] * h[2], JWAS_HIP_EINVAL, "row shards need the same markers on every rank (this rank: %lld)", (long long)c->p);
    NEED(c, h[4] == 0.0 || h[4] == w, JWAS_HIP_EINVAL, "row shards need the same storage on every rank");
    c->row_mode = true;
    return JWAS_HIP_OK;
}

// One sweep of this rank's marker shard + the reconcile of BayesABC.jl:205-253 with one "block" per GPU, all on the
// context's stream: snapshot r, sweep the own markers (exact blocked chain from the snapshot), pack
// (fl64(r_local) - fl64(r_snapshot), marker statistics), ONE ncclAllReduce(sum, fp64), r = fl32(r_snapshot + sum).
// On return every rank holds the same residual and the same all-rank statistics.
int jwas_hip_sweep_sharded(jwas_hip_ctx* c, const jwas_sweep_params* P, jwas_sweep_stats* S)
{
    if (c) NOT_F64(c, "a sharded sweep");
    NEED(c, c && P && S, JWAS_HIP_EINVAL, "NULL argument");
    NEED(c, c->comm || c->loop_slot >= 0, JWAS_HIP_ESTATE, "jwas_hip_comm_init has not been called");
    NEED(c, c->r_snap && c->shard_buf, JWAS_HIP_ESTATE, "attach the communicator after the rank's marker columns are loaded");
    NEED(c, !c->row_mode, JWAS_HIP_ESTATE, "this communicator runs exact row shards: use jwas_hip_sweep");
    NEED(c, c->method >= 0, JWAS_HIP_ESTATE, "jwas_hip_init_state has not been called");
    const int t = c->ntraits;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->r_snap, c->r, sizeof(float) * (size_t)t * c->ld, hipMemcpyDeviceToDevice, c->stream));
    size_t ntimed = 0;
    double timed_bytes = 0.0;
    int rc = sweep_enqueue(c, P, &ntimed, &timed_bytes);
    if (rc) return rc;
    const int64_t total = (int64_t)t * c->ld;
    hipLaunchKernelGGL(k_shard_pack, dim3((unsigned)((total + 255) / 256 + 1)), dim3(256), 0, c->stream, t, c->ld, c->r, c->r_snap,
                       c->stat_out, kStatGrid, c->counters, c->shard_buf);
    HIPCHK(c, hipGetLastError());
    rc = row_allreduce(c, c->shard_buf, (size_t)(total + kShardStats), true);      // ncclAllReduce(sum, fp64) on the stream, or the loopback transport
    if (rc) return rc;
    switch (t) {
        case 1: hipLaunchKernelGGL((k_shard_apply<1>), dim3(c->nslices), dim3(256), 0, c->stream, c->w, c->ld, c->r_snap, c->shard_buf, c->r, c->fin_out); break;
        case 2: hipLaunchKernelGGL((k_shard_apply<2>), dim3(c->nslices), dim3(256), 0, c->stream, c->w, c->ld, c->r_snap, c->shard_buf, c->r, c->fin_out); break;
        case 3: hipLaunchKernelGGL((k_shard_apply<3>), dim3(c->nslices), dim3(256), 0, c->stream, c->w, c->ld, c->r_snap, c->shard_buf, c->r, c->fin_out); break;
        default: hipLaunchKernelGGL((k_shard_apply<4>), dim3(c->nslices), dim3(256), 0, c->stream, c->w, c->ld, c->r_snap, c->shard_buf, c->r, c->fin_out);
    }
    HIPCHK(c, hipGetLastError());
    return sweep_collect(c, S, ntimed, timed_bytes, c->shard_buf + total);
}

// ---- multi-trait BayesA/B: the per-marker effect covariances drawn on the device -------------------------------
int jwas_hip_sample_marker_covariances(jwas_hip_ctx* c, double df, const double* scale, uint64_t seed, uint32_t iteration, uint32_t marker_offset)
{
    if (c) NOT_F64(c, "per-marker effect covariances");
    NEED(c, c && scale, JWAS_HIP_EINVAL, "NULL argument");
    NEED(c, has_marker_cov(c->method), JWAS_HIP_ESTATE, "jwas_hip_sample_marker_covariances needs init_state(JWAS_HIP_MTBAYESB1 | JWAS_HIP_MTBAYESB2 | JWAS_HIP_MEGABAYESB, t)");
    const int t = c->ntraits;
    const bool diag = is_mega(c->method);           // constraint = true: scaled inverse chi-square draws of the diagonal only
    NEED(c, df > (diag ? 0.0 : (double)(t - 1)), JWAS_HIP_EINVAL, "inverse-Wishart degrees of freedom must exceed ntraits - 1 (got %g)", df);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t mb = sizeof(float) * (size_t)t * t * c->p;
    if (!c->var_mat) HIPCHK(c, hipMalloc(&c->var_mat, mb));
    IwParams Q;
    std::memset(&Q, 0, sizeof Q);
    Q.df = df; Q.diagonal = diag ? 1 : 0;
    for (int i = 0; i < t * t; ++i) Q.scale[i] = scale[i];
    Q.seed_lo = (uint32_t)seed; Q.seed_hi = (uint32_t)(seed >> 32); Q.iter = iteration; Q.marker0 = marker_offset;
    const dim3 g((unsigned)((c->p + 255) / 256)), b(256);
    if (t == 2) hipLaunchKernelGGL((k_sample_marker_covariances<2>), g, b, 0, c->stream, Q, c->p, c->beta, c->var_mat);
    else if (t == 3) hipLaunchKernelGGL((k_sample_marker_covariances<3>), g, b, 0, c->stream, Q, c->p, c->beta, c->var_mat);
    else hipLaunchKernelGGL((k_sample_marker_covariances<4>), g, b, 0, c->stream, Q, c->p, c->beta, c->var_mat);
    HIPCHK(c, hipGetLastError());
    c->var_mat_resident = true;
    return JWAS_HIP_OK;
}

int jwas_hip_get_marker_covariances(jwas_hip_ctx* c, float* out)
{
    NEED(c, c && out, JWAS_HIP_EINVAL, "NULL argument");
    NEED(c, c->var_mat && c->var_mat_resident, JWAS_HIP_ESTATE, "no per-marker effect covariances are resident");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->var_mat, sizeof(float) * (size_t)c->ntraits * c->ntraits * c->p, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

// ---- posterior accumulators ---------------------------------------------------------------------------
int jwas_hip_accumulate(jwas_hip_ctx* c, double k)
{
    if (c && IS_F64(c)) {
        NEED(c, c->method >= 0, JWAS_HIP_ESTATE, "jwas_hip_init_state has not been called");
        NEED(c, k >= 1.0, JWAS_HIP_EINVAL, "nsamples must be >= 1");
        HIPCHK(c, hipSetDevice(c->device));
        const int64_t count = (int64_t)c->ntraits * c->p;
        hipLaunchKernelGGL(jw64::k64_accumulate, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, count, (int)(c->method == JWAS_HIP_BAYESR), k,
                           c->f64->alpha, c->f64->delta, c->f64->mean_a, c->f64->mean_a2, c->f64->mean_d);
        HIPCHK(c, hipGetLastError());
        return JWAS_HIP_OK;
    }
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED(c, c->method >= 0, JWAS_HIP_ESTATE, "jwas_hip_init_state has not been called");
    NEED(c, k >= 1.0, JWAS_HIP_EINVAL, "nsamples must be >= 1");
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t count = (int64_t)c->ntraits * c->p;
    hipLaunchKernelGGL(k_accumulate, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, count,
                       (int)(c->method == JWAS_HIP_BAYESR), k, c->alpha, c->delta, c->mean_a, c->mean_a2, c->mean_d);
    HIPCHK(c, hipGetLastError());
    return JWAS_HIP_OK;
}

int jwas_hip_get_posterior(jwas_hip_ctx* c, int32_t trait, float* ma, float* ma2, float* md)
{
    if (c) NOT_F64(c, "jwas_hip_get_posterior (use jwas_hip_get_posterior_f64)");
    NEED(c, c, JWAS_HIP_EINVAL, "ctx is NULL");
    NEED_TRAIT(c, trait);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nb = sizeof(float) * c->p, off = (size_t)trait * c->p;
    if (ma) HIPCHK(c, hipMemcpyAsync(ma, c->mean_a + off, nb, hipMemcpyDeviceToHost, c->stream));
    if (ma2) HIPCHK(c, hipMemcpyAsync(ma2, c->mean_a2 + off, nb, hipMemcpyDeviceToHost, c->stream));
    if (md) HIPCHK(c, hipMemcpyAsync(md, c->mean_d + off, nb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return JWAS_HIP_OK;
}

}  // extern "C"
