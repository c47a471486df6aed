// sampler_common.hpp -- SAMPLER role, shared pieces: SamplerArgs, row-group partial sums, the lookahead-correction phases, Gram-row staging and
// the L2 prefetch helpers.  Included by sweep.hpp.
#pragma once
#include "kernels.hpp"

// v_writelane_b32 (lane `lane` of `old` := the uniform value v; the other lanes keep theirs): this clang has no builtin for it --
// the LLVM intrinsic by its own name (the compiler then places the required wait states itself)
extern "C" __device__ int jw_llvm_amdgcn_writelane_i32(int v, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

namespace jw {

// ---------------------------------------------------------------------------------------------
// SAMPLER role, shared front end: rhs assembly, cross-Gram correction, candidate row staging.
// Per-marker running state (rhs, alpha, ...) lives in LDS; the serial wave keeps only the active
// 64-marker sub-block in registers.
// ---------------------------------------------------------------------------------------------
struct SamplerArgs {
    const DevParams* P;
    const double* partials;       // [NT][nrg][bstride] of THIS block
    int nrg, bstride;
    int64_t j0; int b; int64_t p;
    int bsz;                      // nominal block size (LDS strides)
    const float* xpx;
    const float* gram;            // b x b, this block
    const float* cross_next;      // b x b_next: X_this' X_next (row = marker of THIS block); b_next = 0: none
    int b_next;
    const float* gram_next;       // b_next x b_next Gram of the NEXT block (L2 prefetch only), or NULL
    const float* cross_after;     // cross-Gram X_next' X_(next+1) the NEXT launch's sampler reads (L2 prefetch only), or NULL
    int lines_after;              // ... its size in 128-byte lines
    int dense_big_off;            // != 0: never take dense_big_st (tests: the same chain through the general path)
    int compact_off;              // != 0: never take the compact chain (tests: the same chain through the speculative rounds)
    const float* tsec;            // Rule T (jwas_sweep_params.section_solve): the inverses of THIS block's 64-marker sections (see sampler_mt.hpp /
                                  // sampler_st.hpp), or NULL = the sequential chain
    const float* tsec_next;       // ... of the NEXT block (L2 prefetch only), or NULL
    int tsec_lines;               // ... its size in 128-byte lines
    unsigned long long* xch;      // Rule T: [kMaxT][256] the changes of the block's sections, published by the sampler workgroup for the helper
                                  // workgroup (corr_helper_mt): {bits of the change, tag} in ONE 8-byte word, tag = xch_epoch + section + 1
    int xch_epoch;                // ... of THIS launch (grows by 8 per launch: a tag is never seen twice, nothing is reset between launches)
    const float* corr_in;         // [NT][bsz] lookahead correction of THIS block (written by the previous sampler)
    float* corr_out;              // [NT][bsz] lookahead correction of the NEXT block
    const double* prep_d; const float* prep_f;
    const double* mt2_tab;        // sampler II, <= 3 traits: per-marker state tables (k_prepare_mt2), else NULL
    const double* lpr_mat;        // multi-trait: p x 2^t marker-specific log prior of the joint states, else NULL
    const float* ginv_mat;        // multi-trait BayesA/B (kMTBayesB1): p x t x t per-marker G^-1 (k_prepare), else NULL
    float* alpha; float* beta; void* delta;
    Events* ev_out;
    unsigned long long* counters;
    // grouped launches (k_group_step, sweep.hpp; single trait): two further lookahead corrections of THIS block, added to corr_in
    // (always valid pointers there: a zero buffer when a block has none), and the group's merged change list -- this block's
    // changes go behind the ones of the group's earlier blocks, the header line (count, first 7 changes) is ev_out's
    const float* corr_in2; const float* corr_in3;
    int32_t* ev_idx; float* ev_delta;
    // PING-PONG samplers (round 6; k_group_step in the sampler-bound regimes): the blocks of a group are sampled by ONE WORKGROUP EACH
    // (block s by workgroup 8 s, all idle on the sampler's XCD), every front -- state, constants, row-group partial sums: one memory
    // latency -- running at launch start.  What a block needs from the blocks before it travels as TAGGED 8-byte words
    // {tag << 32 | payload} (one relaxed agent-scope store each, self-validating: the reader polls the word it needs until the tag is
    // this launch's; nothing is ever reset, a tag is never seen twice).  One-directional: a workgroup only ever waits for lower ones.
    const unsigned long long* pp_cw_in;   // [bsz] cW of THIS block from the workgroup of the pair's first block, or NULL (corr_in is read)
    const unsigned long long* pp_cp_in;   // [bsz] cP of THIS block from the workgroup of block 1 (4 blocks per launch), or NULL (corr_in3)
    unsigned long long* pp_cw_out;        // [bsz] where this block posts the cW of the pair's second block, or NULL
    const unsigned long long* pp_cnt_in;  // the number of list entries in front of this block's (posted by the block before), or NULL = ev_base
    unsigned long long* pp_cnt_out;       // where this block posts the list length behind it (its entries acknowledged), or NULL = last block
    unsigned pp_tag;                      // this launch's tag
    int nreps;                            // = P->nreps (jwas_sweep_params.nreps), from the argument segment: control flow that precedes the front's
                                          // loads must not wait for a load through P (a memory latency under the stream, sampler_role_mt)
};

constexpr int kPpTimeoutCounter = 24;      // sweep counter: hand-over words that never arrived (must stay 0; the host fails the sweep otherwise)
// Poll one tagged word until it carries this launch's tag.  Bounded (~1 s): a hand-over that never arrives is reported, never a hang.
__device__ __forceinline__ unsigned pp_wait_word(const unsigned long long* w, unsigned tag, unsigned long long* counters)
{
    unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while ((unsigned)(v >> 32) != tag) {
        if (++spins > (1 << 21)) { atomicAdd(&counters[kPpTimeoutCounter], 1ull); break; }
        __builtin_amdgcn_s_sleep(8);
        v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return (unsigned)v;
}
__device__ __forceinline__ void pp_post_word(unsigned long long* w, unsigned tag, unsigned payload)
{
    __hip_atomic_store(w, ((unsigned long long)tag << 32) | (unsigned long long)payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ping-pong: the number of list entries in front of this block's (posted by the block before with its entries acknowledged)
__device__ __forceinline__ int pp_count_in(const SamplerArgs& A) { return A.pp_cnt_in != nullptr ? (int)pp_wait_word(A.pp_cnt_in, A.pp_tag, A.counters) : 0; }

// fp64 sum of one column's row-group partials in fixed (ascending row group) order; the first N loads are issued
// back to back from clamped addresses (no load depends on another).
template <int N>
__device__ __forceinline__ double sum_partials_n(const double* pp, int nrg, int64_t stride)
{
    double v[N];
#pragma unroll
    for (int u = 0; u < N; ++u) v[u] = *(pp + (int64_t)(u < nrg ? u : nrg - 1) * stride);
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < N; ++u) if (u < nrg) sum += v[u];
    for (int rg = N; rg < nrg; rg += 16) {                        // very tall matrices only
        double w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = *(pp + (int64_t)(rg + u < nrg ? rg + u : nrg - 1) * stride);
#pragma unroll
        for (int u = 0; u < 16; ++u) if (rg + u < nrg) sum += w[u];
    }
    return sum;
}
// The same in two steps, so that a caller can ISSUE the loads of several columns (and whatever else it needs) before the first sum
// waits for any of them: load_partials_n issues the first N loads (clamped addresses), sum_loaded_n is the rest of sum_partials_n --
// the same additions in the same order.
template <int N>
__device__ __forceinline__ void load_partials_n(const double* pp, int nrg, int64_t stride, double (&v)[N])
{
#pragma unroll
    for (int u = 0; u < N; ++u) v[u] = *(pp + (int64_t)(u < nrg ? u : nrg - 1) * stride);
}
template <int N>
__device__ __forceinline__ double sum_loaded_n(const double (&v)[N], const double* pp, int nrg, int64_t stride)
{
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < N; ++u) if (u < nrg) sum += v[u];
    if (nrg > N)
    for (int rg = N; rg < nrg; rg += 16) {                        // very tall matrices only
        double w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = *(pp + (int64_t)(rg + u < nrg ? rg + u : nrg - 1) * stride);
#pragma unroll
        for (int u = 0; u < 16; ++u) if (rg + u < nrg) sum += w[u];
    }
    return sum;
}
__device__ __forceinline__ double sum_partials(const double* pp, int nrg, int64_t stride)
{
    if (nrg <= 8) return sum_partials_n<8>(pp, nrg, stride);      // (uniform branches: nrg is a launch constant)
    if (nrg <= 16) return sum_partials_n<16>(pp, nrg, stride);
    return sum_partials_n<32>(pp, nrg, stride);
}

// All NT traits of one column at once: the loads of a chunk of row groups are issued back to back for every trait (one
// memory latency per chunk instead of one per trait), the sums per trait in the same ascending order as sum_partials.
template <int NT>
__device__ __forceinline__ void sum_partials_traits(const double* pp, int64_t tstride, int nrg, int64_t stride, double (&sum)[NT])
{
    constexpr int kC = (NT <= 2) ? 16 : (NT == 3 ? 12 : 8);
    // the first chunk as STRAIGHT-LINE code: in front of a loop that holds loads the compiler waits for every load in flight
    // (s_waitcnt vmcnt(0) at the loop's entry) -- the markers' state loads issued just before, i.e. a second memory latency in the
    // sampler's front.  The loop below is only entered by very tall matrices (more row groups than one chunk).
    {
        double v[NT][kC];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int u = 0; u < kC; ++u) v[t][u] = *(pp + t * tstride + (int64_t)(u < nrg ? u : nrg - 1) * stride);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            sum[t] = 0.0;
#pragma unroll
            for (int u = 0; u < kC; ++u) if (u < nrg) sum[t] += v[t][u];
        }
    }
    if (nrg > kC)
    for (int rg = kC; rg < nrg; rg += kC) {
        double v[NT][kC];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int u = 0; u < kC; ++u) v[t][u] = *(pp + t * tstride + (int64_t)(rg + u < nrg ? rg + u : nrg - 1) * stride);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int u = 0; u < kC; ++u) if (rg + u < nrg) sum[t] += v[t][u];
    }
}

// End of the sampler role (all threads): the lookahead correction of the NEXT block from the net changes
// of this one,  corr[c] = fmaf(d_e, C[e][c], corr[c])  from 0 in marker order (C = X_this' X_next).
// fin (LDS, int2 {local column, bits(d)} per trait-0 ... ) holds the compact change list; dlds the
// per-trait changes [NT][B] indexed by local column.
template <int NT>
__device__ __forceinline__ void corr_phase(char* smem, const StepSmem& SM, const SamplerArgs& A, int nfin, bool cross_in_lds = false)
{
    const int B = SM.B;
    const int* fin = reinterpret_cast<const int*>(smem + SM.log_off);              // local columns, marker order
    const float* acur = reinterpret_cast<const float*>(smem + SM.acur_off);
    const float* astart = reinterpret_cast<const float*>(smem + SM.astart_off);
    const int bn = A.b_next;
    for (int c = threadIdx.x; c < B; c += kStepThreads) {
        float corr[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) corr[t] = 0.f;
        if (c < bn) {
            if (cross_in_lds) {                                  // rows copied by copy_cross_rows during the serial phase
                const float* crossL = reinterpret_cast<const float*>(smem + SM.cross_off);
                for (int e = 0; e < nfin; ++e) {
                    const int ce = fin[e];
                    const float g = crossL[ce * B + c];
#pragma unroll
                    for (int t = 0; t < NT; ++t) corr[t] = fmaf(astart[t * B + ce] - acur[t * B + ce], g, corr[t]);
                }
            } else
            for (int e0 = 0; e0 < nfin; e0 += 16) {
                float g[16];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    g[u] = A.cross_next[(int64_t)fin[e0 + u < nfin ? e0 + u : nfin - 1] * bn + c];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (e0 + u < nfin) {
                        const int ce = fin[e0 + u];
#pragma unroll
                        for (int t = 0; t < NT; ++t) corr[t] = fmaf(astart[t * B + ce] - acur[t * B + ce], g[u], corr[t]);
                    }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) A.corr_out[t * B + c] = corr[t];
    }
}

// Waves 1..4 of the sampler workgroup in single-pass sweeps: the lookahead correction of the NEXT block,
//   corr[c] = fmaf(d_e, C[e][c], corr[c])  from 0 over the changed markers e in marker order  (C = X_this' X_next),
// accumulated WHILE the serial wave runs.  The serial wave commits in marker order and publishes every change
// {local column, alpha_old - alpha_new} to a log in LDS with one 8-byte write (entries are pre-set to column -1; wc[13] is
// set after the last one); each helper lane owns four columns of the next block and consumes the log as it grows -- the
// cross-Gram rows come from L2 (prefetch_cross_rows) or HBM, off the critical path.  When the serial wave is done the
// correction is (nearly) done too: no dependent fetch of the changed markers' rows at the end of the block.
// Spinning on LDS inside one workgroup is safe: all its waves are resident.
// The helpers are waves 1..4 (keeping wave 4 -- the serial wave's SIMD -- idle instead was measured: no difference);
// waves 5..7 prefetch.
__device__ __forceinline__ bool is_corr_helper(int wave) { return wave >= 1 && wave <= 4; }
__device__ __forceinline__ int corr_helper_index(int wave) { return wave - 1; }
__device__ __forceinline__ float4 stream_corr_role(char* smem, const StepSmem& SM, const SamplerArgs& A)
{
    const int B = SM.B, bn = A.b_next;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int* wc = reinterpret_cast<int*>(smem + SM.wcnt_off);
    const int2* plog = reinterpret_cast<const int2*>(smem + SM.log_off);
    const int col = (corr_helper_index(wave) * 64 + lane) * 4;
    const bool vec = (bn & 3) == 0;                         // full next block: rows 16-byte aligned
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    int done = 0;
    const volatile int* vcol = reinterpret_cast<const volatile int*>(plog);       // entry e: {column, bits(d)}; column -1 = not written yet
    while (true) {
        const int fin = __hip_atomic_load(&wc[13], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        // the serial wave publishes an entry with ONE 8-byte LDS write (no counter, no wait on its side): count the
        // valid entries after `done` (at most 8 per visit)
        int n = done;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (n == done + u && n < B && vcol[2 * n] >= 0) ++n;
        if (n > done) {
            if (col < bn) {
                for (int e0 = done; e0 < n; e0 += 8) {
                    int2 le[8];
                    float4 g[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) le[u] = plog[e0 + u < n ? e0 + u : n - 1];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float* src = A.cross_next + (int64_t)le[u].x * bn + col;
                        if (vec) g[u] = *reinterpret_cast<const float4*>(src);
                        else { g[u].x = src[0]; g[u].y = src[col + 1 < bn ? 1 : 0]; g[u].z = src[col + 2 < bn ? 2 : 0]; g[u].w = src[col + 3 < bn ? 3 : 0]; }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (e0 + u < n) {
                            const float d = __int_as_float(le[u].y);
                            c0 = fmaf(d, g[u].x, c0); c1 = fmaf(d, g[u].y, c1); c2 = fmaf(d, g[u].z, c2); c3 = fmaf(d, g[u].w, c3);
                        }
                }
            }
            done = n;
        } else if (fin) break;
        else __builtin_amdgcn_s_sleep(8);                  // ~500 cycles between polls
    }
    return float4{c0, c1, c2, c3};      // stored by the caller with the role's other global stores (after the barrier)
}

// Waves 1..7 (after their other prefetch work, while wave 0 runs the serial phase): pull the Gram rows the NEXT block's
// sampler will stage into this XCD's L2 -- the whole Gram block for small (dense-prior) blocks, else the rows of the
// markers that are in the model (alpha != 0: always candidates).  A row fetch of the sampler workgroup competes with
// ~220 streaming workgroups for HBM; here it is off the critical path, in the next launch it is an L2 hit.  Speed only.
// stop (LDS, may be NULL): set by the serial wave when it is done -- prefetching is optional work and must never hold the
// workgroup's barrier back.
__device__ __forceinline__ void prefetch_next_gram(const SamplerArgs& A, bool whole_block, int w0 = 1, const int* stop = nullptr)
{
    const int lane = threadIdx.x & 63;
    const int bn = A.b_next;
    const int nw = kStepThreads / 64 - w0;                 // waves w0 .. 7 do the work
    if ((int)(threadIdx.x >> 6) < w0 || bn <= 0 || A.gram_next == nullptr) return;
    const int wave = (int)(threadIdx.x >> 6) - w0 + 1;     // 1 .. nw
    float sink = 0.f;
    if (whole_block) {
        const int nlines = (bn * bn + 31) / 32;                                    // 128-byte lines of the next Gram block
        float v[4];                                                                // (<= 4 x 448 lines: a 128 x 128 block has 512)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int l = (wave - 1) * 64 + lane + u * nw * 64;
            v[u] = A.gram_next[(int64_t)(l < nlines ? l : nlines - 1) * 32];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) sink += v[u];
    } else {
        // every lane whose marker is in the model touches the lines of ITS row: independent loads, one wait at the end
        const int lines_per_row = (bn + 31) / 32;                                  // <= 32 for 1024-marker blocks
        for (int c0 = (wave - 1) * 64; c0 < bn; c0 += nw * 64) {
            if (stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
            const int c = c0 + lane;
            const float a = A.alpha[A.j0 + A.b + (c < bn ? c : 0)];
            if (c < bn && a != 0.f) {
                const float* row = A.gram_next + (int64_t)c * bn;
                for (int l0 = 0; l0 < lines_per_row; l0 += 8) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = row[(l0 + u < lines_per_row ? l0 + u : lines_per_row - 1) * 32];
#pragma unroll
                    for (int u = 0; u < 8; ++u) sink += v[u];
                }
            }
        }
    }
    asm volatile("" ::"v"(sink));
}

// Waves 1..7: touch the cross-Gram rows (X_this' X_next) of the staged candidates so that corr_phase finds
// them in L2 instead of paying an HBM round trip at the end of the chain.
__device__ __forceinline__ void prefetch_cross_rows(char* smem, const StepSmem& SM, const SamplerArgs& A, int ncand, int w0 = 1, const int* stop = nullptr)
{
    const short* cand_list = reinterpret_cast<const short*>(smem + SM.cand_off);
    const int lane = threadIdx.x & 63;
    const int bn = A.b_next;
    const int nw = kStepThreads / 64 - w0;                 // waves w0 .. 7 do the work
    if ((int)(threadIdx.x >> 6) < w0 || bn <= 0) return;
    const int wave = (int)(threadIdx.x >> 6) - w0 + 1;     // 1 .. nw
    const int nchunk = (bn + 63) / 64, ntask = ncand * nchunk;
    float sink = 0.f;
    for (int t0 = (wave - 1) * 8; t0 < ntask; t0 += nw * 8) {
        if (stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int task = (t0 + u < ntask) ? t0 + u : ntask - 1;
            const int row = task / nchunk, c = (task - row * nchunk) * 64 + lane;
            v[u] = A.cross_next[(int64_t)cand_list[row] * bn + (c < bn ? c : 0)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) sink += v[u];
    }
    asm volatile("" ::"v"(sink));
}

// Waves 1..7 (small blocks): copy the block's cross-Gram rows X_this' X_next (b rows x bn columns) into LDS while wave 0
// runs the serial phase; corr_phase then needs no global access at the end of the chain.
__device__ __forceinline__ void copy_cross_rows(char* smem, const StepSmem& SM, const SamplerArgs& A)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bn = A.b_next, b = A.b, B = SM.B;
    if (wave == 0 || bn <= 0) return;
    float* crossL = reinterpret_cast<float*>(smem + SM.cross_off);
    const int nchunk = (bn + 63) / 64, ntask = b * nchunk;
    constexpr int kD = 16;
    for (int t0 = (wave - 1) * kD; t0 < ntask; t0 += (kStepThreads / 64 - 1) * kD) {
        float v[kD];
#pragma unroll
        for (int u = 0; u < kD; ++u) {
            const int task = (t0 + u < ntask) ? t0 + u : ntask - 1;
            const int row = task / nchunk, c = (task - row * nchunk) * 64 + lane;
            v[u] = A.cross_next[(int64_t)row * bn + (c < bn ? c : 0)];
        }
#pragma unroll
        for (int u = 0; u < kD; ++u) {
            const int task = t0 + u;
            if (task < ntask) {
                const int row = task / nchunk, c = (task - row * nchunk) * 64 + lane;
                if (c < B) crossL[row * B + c] = v[u];
            }
        }
    }
}

// Linear copy global -> LDS with direct loads (global_load_lds_dwordx4: 1 KB per wave instruction), all 8 waves, rolled
// loop, no registers; nfloats a multiple of 256.  The caller waits (s_waitcnt vmcnt(0)) and synchronises.
__device__ __forceinline__ void dma_copy_to_lds(const float* __restrict__ src, float* lds_dst, int nfloats, int w0 = 0)
{
    typedef __attribute__((address_space(3))) void lds_void;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < w0) return;                                   // (waves w0..7 share the copy)
    for (int k = (wave - w0) * 256; k < nfloats; k += (kStepThreads / 64 - w0) * 256)
        __builtin_amdgcn_global_load_lds(src + k + lane * 4, (lds_void*)(lds_dst + k), 16, 0, 0);
}

// Stage the Gram rows of the candidate markers (cand[q] for marker c = tid + q*kStepThreads) in LDS.
// Stage the Gram rows of the candidate markers (cand[q] for marker c = tid + q*kStepThreads) in LDS.  Two steps, so that a
// caller can issue loads of its own between them that share the rows' memory latency (sampler_st.hpp: the cross-Gram rows of
// the compact chain): stage_assign (slots in marker order, candidate list; returns the number of rows that will be staged,
// total = all candidates) and stage_load.
__device__ __forceinline__ int stage_assign(char* smem, const StepSmem& SM, const SamplerArgs& A, const bool (&cand)[2], int& total)
{
    const int B = SM.B;
    short* slot_of = reinterpret_cast<short*>(smem + SM.slot_off);
    short* cand_list = reinterpret_cast<short*>(smem + SM.cand_off);
    int* wcnt = reinterpret_cast<int*>(smem + SM.wcnt_off);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    (void)A;
    int base = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (q * kStepThreads >= B) break;
        const int c = tid + q * kStepThreads;
        const unsigned long long m = __ballot(cand[q]);
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int pre = base, tot = base;
        for (int w = 0; w < kStepThreads / 64; ++w) { if (w < wave) pre += wcnt[w]; tot += wcnt[w]; }
        if (c < B) {
            int sl = -1;
            if (cand[q]) {
                sl = pre + __popcll(m & ((1ull << lane) - 1ull));
                if (sl < SM.max_cand) cand_list[sl] = (short)c; else sl = -1;
            }
            slot_of[c] = (short)sl;
        }
        base = tot;
        __syncthreads();
    }
    total = base;
    return base < SM.max_cand ? base : SM.max_cand;
}
// phase: 0 = issue + wait (the whole thing); 1 = issue only (full blocks: the direct loads; returns true -- otherwise nothing
// is done and false comes back: call phase 0 instead); 2 = wait for the direct loads of phase 1 and synchronise.
// keep_younger (phase 2): the caller issued exactly kStageYounger vector loads AFTER phase 1 on the waves != 0 and wants them
// to stay in flight -- the wait is for the direct loads only (vmcnt counts in order) and the barrier is LDS-only.
constexpr int kStageYounger = 19;
__device__ __forceinline__ bool stage_load(char* smem, const StepSmem& SM, const SamplerArgs& A, int ncand, long long* ts = nullptr,
                                           int phase = 0, bool keep_younger = false)
{
    const int B = SM.B;
    const short* cand_list = reinterpret_cast<const short*>(smem + SM.cand_off);
    float* rows = reinterpret_cast<float*>(smem + SM.rows_off);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = A.b;
    if (ts) ts[0] = clock64();
    // ALL row loads of the workgroup are issued before the first one is consumed: the fetch costs ONE memory latency
    // (microseconds under full-rate streaming), not one per batch.  Full blocks of 256 / 512 / 1024 markers (rows 16-byte
    // aligned): (row, 256-column) tasks, one float4 per lane, task = u*8 + wave.  The task -> (row, chunk) mapping uses
    // shifts only and the candidates' row indices are fetched from LDS in one batch first: measured, the address
    // arithmetic (a runtime division and a dependent LDS read per task, 24 tasks per wave whatever the count) cost more
    // than the memory latency itself -- 20 k of the 21 k cycles this function took per 512-marker block.
    const int b4 = A.b;
    const bool direct = b4 == B && (B == 256 || B == 512 || B == 1024);
    if (phase == 1 && !direct) return false;
    if (phase == 2) {
        if (ts) ts[1] = clock64();
        if (keep_younger && wave != 0) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");      // kStageYounger
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ts) ts[2] = clock64();
        if (keep_younger) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (__syncthreads would wait for every load)
        else __syncthreads();
        return true;
    }
    if (direct) {
        // Direct global -> LDS loads (global_load_lds_dwordx4: each lane's 16 bytes land at M0 + lane*16, i.e. one task =
        // 1 KB of a row straight into its slot): no staging registers, a ROLLED loop of a few instructions with every load
        // in flight, one wait at the end.  (The unrolled register version spent 14 k cycles per block just issuing: cold
        // straight-line code is fetched at memory latency.)
        typedef __attribute__((address_space(3))) void lds_void;
        const int sh = (B == 1024) ? 2 : (B == 512 ? 1 : 0);             // log2(256-column chunks per row)
        const int ntask = ncand << sh;
        // lane u of the wave holds the marker of its u-th task (task = wave + 8u)
        const int tmine = wave + 8 * lane;
        const int mycand = (int)cand_list[(tmine < ntask ? tmine : 0) >> sh];
        int u = 0;
        for (int task = wave; task < ntask; task += kStepThreads / 64, ++u) {
            const int crow = __builtin_amdgcn_readlane(mycand, u);
            const int ch = (task & ((1 << sh) - 1)) << 8;
            __builtin_amdgcn_global_load_lds(A.gram + (crow * B + ch + lane * 4),
                                             (lds_void*)(rows + ((task >> sh) * B + ch)), 16, 0, 0);
        }
        if (phase == 1) return true;
        if (ts) ts[1] = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ts) ts[2] = clock64();
    } else {
    // (row, 64-column chunk) tasks, kSL independent loads in flight per wave (ragged last block, 64/128-marker blocks)
    constexpr int kSL = 16;
    const int nchunk = B / 64, ntask = ncand * nchunk;
    for (int t0 = wave * kSL; t0 < ntask; t0 += (kStepThreads / 64) * kSL) {
        float v[kSL];
        int dst[kSL];
#pragma unroll
        for (int u = 0; u < kSL; ++u) {
            const int task = (t0 + u < ntask) ? t0 + u : ntask - 1;
            const int row = task / nchunk, c = (task - row * nchunk) * 64 + lane;
            v[u] = A.gram[(int64_t)cand_list[row] * b + (c < b ? c : 0)];
            dst[u] = row * B + c;
        }
#pragma unroll
        for (int u = 0; u < kSL; ++u) if (t0 + u < ntask) rows[dst[u]] = v[u];
    }
    }
    __syncthreads();
    return direct;
}
static_assert(kStageYounger == 19, "stage_load's s_waitcnt immediate");
__device__ __forceinline__ int stage_rows(char* smem, const StepSmem& SM, const SamplerArgs& A, const bool (&cand)[2], long long* ts = nullptr)
{
    int total = 0;
    const int ncand = stage_assign(smem, SM, A, cand, total);
    stage_load(smem, SM, A, ncand, ts);
    return ncand;
}

// acc[t] = fmaf(D[t][u], pq[u], acc[t]) for the 64 changes u of a section in marker order (D: LDS, trait stride dstride; pq: the
// thread's 64 Gram / cross-Gram values).  Batches of KB with the next batch's broadcast reads in flight behind the current
// one's multiply-adds -- pinned by a data dependence per batch: left alone the compiler hoists all 64 NT reads to the top (192
// registers on top of pq: spills); every batch boundary exposes one LDS latency, so few, big batches (16: 96 registers).
template <int NT, int KB = 8>
__device__ __forceinline__ void apply_section_changes(const float* D, int dstride, float (&acc)[NT], const float (&pq)[64])
{
    float dn[NT][KB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < KB; q += 4) {
                const float4 d0 = *reinterpret_cast<const float4*>(D + t * dstride + k0 + q);
                dn[t][q] = d0.x; dn[t][q + 1] = d0.y; dn[t][q + 2] = d0.z; dn[t][q + 3] = d0.w;
            }
    };
    fetch(0);
#pragma unroll
    for (int k0 = 0; k0 < 64; k0 += KB) {
        float dv[NT][KB];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int u = 0; u < KB; ++u) dv[t][u] = dn[t][u];
        if (k0 + KB < 64) fetch(k0 + KB);
#pragma unroll
        for (int u = 0; u < KB; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = fmaf(dv[t][u], pq[k0 + u], acc[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(acc[t]) :: "memory");
    }
}

// The HELPER workgroup of a dense launch (workgroup 16: an idle one on the sampler's XCD; sweep.hpp): the lookahead correction of
// the NEXT block,  corr[t][c'] = fmaf(D_e[t], C[e][c'], corr)  from 0 over the markers e of the block in marker order  (C =
// X_this'X_next), from the changes the sampler workgroup publishes section by section (A.xch: value + tag per 8-byte word) --
// the cross-Gram block (256 KB at 256 markers, 1 MB at 512) and most of the block's off-diagonal multiply-adds no longer go
// through the sampler's CU.  Thread c' < b_next owns column c'; the section's 64 cross-Gram values per thread are fetched a
// section ahead.  The wait is one-directional (the sampler workgroup waits for nobody and is dispatched before this one), its
// result is consumed by the NEXT launch.  Same operations in the same order as the in-workgroup form: bit-identical.
// A sampler that does not take the path that publishes (dense_big_st's vote failed) says so with the ABORT tag in section 0.
constexpr unsigned kXchAbort = 0x80000000u;
template <int NT>
__device__ __forceinline__ void corr_helper(char* smem, const SamplerArgs& A)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int bn = A.b_next, B = A.bsz, nsec = A.b >> 6;
    if (bn <= 0) return;                                                 // (the sweep's last block: no correction to form)
    float* dl = reinterpret_cast<float*>(smem);                          // [NT][64] the section's changes
    int* stop = reinterpret_cast<int*>(dl + NT * 64);
    const bool colthr = tid < bn;
    const int cn = colthr ? tid : 0;
    float corr[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) corr[t] = 0.f;
    float pq[64];
    auto load_c = [&](int s) {
        const char* base = reinterpret_cast<const char*>(A.cross_next + (int64_t)(64 * s) * bn);
        unsigned off = 4u * (unsigned)cn;
#pragma unroll
        for (int u = 0; u < 64; ++u) { pq[u] = *reinterpret_cast<const float*>(base + off); off += 4u * (unsigned)bn; asm volatile("" : "+v"(off)); }
    };
    if (tid == 0) *stop = 0;
    load_c(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll 1
    for (int s = 0; s < nsec; ++s) {
        if (tid < NT * 64) {
            // thread (t, lane): its value of the section, valid once the word carries this section's tag
            const int t = tid >> 6;
            const unsigned long long* src = A.xch + t * B + 64 * s + lane;
            const unsigned want = (unsigned)(A.xch_epoch + s + 1) & 0x7fffffffu;
            unsigned long long v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (((unsigned)(v >> 32) & 0x7fffffffu) != want) {
                __builtin_amdgcn_s_sleep(8);
                v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if ((unsigned)(v >> 32) & kXchAbort) *stop = 1;
            dl[tid] = __uint_as_float((unsigned)v);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // (LDS only: the cross-Gram loads stay in flight)
        if (*stop) return;                                                   // (the sampler forms the correction itself)
        if (colthr) apply_section_changes<NT>(dl, 64, corr, pq);
        if (s + 1 < nsec) load_c(s + 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // (dl is rewritten for the next section)
    }
    if (tid < B) {
#pragma unroll
        for (int t = 0; t < NT; ++t) A.corr_out[t * B + tid] = colthr ? corr[t] : 0.f;
    }
}

// rhs[t][:] += D[t] * G[ce][:]  for the committed marker ce (BayesABC.jl:169,172); one wave.
// c_from (a multiple of 64): only the columns from there on -- a single pass never reads the right-hand side of a marker in front of
// the committed one again, and the multi-trait skip-and-verify pass (sampler_role_mt) counts on their values staying what they
// were at those markers' own steps.
template <int NT>
__device__ __forceinline__ void apply_gram_row(char* smem, const StepSmem& SM, const SamplerArgs& A, int ce,
                                               const float (&D)[NT], int lane, int c_from = 0)
{
    const int B = SM.B, b = A.b;
    float* rhs_lds = reinterpret_cast<float*>(smem + SM.rhs_off);
    const short* slot_of = reinterpret_cast<const short*>(smem + SM.slot_off);
    const float* rows = reinterpret_cast<const float*>(smem + SM.rows_off);
    const int sl = __builtin_amdgcn_readfirstlane((int)slot_of[ce]);
    if (sl >= 0) {                                               // staged row: LDS only (explicit branch --
        for (int c2 = c_from + lane; c2 < B; c2 += 64) {         // a select would still issue the global load)
            const float g = rows[sl * B + c2];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (D[t] != 0.f) rhs_lds[t * B + c2] = fmaf(D[t], g, rhs_lds[t * B + c2]);
        }
    } else {
        const float* grow = A.gram + (int64_t)ce * b;             // symmetric: row = column
        for (int c2 = c_from + lane; c2 < B; c2 += 64) {
            const float g = grow[c2 < b ? c2 : 0];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (D[t] != 0.f) rhs_lds[t * B + c2] = fmaf(D[t], g, rhs_lds[t * B + c2]);
        }
        if (lane == 0) atomicAdd(&A.counters[1], 1ull);           // diagnostic: changes whose row was not staged
    }
}


}  // namespace jw
