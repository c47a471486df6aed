// sweep.hpp -- the fused per-block step of the marker sweep (gfx950, wave64).
//
// One sweep = nblocks + 1 launches of k_block_step, one per marker block, each holding TWO roles
// that run concurrently on different CUs (one-block lookahead, oracle: orc_*_lookahead_sweep):
//
//   launch k:  workgroup 0      = SAMPLER of block k-1
//              workgroups 1..   = UPDATE/PARTIAL of block k
//
//   UPDATE/PARTIAL (all other CUs, HBM-bound).  Workgroup (rg, g): row group rg = 8 wavefronts, each
//       owning a 256-row slice of the residual in registers; column group g = columns g, g+ncg, ...
//       (a) applies the net effect changes of block k-2   r += X[:,events] * d   (sparse exit update,
//           BayesABC.jl:181-185; sequential fmaf in marker order) -> residual r(k-2);
//       (b) streams the columns of block k ONCE from HBM and forms the partial block RHS
//           X_k[slice,:]' r(k-2)[slice]  (block_rhs!, tools4genotypes.jl:59-78): fp64-accumulated,
//           wave64 butterfly shuffle reduction, cross-wave combine in LDS, one partial per
//           (column, row group).
//   SAMPLER (one workgroup).  rhs_{k-1} = sum of row-group partials (computed by launch k-1 from the
//       residual r(k-3)), corrected for the changes of block k-2 with the cross-Gram X_{k-2}'X_{k-1}
//           rhs[c] = fmaf(d_j, C[j][c], rhs[c])       for the changed markers j, in marker order,
//       then ONE wavefront runs the exact single-site chain of the block (BayesABC.jl:153-179) by
//       speculative parallel evaluation: all lanes evaluate their marker against the current rhs; the
//       first lane whose effect changes commits; its Gram row corrects every rhs (BayesABC.jl:169,172);
//       the rest re-evaluate.  Lanes before the first change are final, so the result is the sequential
//       chain's.  Gram rows of the markers that look like changes at entry are staged in LDS by the
//       whole workgroup before the serial part, so a committed change costs an LDS read, not an HBM
//       round trip.
//
// Launch k depends only on launch k-1 (stream order).  The steady-state kernel has no in-kernel inter-workgroup
// communication; the COOP instantiation (dense sweeps) lets the column groups of a row group share the apply work through a
// bounded, fall-back-protected exchange (update_role.hpp).
//
// Files: update_role.hpp (UPDATE/PARTIAL), sampler_common.hpp, sampler_st.hpp (single trait, dense_big_st, Rule D),
// sampler_mt.hpp (multi-trait, Rule L); this file: the LDS carve, the step kernel, the independent-block kernels.
#pragma once
#include "kernels.hpp"

namespace jw {

constexpr int kStepThreads = 512;
constexpr int kRowsBytes = 128 * 1024;         // LDS budget for staged Gram rows (sampler role)
// Dynamic-LDS carve of one step workgroup (bytes); B = block size, NT = traits, (nd, nf) = doubles /
// floats of per-marker sampler constants staged for the serial wave (0 = none).
constexpr int kLdsBytes = 160 * 1024;
struct StepSmem {
    int B, NT, max_cand;
    int rhs_off, acur_off, astart_off, bcur_off, dcur_off, slot_off, cand_off, wcnt_off, lpr_off, log_off, prepd_off, prepf_off, rows_off, cross_off, bytes;
    bool has_cross;
    __host__ __device__ StepSmem(int B_, int NT_, int nd, int nf) : B(B_), NT(NT_)
    {
        rhs_off  = 0;                               // float [NT][B]  running block RHS
        acur_off = rhs_off + NT * B * 4;            // float [NT][B]  current alpha
        astart_off = acur_off + NT * B * 4;         // float [NT][B]  alpha at block entry
        bcur_off = astart_off + NT * B * 4;         // float [NT][B]  current beta   (multi-trait)
        dcur_off = bcur_off + NT * B * 4;           // float [NT][B]  current delta  (multi-trait)
        slot_off = dcur_off + NT * B * 4;           // int16 [B]      LDS slot of a marker's Gram row, -1 = not staged
        cand_off = slot_off + B * 2;                // int16 [B]
        wcnt_off = (cand_off + B * 2 + 15) / 16 * 16;          // int [16]
        lpr_off  = wcnt_off + 64;                   // double [16]    multi-trait: log prior probability of each of the 2^NT states
        log_off  = lpr_off + 128;                   // int2  [B]      committed changes of this block: {slot, bits(D)}
        prepd_off = log_off + B * 8;                // double [nd][B] per-marker constants (rep 0)
        prepf_off = prepd_off + nd * B * 8;         // float  [nf][B]
        rows_off = prepf_off + nf * B * 4;          // float [max_cand + 1][B] staged Gram rows + one overflow row
        int room = (kLdsBytes - 1024 - rows_off) / (4 * B) - 1;
        if (room > kRowsBytes / (4 * B)) room = kRowsBytes / (4 * B);
        max_cand = room < B ? room : B;
        if (max_cand < 1) max_cand = 1;
        // small blocks (B <= 128, all rows staged): the cross-Gram rows X_this'X_next of the block are copied to LDS too
        // (by the waves that idle during the serial phase), so the lookahead correction at the end reads LDS only
        cross_off = rows_off + (max_cand + 1) * B * 4;
        has_cross = (B <= 128) && (max_cand >= B) && (cross_off + B * B * 4 <= kLdsBytes - 1024);
        const int samp = cross_off + (has_cross ? B * B * 4 : 0);
        const int red = kRowGroupSlices * kColChunk * NT * 8;   // update role: double [8][64][NT]
        bytes = samp > red ? samp : red;
    }
};

// Multi-trait samplers park the per-marker draws of repetition 0 (NT thresholds + NT normals, fp64) and x'x in LDS
// when the block is small enough to leave room for the staged Gram rows; otherwise the serial wave reads them from HBM.
// (round 6: up to 3072 = 1024 markers x 3 traits -- four Gram rows still fit beside them, which is what the SPARSE steady state of a
// multi-trait chain needs: half the launches and fronts of 512-marker blocks)
__host__ __device__ constexpr int mt_park_nd(int B, int NT) { return (B * NT <= 3072) ? 2 * NT : 0; }
__host__ __device__ constexpr int mt_park_nf(int B, int NT) { return (B * NT <= 3072) ? 1 + NT : 0; }   // x'x, log C11 per trait

}  // namespace jw

#include "update_role.hpp"
#include "sampler_common.hpp"
#include "sampler_st.hpp"
#include "sampler_mt.hpp"

namespace jw {

// ---------------------------------------------------------------------------------------------
// The fused step.  grid = 1 + nrg*ncg, block = 512.
// ---------------------------------------------------------------------------------------------
struct UpdateArgs {
    const float* r_in; float* r_out;
    const Events* ev;             // changes to apply (block k-2)
    int64_t j0; int b;            // block whose partial RHS is formed (b = 0: none)
    int nslices, nrg, ncg;
    int spg;                      // slices (waves that stream) per row group: <= 8
    double* partials; int bstride;
    int quiet_xcd;                // 1: ids = 0 mod 8 (the sampler's XCD) do no update work
    unsigned long long* dbg;      // phase cycle counters (diagnostics) or NULL
    int* sync_now; int* sync_next; // [nrg] arrival counters of the cooperative dense apply: this launch's / the next one's (or NULL)
    int dbg_throttle;             // > 1: only every n-th update workgroup runs (timing experiments; results wrong)
};
template <class CX>
struct UpdateArgsT : UpdateArgs {
    CX cx;                        // genotype storage accessor
};

// COOP: the update role may split a dense change list among the column groups of a row group (see update_role); a separate
// instantiation so that the steady-state kernel's code is exactly the one without it (its presence alone cost 1 %).
template <int METHOD, int NT, class CX, bool COOP = false, bool DENSE = false>
__global__ __launch_bounds__(kStepThreads) void k_block_step(UpdateArgsT<CX> U, SamplerArgs S, int do_sample)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (blockIdx.x == 0) {
        if (!do_sample) return;
#ifdef JWAS_HIP_POISON_LDS
        // development builds (-DJWAS_HIP_POISON_LDS=0x11111111): the sampler starts from a known-bad LDS image instead of
        // whatever the previous kernel left there -- a read of something this launch did not write shows up as a parity failure
        for (int i = threadIdx.x; i < 150000 / 4; i += kStepThreads) reinterpret_cast<unsigned*>(smem)[i] = (unsigned)(JWAS_HIP_POISON_LDS);
        __syncthreads();
#endif
        if constexpr (is_mt_method(METHOD)) sampler_role_mt<METHOD, NT, DENSE>(smem, S);
        else sampler_role_st<METHOD, DENSE>(smem, S);
        return;
    }
    int w = blockIdx.x - 1;
    if (U.quiet_xcd) {
        // Speed heuristic only (never correctness): workgroup ids are observed to round-robin over the 8 XCDs,
        // so ids = 0 mod 8 share the sampler's XCD/L2.  Leaving them idle keeps that L2 free of the streaming
        // traffic, which shortens every dependent load of the sampler chain.
        if constexpr (DENSE) {
            // ... and ONE of them (id 8) pulls what the NEXT launch's sampler will read -- the next block's Gram and its
            // cross-Gram, 2 MB at 512 markers -- into that L2: dense_big_st moves ~1.5 MB per block through one CU, and with
            // HBM latency under the update role's streaming (~5 us) a CU's 63 loads in flight per wave cap it at ~25 GB/s.
            if (blockIdx.x == 8 && do_sample && (S.bsz == 256 || S.bsz == 512) && !S.dense_big_off) {
                const int nl_g = (S.gram_next != nullptr) ? (S.b_next * S.b_next + 31) / 32 : 0;
                const int nl_c = (S.cross_after != nullptr) ? S.lines_after : 0;
                const int nl_t = (S.tsec_next != nullptr) ? S.tsec_lines : 0;          // Rule T: the next block's section inverses
                float sink = 0.f;
                if constexpr (is_mt_method(METHOD)) {
                    // ... and, first, the per-marker state the next launch's FRONT reads (x'x, alpha / beta / delta, the sweep's draws
                    // and logs: ~30 KB per 256 markers): the front is ONE memory latency -- 3.5 us from HBM under the stream's load
                    const int bnx = S.b_next;
                    const int64_t jn = S.j0 + S.b, pp = S.p;
                    if (bnx > 0) {
                        const int nlf = (bnx + 31) / 32, nld = (bnx + 15) / 16, nfa = 1 + 4 * NT;
                        int task = (int)threadIdx.x;
                        if (task < nfa * nlf) {
                            const int ar = task / nlf, l = task - ar * nlf;
                            const float* base = ar == 0 ? S.xpx
                                              : ar <= NT ? S.alpha + (int64_t)(ar - 1) * pp
                                              : ar <= 2 * NT ? S.beta + (int64_t)(ar - NT - 1) * pp
                                              : ar <= 3 * NT ? reinterpret_cast<const float*>(S.delta) + (int64_t)(ar - 2 * NT - 1) * pp
                                              : S.prep_f + (int64_t)(ar - 3 * NT - 1) * pp;
                            sink += base[jn + (int64_t)l * 32];
                        }
                        task -= nfa * nlf;
                        if (task >= 0 && task < 2 * NT * nld) {
                            const int ar = task / nld, l = task - ar * nld;
                            sink += (float)S.prep_d[(int64_t)ar * pp + jn + (int64_t)l * 16];
                        }
                    }
                }
                for (int l0 = 0; l0 < nl_g + nl_c + nl_t; l0 += 8 * kStepThreads) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        int l = l0 + u * kStepThreads + (int)threadIdx.x;
                        l = l < nl_g + nl_c + nl_t ? l : nl_g + nl_c + nl_t - 1;
                        v[u] = (l < nl_g) ? S.gram_next[(int64_t)l * 32] : (l < nl_g + nl_c ? S.cross_after[(int64_t)(l - nl_g) * 32] : S.tsec_next[(int64_t)(l - nl_g - nl_c) * 32]);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) sink += v[u];
                }
                asm volatile("" ::"v"(sink));
                return;
            }
        }
        if constexpr (DENSE) {
            // launches whose sampler publishes its sections' changes (S.xch: multi-trait Rule T, single-trait dense_big_st): workgroup
            // 16 -- another idle one on the sampler's XCD -- forms the next block's lookahead correction from them (corr_helper,
            // sampler_common.hpp; the launcher makes sure the grid holds it)
            if (blockIdx.x == 16 && do_sample && S.xch != nullptr) { corr_helper<NT>(smem, S); return; }
        }
        if ((blockIdx.x & 7) == 0) return;
        w = (int)(blockIdx.x - 1) - (int)((blockIdx.x - 1) >> 3);
    }
    if (w >= U.nrg * U.ncg) return;
    if (U.dbg_throttle > 1 && (w % U.dbg_throttle) != 0) return;       // timing experiments only
    constexpr bool kRoll = (METHOD == kBayesC || METHOD == kBayesB) && NT == 1 && !DENSE && CX::kDepth == 1;      // (see update_role; dense storage: the packed stream keeps 8 batches in registers)
    update_role<NT, CX, COOP, kRoll>(smem, w % U.nrg, w / U.nrg, U.cx, U.r_in, U.r_out, EvPlain{U.ev}, U.j0, U.b,
                                            U.nslices, U.nrg, U.ncg, U.partials, U.bstride, U.spg, U.sync_now, U.sync_next, U.dbg);
}

// ---------------------------------------------------------------------------------------------
// GROUPED LAUNCHES (round 5; single trait, single-pass sweeps with a sparse prior in their steady state).  A launch of
// k_block_step costs ~3.5 us that are not bandwidth (dispatch gap, ramp-up, drain: NOTES R1.1) -- 11 % of a 1024-marker
// launch.  k_group_step streams a GROUP of m = 2 or 4 consecutive blocks per launch and its sampler workgroup samples the m
// blocks of the previous group one after the other (the same sampler_role_st, the same code warm in the instruction cache):
//
//   launch K:  workgroup 0      = SAMPLER of blocks m(K-1) .. m(K-1)+m-1, in order
//              workgroups 1..   = UPDATE/PARTIAL: apply the changes of group K-2 (ONE merged list), stream group K
//
// Same chain, another lookahead: the partial sums of ALL blocks of group K were formed from the residual that holds the changes
// up to group K-2, so block s of the group is corrected for (oracle: orc_set_lookahead_group, la_group_*)
//   cG  the changes of the WHOLE previous group          (group_corr at the end of the previous launch, cross-Gram group -> group)
//   cW  s odd: the changes of block s-1, its pair's first (the sampler's own lookahead correction: corr_out -> corr_in, as ever)
//   cP  m = 4, s >= 2: the changes of blocks 0 and 1     (group_corr after block 1, cross-Gram pair -> pair)
// each a fused-multiply-add chain from 0 over the changed markers in marker order, and  rhs = fl32(sum of partials) + ((cW + cG) + cP)
// (absent terms are +0).  A hierarchy: block -> block inside a pair, pair -> pair inside a four, group -> group.
// ---------------------------------------------------------------------------------------------
struct GroupArgs {
    int ns;                        // blocks of the SAMPLED group (0 = none: first launch of a sweep)
    int m;                         // blocks per group: 2 or 4
    int64_t j0;                    // first marker of the sampled group (rows of the cross-Grams below)
    const float* cross_grp;        // X_group' X_nextgroup (row stride bn_grp), or NULL: the sampled group is the last one
    int bn_grp;
    const float* cross_pair;       // m = 4: X_{blocks 0,1}' X_{blocks 2,3} of the sampled group (row stride bn_pair), or NULL
    int bn_pair;
    float* corr_g_out;             // [m * bs] cG of the next group
    float* corr_p;                 // [2 * bs] cP
    const int32_t* ev_idx; const float* ev_delta;     // merged change list of the sampled group (capacity m * bs; header: the blocks' ev_out)
    int pp;                        // PING-PONG samplers: block s in workgroup 8 s (SamplerArgs::pp_*; the launcher keeps workgroup ids = 0 mod 8 free
                                   // of update work and the grid large enough); relay buffers of tagged words:
    unsigned long long* pp_ph;     // [2 bs] block 0's part of the cP chain
    unsigned long long* pp_cp;     // [2 bs] cP (read by blocks 2 and 3)
    unsigned long long* pp_h;      // [3][m bs] the cG chain after blocks 0, 1, 2
    int pp_kernel;                 // launch k_group_step<., ., PP = true> (the host's choice per sweep: ping-pong samplers and / or cooperative apply wanted)
};
// The sampler arguments of the group's blocks, one full set per block, filled by the host (sweep_enqueue): block s reads ITS set
// from the kernel-argument segment -- a copy of one set edited per block inside the kernel kept ~80 scalars live across the whole
// sampler (492 scalar and 277 vector registers spilled).
struct GroupSamplers { SamplerArgs a[4]; };

// corr[c] = fmaf(d_e, C[row_e][c], corr[c]) over list entries [e_lo, e_hi) in list (= marker) order, for ncols_out columns (columns
// >= bn: 0), starting from 0 -- or, seed != NULL, from the chain another workgroup formed over the entries before e_lo and posted as
// tagged words (ping-pong samplers: block 0's workgroup runs ITS part of the next group's correction while block 1 is still being
// walked; continuing a fused-multiply-add chain where it stopped gives the bits of the one chain).  post != NULL: the result goes
// out as tagged words instead of to `out`.  All threads; the list is staged through LDS in chunks of 512 entries; NQ column slots
// per thread (ncols_out <= NQ * 512), 64 / NQ rows = 64 loads in flight per thread and pass: a pass is one memory round trip, and
// with 30-60 changes per group (BayesR, a fixed pi) four rows per pass were 15 dependent round trips at the end of every launch.
// coherent: part of the list was written by ANOTHER workgroup of this launch (write-through stores, acknowledged before the count
// was handed over) -- read it at the coherence point, not through this XCD's L2.
template <int NQ>
__device__ __attribute__((noinline)) void group_corr_n(char* smem, const int32_t* eidx, const float* edel, int e_lo, int e_hi, int64_t jrow0,
                                                       const float* __restrict__ cross, int bn, float* __restrict__ out, int ncols_out, bool coherent,
                                                       const unsigned long long* seed, unsigned long long* post, unsigned tag, unsigned long long* counters)
{
    constexpr int R = NQ >= 8 ? 4 : 64 / NQ;          // (8 column slots = 4 x 1024-marker launches: a handful of changes per group; the pass of round 5)
    int* lrow = reinterpret_cast<int*>(smem);
    float* ld = reinterpret_cast<float*>(smem) + kStepThreads;
    const int tid = threadIdx.x;
    float corr[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = tid + q * kStepThreads;
        corr[q] = (seed != nullptr && c < ncols_out) ? __uint_as_float(pp_wait_word(seed + c, tag, counters)) : 0.f;
    }
    for (int e0 = e_lo; e0 < e_hi; e0 += kStepThreads) {
        const int nc = (e_hi - e0) < kStepThreads ? (e_hi - e0) : kStepThreads;
        __syncthreads();
        if (tid < nc) {
            if (coherent) {
                lrow[tid] = (int)((int64_t)__hip_atomic_load(eidx + e0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - jrow0);
                ld[tid] = __hip_atomic_load(edel + e0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else { lrow[tid] = (int)((int64_t)eidx[e0 + tid] - jrow0); ld[tid] = edel[e0 + tid]; }
        }
        __syncthreads();
        for (int h = 0; h < nc; h += R) {
            float g[R][NQ];
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int64_t row = lrow[h + u < nc ? h + u : nc - 1];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int c = tid + q * kStepThreads;
                    g[u][q] = cross[row * bn + (c < bn ? c : 0)];
                }
            }
#pragma unroll
            for (int u = 0; u < R; ++u) {
                if (h + u < nc) {
                    const float d = ld[h + u];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) corr[q] = fmaf(d, g[u][q], corr[q]);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = tid + q * kStepThreads;
        if (c < ncols_out) {
            const float v = (c < bn) ? corr[q] : 0.f;
            if (post != nullptr) pp_post_word(post + c, tag, __float_as_uint(v)); else out[c] = v;
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void group_corr(char* smem, const int32_t* eidx, const float* edel, int e_lo, int e_hi, int64_t jrow0,
                                           const float* __restrict__ cross, int bn, float* __restrict__ out, int ncols_out, bool coherent = false,
                                           const unsigned long long* seed = nullptr, unsigned long long* post = nullptr, unsigned tag = 0u,
                                           unsigned long long* counters = nullptr)
{
    if (ncols_out <= kStepThreads) group_corr_n<1>(smem, eidx, edel, e_lo, e_hi, jrow0, cross, bn, out, ncols_out, coherent, seed, post, tag, counters);
    else if (ncols_out <= 2 * kStepThreads) group_corr_n<2>(smem, eidx, edel, e_lo, e_hi, jrow0, cross, bn, out, ncols_out, coherent, seed, post, tag, counters);
    else if (ncols_out <= 4 * kStepThreads) group_corr_n<4>(smem, eidx, edel, e_lo, e_hi, jrow0, cross, bn, out, ncols_out, coherent, seed, post, tag, counters);
    else group_corr_n<8>(smem, eidx, edel, e_lo, e_hi, jrow0, cross, bn, out, ncols_out, coherent, seed, post, tag, counters);
}

// PP: the instantiation of the high-turnover sweeps -- ping-pong samplers (G.pp) and the cooperative apply of the merged list
// (U.sync_now); the steady-state sweeps launch the one without either (neither code path exists in it).
// The steady-state kernel's form: the whole list [0, ne) of ONE workgroup, from 0, to `out` -- no seed, no tagged words, no coherent
// reads (a handful of changes per group: four rows x NQ column slots per pass).
template <int NQ>
__device__ __attribute__((noinline)) void group_corr_plain_n(char* smem, const int32_t* __restrict__ eidx, const float* __restrict__ edel, int ne, int64_t jrow0,
                                                             const float* __restrict__ cross, int bn, float* __restrict__ out, int ncols_out)
{
    int* lrow = reinterpret_cast<int*>(smem);
    float* ld = reinterpret_cast<float*>(smem) + kStepThreads;
    const int tid = threadIdx.x;
    float corr[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) corr[q] = 0.f;
    for (int e0 = 0; e0 < ne; e0 += kStepThreads) {
        const int nc = (ne - e0) < kStepThreads ? (ne - e0) : kStepThreads;
        __syncthreads();
        if (tid < nc) { lrow[tid] = (int)((int64_t)eidx[e0 + tid] - jrow0); ld[tid] = edel[e0 + tid]; }
        __syncthreads();
        for (int h = 0; h < nc; h += 4) {
            float g[4][NQ];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t row = lrow[h + u < nc ? h + u : nc - 1];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int c = tid + q * kStepThreads;
                    g[u][q] = cross[row * bn + (c < bn ? c : 0)];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (h + u < nc) {
                    const float d = ld[h + u];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) corr[q] = fmaf(d, g[u][q], corr[q]);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = tid + q * kStepThreads;
        if (c < ncols_out) out[c] = (c < bn) ? corr[q] : 0.f;
    }
    __syncthreads();
}
__device__ __forceinline__ void group_corr_plain(char* smem, const int32_t* eidx, const float* edel, int ne, int64_t jrow0,
                                                 const float* cross, int bn, float* out, int ncols_out)
{
    if (ncols_out <= 4 * kStepThreads) group_corr_plain_n<4>(smem, eidx, edel, ne, jrow0, cross, bn, out, ncols_out);
    else group_corr_plain_n<8>(smem, eidx, edel, ne, jrow0, cross, bn, out, ncols_out);
}

template <int METHOD, class CX, bool PP = false>
__global__ __launch_bounds__(kStepThreads) void k_group_step(UpdateArgsT<CX> U, const int32_t* uev_idx, const float* uev_delta,
                                                             GroupSamplers SS, GroupArgs G)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ping-pong (G.pp): block s of the sampled group has its own workgroup, id 8 s -- idle on the sampler's XCD --, so that every
    // block's front runs at launch start and only the chain itself is sequential (sampler_role_st, SamplerArgs::pp_*); the
    // corrections for the blocks behind (cP inside a four, cG for the next group) are RELAYED: each workgroup continues the
    // fused-multiply-add chain over its own block's changes from where the workgroup before it stopped (group_corr: seed / post)
    if constexpr (!PP) {
        if (blockIdx.x == 0) {
            if (G.ns <= 0) return;
            int nev = 0;
#pragma unroll 1
            for (int s = 0; s < G.ns; ++s) {
                nev = sampler_role_st<METHOD, false, true, false>(smem, SS.a[s], nev);      // (-> the length of the merged list behind block s)
                __syncthreads();                                // the block's global stores (cW, the list) are visible to the workgroup
                if (s == 1 && G.cross_pair != nullptr && G.ns > 2)
                    group_corr_plain(smem, G.ev_idx, G.ev_delta, nev, G.j0, G.cross_pair, G.bn_pair, G.corr_p, 2 * SS.a[0].bsz);
            }
            if (G.cross_grp != nullptr) group_corr_plain(smem, G.ev_idx, G.ev_delta, nev, G.j0, G.cross_grp, G.bn_grp, G.corr_g_out, G.m * SS.a[0].bsz);
            return;
        }
    }
    const int pp_s = (PP && G.pp != 0 && (blockIdx.x & 7u) == 0u && blockIdx.x < 32u) ? (int)(blockIdx.x >> 3) : -1;
    if (PP && (blockIdx.x == 0 || pp_s > 0)) {
        if (G.ns <= 0 || pp_s >= G.ns) return;
        const int s_lo = pp_s >= 0 ? pp_s : 0, s_hi = pp_s >= 0 ? pp_s + 1 : G.ns;
        int nev = 0;
        const long long tg0 = clock64();
        long long tgp = 0;
#pragma unroll 1
        for (int s = s_lo; s < s_hi; ++s) {
            nev = sampler_role_st<METHOD, false, true, PP>(smem, SS.a[s], nev);      // (-> the length of the merged list behind block s)
            __syncthreads();                                    // the block's global stores (cW, the list) are visible to the workgroup
            if (pp_s < 0 && s == 1 && G.cross_pair != nullptr && G.ns > 2) {
                const long long t0 = clock64();
                group_corr(smem, G.ev_idx, G.ev_delta, 0, nev, G.j0, G.cross_pair, G.bn_pair, G.corr_p, 2 * SS.a[0].bsz);
                tgp = clock64() - t0;
            }
        }
        const long long tg1 = clock64();
        const int bsz = SS.a[0].bsz, ncols = G.m * bsz;
        if (PP && pp_s >= 0) {
            const SamplerArgs& A = SS.a[pp_s];
            const int e_lo = pp_count_in(A);                    // this block's entries of the merged list: [e_lo, nev)
            const bool mine_wt = A.pp_cnt_out != nullptr;       // (written through: read them back at the coherence point)
            // cP (4 blocks per launch): the chain over blocks 0 and 1 for the columns of blocks 2 and 3 -- block 0's part, then block 1's
            if (G.cross_pair != nullptr && G.ns > 2 && pp_s < 2)
                group_corr(smem, G.ev_idx, G.ev_delta, e_lo, nev, G.j0, G.cross_pair, G.bn_pair, nullptr, 2 * bsz, mine_wt,
                           pp_s == 0 ? nullptr : G.pp_ph, pp_s == 0 ? G.pp_ph : G.pp_cp, A.pp_tag, A.counters);
            // cG of the next group: block s continues the chain of blocks 0 .. s-1; the last block stores the result
            if (G.cross_grp != nullptr) {
                const bool last = pp_s == G.ns - 1;
                group_corr(smem, G.ev_idx, G.ev_delta, e_lo, nev, G.j0, G.cross_grp, G.bn_grp, last ? G.corr_g_out : nullptr, ncols, mine_wt,
                           pp_s == 0 ? nullptr : G.pp_h + (size_t)(pp_s - 1) * ncols, last ? nullptr : G.pp_h + (size_t)pp_s * ncols, A.pp_tag, A.counters);
            }
        } else if (G.cross_grp != nullptr) group_corr(smem, G.ev_idx, G.ev_delta, 0, nev, G.j0, G.cross_grp, G.bn_grp, G.corr_g_out, ncols);
        if (threadIdx.x == 0) {                                 // (diagnostics: cP pass | the corrections behind the last block | the whole workgroup)
            const long long tg2 = clock64();
            atomicAdd(&SS.a[0].counters[25], (unsigned long long)tgp);
            atomicAdd(&SS.a[0].counters[26], (unsigned long long)(tg2 - tg1));
            atomicAdd(&SS.a[0].counters[27], (unsigned long long)(tg2 - tg0));
            if (pp_s == G.ns - 1) atomicAdd(&SS.a[0].counters[28], (unsigned long long)(tg2 - tg0));      // ping-pong: the LAST block's workgroup, start to end
        }
        return;
    }
    int w = blockIdx.x - 1;
    if (U.quiet_xcd) {                                          // (see k_block_step)
        if ((blockIdx.x & 7) == 0) return;
        w = (int)(blockIdx.x - 1) - (int)((blockIdx.x - 1) >> 3);
    }
    if (w >= U.nrg * U.ncg) return;
    // (no rolling-window apply here: grouped launches run the sparse steady state -- a handful of changes per group -- and the
    // window's 32 columns in registers are what pushed this kernel, with its sampler loop, into scratch memory)
    // (the cooperative apply -- the column groups of a row group share the rows of the merged list's columns instead of every one of
    // them re-reading all of them -- when the host passes the arrival counters: the high-turnover sweeps, 60-120 changes per launch)
    update_role<1, CX, PP, false, EvGroup>(smem, w % U.nrg, w / U.nrg, U.cx, U.r_in, U.r_out, EvGroup{U.ev, uev_idx, uev_delta}, U.j0, U.b,
                                             U.nslices, U.nrg, U.ncg, U.partials, U.bstride, U.spg, PP ? U.sync_now : nullptr, PP ? U.sync_next : nullptr, U.dbg);
}

// Cross-Gram of consecutive blocks, exact (fp64-accumulated): C[a][c] = x_{jp+a}' x_{j0+c}.
// grid = (bsize, nblocks-1), block = 256; workgroup (a, i) writes row a of cross block i+1.
template <class CX>
__global__ __launch_bounds__(256) void k_cross_f64(CX cx, int64_t p, int bsize,
                                                   float* __restrict__ cross, const int64_t* __restrict__ starts = nullptr, int odd_only = 0)
{
    const int64_t ld = cx.ld;
    const int64_t blk = odd_only ? 2 * (int64_t)blockIdx.y + 1 : (int64_t)blockIdx.y + 1;
    const int64_t j0 = starts ? starts[blk] : blk * bsize, jp = starts ? starts[blk - 1] : j0 - bsize;
    const int b = starts ? (int)(starts[blk + 1] - j0) : (int)((j0 + bsize <= p) ? bsize : (p - j0));
    const int a = blockIdx.x;
    if (a >= (int)(j0 - jp)) return;                   // (explicit starts: the previous block may be shorter than bsize)
    float* C = cross + (odd_only ? (blk >> 1) : blk) * (int64_t)bsize * bsize;      // (odd blocks only: stored compactly, block 2q + 1 at q)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < b; c += 4) {
        double s = 0.0;
        for (int64_t i = (int64_t)lane * 4; i < ld; i += 256) {
            const float4 qa = cx.load4(jp + a, i);
            const float4 qc = cx.load4(j0 + c, i);
            const float4 wv = *reinterpret_cast<const float4*>(cx.w + i);
            s = fma((double)qa.x, (double)(qc.x * wv.x), s);
            s = fma((double)qa.y, (double)(qc.y * wv.y), s);
            s = fma((double)qa.z, (double)(qc.z * wv.z), s);
            s = fma((double)qa.w, (double)(qc.w * wv.w), s);
        }
        s = wave_sum(s);
        if (lane == 0) C[(int64_t)a * b + c] = (float)s;
    }
}

// ---------------------------------------------------------------------------------------------
// INDEPENDENT-BLOCK mode (BayesABC_block_independent!, BayesABC.jl:190-255; BayesR.jl:195-273;
// MTBayesABC.jl:335-440): every block's RHS comes from the SAME residual snapshot, the blocks are
// sampled independently (here: all at once, one workgroup each), and the residual is reconciled
// afterwards with r += sum_b X_b * (alpha_old_b - alpha_new_b) in (block, marker) order.
// ---------------------------------------------------------------------------------------------
// Block RHS of ALL blocks from the snapshot.  grid = (nrg*ncg, nblocks), block = 512.
template <int NT, class CX>
__global__ __launch_bounds__(kStepThreads) void k_indep_rhs(UpdateArgsT<CX> U, int64_t p, int bsz, int64_t pstride,
                                                            const int64_t* __restrict__ starts /* explicit partition (nblocks + 1), or NULL */)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t blk = blockIdx.y;
    const int64_t j0 = starts ? starts[blk] : blk * bsz;
    const int b = starts ? (int)(starts[blk + 1] - j0) : (int)((j0 + bsz <= p) ? bsz : p - j0);
    const int ncg = U.ncg < b ? U.ncg : b;
    const int w = blockIdx.x;
    if (w >= U.nrg * ncg) return;
    update_role<NT, CX>(smem, w % U.nrg, w / U.nrg, U.cx, U.r_in, nullptr, EvPlain{U.ev}, j0, b,
                        U.nslices, U.nrg, ncg, U.partials + blk * pstride, U.bstride, U.spg);
}

// All blocks sampled concurrently.  grid = nblocks, block = 512, dynamic LDS as k_block_step.
template <int METHOD, int NT, bool DENSE = false>
__global__ __launch_bounds__(kStepThreads) void k_indep_sample(SamplerArgs S, int64_t pstride, Events* ev_all,
                                                               const int64_t* __restrict__ starts)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t blk = blockIdx.x;
    S.j0 = starts ? starts[blk] : blk * S.bsz;
    S.b = starts ? (int)(starts[blk + 1] - S.j0) : (int)((S.j0 + S.bsz <= S.p) ? S.bsz : S.p - S.j0);
    S.partials += blk * pstride;
    S.gram += blk * (int64_t)S.bsz * S.bsz;
    S.ev_out = ev_all + blk;
    S.b_next = 0;                                   // no lookahead correction in this mode
    if constexpr (is_mt_method(METHOD)) sampler_role_mt<METHOD, NT>(smem, S);
    else sampler_role_st<METHOD, DENSE>(smem, S);
}

// Exclusive scan of the per-block change counts.  grid = 1, block = 1024.
JW_PLAIN_KERNEL __global__ __launch_bounds__(1024) void k_indep_scan(const Events* __restrict__ ev_all, int nblocks,
                                                     int32_t* __restrict__ offs, int32_t* __restrict__ total)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + tid;
        const int v = i < nblocks ? ev_all[i].count : 0;
        int x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int wpre = 0, tot = 0;
        for (int q = 0; q < 16; ++q) { if (q < wave) wpre += wsum[q]; tot += wsum[q]; }
        if (i < nblocks) offs[i] = carry + wpre + x - v;
        __syncthreads();
        if (tid == 0) carry += tot;
        __syncthreads();
    }
    if (tid == 0) *total = carry;
}

// Compact the per-block change lists into one list in (block, marker) order.  grid = nblocks, block = 256.
JW_PLAIN_KERNEL __global__ __launch_bounds__(256) void k_indep_gather(const Events* __restrict__ ev_all, const int32_t* __restrict__ offs,
                                                      int nt, int32_t* __restrict__ idx_all, float* __restrict__ delta_all,
                                                      int64_t dstride)
{
    const Events* ev = ev_all + blockIdx.x;
    const int ne = ev->count, off = offs[blockIdx.x];
    for (int e = threadIdx.x; e < ne; e += 256) {
        idx_all[off + e] = ev->idx[e];
        for (int t = 0; t < nt; ++t) delta_all[t * dstride + off + e] = ev->delta[t][e];
    }
}

}  // namespace jw
