// sampler_st.hpp -- SAMPLER role, single trait (BayesA/B/C, BayesR): dense_section / dense_big_st (dense priors, Rule D) and sampler_role_st.
// Included by sweep.hpp.
#pragma once
#include "kernels.hpp"

namespace jw {

// ---------------------------------------------------------------------------------------------
// SAMPLER role, single trait.  METHOD in {kBayesC, kBayesB, kBayesR}.
//
// Per-marker constants parked in LDS for the serial wave (rep 0):
//   BayesA/B/C: doubles [zs]                      floats [1/lhs, beta_excl, x'x, lo, hi]   (lo/hi: AbcMarker::thresholds)
//   BayesR    : doubles [1/lhs_k, zs_k, T_k] (9)  floats [x'x, candidate threshold]
// ---------------------------------------------------------------------------------------------
// ---- DENSE blocks: one 64-marker section of the in-lane walk (see sampler_role_st).  Lane l owns marker l of the section
// (Q = 0: running rhs r0; Q = 1: r1); at step l lane l's alpha_old - alpha_new is broadcast with one v_readlane and applied
// to the running rhs of the section's own markers (Q = 0) and of the next section's (TWO) with the marker's Gram row
// (grow: LDS, row stride B; read a batch of eight rows ahead).  rev = the rhs the lane's own marker was evaluated
// against.  ALLINC: every marker is included whatever its rhs (no compare / select on the chain).  No branch inside
// a batch; the dependent chain per step is add, mul, mul, cvt, add(f64), cvt, sub, readlane, fma.
__device__ __forceinline__ float dense_alpha_new(float x, float da, float ie, float invLhs, double zs, bool incl)
{
    const float rhs  = (x + da) * ie;                                       // BayesABC.jl:36  (da = d * alpha_old)
    const float gHat = rhs * invLhs;                                        // :39
    return incl ? (float)((double)gHat + zs) : 0.f;                         // :46 / :55
}
// RULED: the sweep runs under Rule D (uniform pi = 0): alpha_new = fmaf(kc1, x, kc0) -- the chain is fma, sub, readlane, fma.
template <int Q, bool TWO, bool ALLINC, bool RULED = false>
__device__ __forceinline__ void dense_section(const float* grow, int B, int nsteps, int lane, float ie, float lo, float hi,
                                              float il, float da, float ao, double zs, float& r0, float& r1, float& rev,
                                              float kc1 = 0.f, float kc0 = 0.f)
{
    auto step = [&](int l, float c0, float c1) {
        const float x = (Q == 0) ? r0 : r1;
        const float an = RULED ? fmaf(kc1, x, kc0) : dense_alpha_new(x, da, ie, il, zs, ALLINC ? true : abc_included(x, lo, hi));
        const float Dl = ao - an;                                           // (excluded: alpha_old - 0)
        const float D = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Dl), l));
        if (Q == 0) r0 = fmaf(D, c0, r0);                                   // D = 0: exact no-op
        if (TWO) r1 = fmaf(D, c1, r1);
    };
    constexpr int kBatch = 8;
    float n0[kBatch], n1[kBatch];
    auto load = [&](int l0) {
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            n0[u] = (Q == 0) ? grow[(l0 + u) * B + lane] : 0.f;
            n1[u] = TWO ? grow[(l0 + u) * B + 64 + lane] : 0.f;
        }
    };
    int l = 0;
    if (nsteps >= kBatch) load(0);
#pragma unroll 1
    for (; l + kBatch <= nsteps; l += kBatch) {
        float c0[kBatch], c1[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) { c0[u] = n0[u]; c1[u] = n1[u]; }
        if (l + 2 * kBatch <= nsteps) load(l + kBatch);                    // the next batch's rows: in flight during this one
#pragma unroll
        for (int u = 0; u < kBatch; ++u) step(l + u, c0[u], c1[u]);
    }
#pragma unroll 1
    for (; l < nsteps; ++l) step(l, (Q == 0) ? grow[l * B + lane] : 0.f, TWO ? grow[l * B + 64 + lane] : 0.f);
    rev = (Q == 0) ? r0 : r1;       // strictly upper diagonal tile: the lane's rhs has not moved since its own step
}

// The dense walks run on STRICTLY UPPER diagonal Gram tiles: G[l][c] = 0 for c <= l inside a 64-marker section, so that a
// lane's running rhs stops moving at its own step (the later steps' updates are exact no-ops on it) and the walk needs no
// per-step "remember what I was evaluated with" (a compare and a select of the ~10 issue slots of a step -- a lone wave
// issues one instruction every ~8 cycles, scripts/micro/dep_latency.hip, so the walk is bound by its instruction COUNT).
// One 64 x 64 tile with leading dimension ld, by the calling threads (nthr of them, thread index t).
__device__ __forceinline__ void mask_diagonal_tile(float* tile, int ld, int t, int nthr)
{
    for (int e = t; e < 64 * 16; e += nthr) {                  // one float4 column group per thread and pass
        const int l = e >> 4, c4 = (e & 15) * 4;
        float* dst = tile + l * ld + c4;
        if (c4 + 3 <= l) *reinterpret_cast<float4*>(dst) = float4{0.f, 0.f, 0.f, 0.f};
        else if (c4 <= l) { dst[0] = 0.f; if (c4 + 1 <= l) dst[1] = 0.f; if (c4 + 2 <= l) dst[2] = 0.f; }
    }
}

// ---- DENSE blocks of 256 / 512 markers (every marker of the block is included whatever its rhs: Pi = 0, RR-BLUP, BayesA,
// the reference's own benchmark setting).  The block chain is a forward substitution: marker c needs the changes of all
// markers before it.  Section s (64 markers) is walked by wave s exactly as dense_section walks a small block -- same
// operations, same order, bit-identical to the sequential chain -- with its 64 x 64 DIAGONAL Gram tile from LDS (all
// tiles are fetched with direct loads at the very start of the launch); everything off the diagonal runs in parallel:
// thread c owns row c (its running rhs in a register) and column c of the next block's lookahead correction, and after
// section s is done applies its 64 changes from Gram / cross-Gram values it prefetched into registers while the section
// was being walked (rhs = fmaf(D_k, G[k][c], rhs) in marker order: the sequential chain's own fmaf sequence).
// One barrier per section; the serial part per marker is dense_section's chain and nothing else.
template <int METHOD>
__device__ __forceinline__ void dense_big_st(char* smem, const StepSmem& SM, const SamplerArgs& A, float ie, long long tk0)
{
    const int B = SM.B, b = A.b, bn = A.b_next;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t j0 = A.j0;
    float* rhs_lds = reinterpret_cast<float*>(smem + SM.rhs_off);      // entry rhs; reused as D[c] = alpha_old - alpha_new once c is done
    float* acur = reinterpret_cast<float*>(smem + SM.acur_off);
    const float* astart = reinterpret_cast<const float*>(smem + SM.astart_off);
    const double* lpd = reinterpret_cast<const double*>(smem + SM.prepd_off);
    const float* lpf = reinterpret_cast<const float*>(smem + SM.prepf_off);
    const float* tiles = reinterpret_cast<const float*>(smem + SM.rows_off);      // [B/64][64][64] diagonal Gram tiles
    int* wcnt = reinterpret_cast<int*>(smem + SM.wcnt_off);
    const int c = tid;
    const bool own = c < b;
    const int cl = own ? c : 0;
    float rr = rhs_lds[cl];
    const float il = lpf[cl], dj = lpf[2 * B + cl];
    const float kc1 = lpf[B + cl], kc0 = lpf[3 * B + cl];               // Rule D: alpha_new = fmaf(kc1, x, kc0)  (see the front)
    const double zs = lpd[cl];
    const float ao = acur[cl];
    const float da = dj * ao;                                           // d * alpha_old (BayesABC.jl:36)
    float an_own = 0.f;
    const int nsec = (b + 63) >> 6;                                     // (a ragged last block: its last section is partial)
    // (A.xch: a HELPER workgroup forms the next block's lookahead correction from the changes each section publishes --
    // corr_helper, sampler_common.hpp: the cross-Gram block, 1 MB of the 1.5 MB a 512-marker block moves, stays off this CU)
    const bool offload = A.xch != nullptr;
    const bool has_col = tid < bn && !offload;
    float corr = 0.f;
    float gq[64], cq[64];
    // thread c reads column c of the Gram rows / cross-Gram rows of a section: one dword per lane, coalesced (256 bytes = two
    // lines per wave instruction).  (Reading the thread's own ROW instead -- G is symmetric -- as 16 dwordx4 loads was
    // measured 3x slower: 64 different lines per wave instruction keep the texture addresser busy ~360 cycles each.)
    // (address = UNIFORM row pointer + the lane's column: the load takes the row pointer from scalar registers and needs no
    // vector address arithmetic -- 128 loads per thread and section, and the seven waves doing this share the walker's CU)
    const int ccl = has_col ? tid : 0;
    auto load_g = [&](int s) {
        const char* base = reinterpret_cast<const char*>(A.gram + (int64_t)(64 * s) * b);      // G[k][c] = gram[k * b + c]
        unsigned off = 4u * (unsigned)cl;                               // (32-bit byte offset: one v_add per load)
#pragma unroll
        for (int u = 0; u < 64; ++u) { gq[u] = *reinterpret_cast<const float*>(base + off); off += 4u * (unsigned)b; asm volatile("" : "+v"(off)); }
    };
    auto load_c = [&](int s) {
        const char* base = reinterpret_cast<const char*>(A.cross_next + (int64_t)(64 * s) * bn);   // C[k][c'] = cross[k * bn + c']
        unsigned off = 4u * (unsigned)ccl;
#pragma unroll
        for (int u = 0; u < 64; ++u) { cq[u] = *reinterpret_cast<const float*>(base + off); off += 4u * (unsigned)bn; asm volatile("" : "+v"(off)); }
    };
    // Barrier of the section loop: LDS traffic only.  (__syncthreads() also waits for every outstanding GLOBAL load -- the
    // prefetches below are meant to stay in flight across it.)
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto prefetch = [&](int sec) {
        if (wave > sec && own) load_g(sec);
        if (has_col) load_c(sec);
    };
    prefetch(0);
    // (wave w walks tile w and nobody else reads it: it masks the tile itself -- LDS operations of one wave are in order)
    if (wave < nsec) mask_diagonal_tile(reinterpret_cast<float*>(smem + SM.rows_off) + wave * 4096, 64, lane, 64);
#pragma unroll 1
    for (int s = 0; s < nsec; ++s) {
        if (wave == s) {
            float r0 = rr, r1 = 0.f, rev = rr;
            dense_section<0, false, true, true>(tiles + s * 4096, 64, (b - 64 * s < 64) ? b - 64 * s : 64, lane, ie, 0.f, 0.f, il, da, ao, zs, r0, r1, rev, kc1, kc0);
            an_own = fmaf(kc1, rev, kc0);
            acur[c] = an_own;
            rhs_lds[c] = ao - an_own;                                   // D of this marker, read by everybody after the barrier
            if (offload)                                                // ... and by the helper workgroup: value + tag, fire and forget
                __hip_atomic_store(A.xch + c, ((unsigned long long)((unsigned)(A.xch_epoch + s + 1) & 0x7fffffffu) << 32) | (unsigned long long)__float_as_uint(ao - an_own),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        lds_barrier();
        // (Keep this body simple: a variant in which the next walker skipped the correction and caught up in a loop after its
        // walk made the register allocator keep several copies of the prefetch arrays alive -- 218 spilled VGPRs.)
        // The section's 64 changes first (16 broadcast reads back to back: one LDS latency instead of one per group of four),
        // then the two fmaf chains in marker order, interleaved
        float dv[64];
        {
            const float4* Dv4 = reinterpret_cast<const float4*>(rhs_lds + 64 * s);
#pragma unroll
            for (int u = 0; u < 16; ++u) { const float4 d = Dv4[u]; dv[4 * u] = d.x; dv[4 * u + 1] = d.y; dv[4 * u + 2] = d.z; dv[4 * u + 3] = d.w; }
        }
        // the wave that just walked did not prefetch its cross-Gram values (the issue sat between the barrier and its walk,
        // on the critical path): it fetches them now -- its rows are done, nobody waits for it
        if (wave == s && has_col && s > 0) load_c(s);
        const bool do_r = wave > s && own;
        if (do_r && has_col) {
#pragma unroll
            for (int u = 0; u < 64; ++u) { rr = fmaf(dv[u], gq[u], rr); corr = fmaf(dv[u], cq[u], corr); }
        } else if (do_r) {
#pragma unroll
            for (int u = 0; u < 64; ++u) rr = fmaf(dv[u], gq[u], rr);
        } else if (has_col) {
#pragma unroll
            for (int u = 0; u < 64; ++u) corr = fmaf(dv[u], cq[u], corr);
        }
        if (s + 1 < nsec) {
            if (wave > s + 1 && own) load_g(s + 1);
            if (has_col && wave != s + 1) load_c(s + 1);                // (the next walker: see above)
        }
    }
    __syncthreads();
    const long long tk4 = clock64();
    // the block's change list in marker order (an effect that came out bit-equal to the old one is no change)
    const bool changed = own && (ao != an_own);
    const unsigned long long cm = __ballot(changed);
    __syncthreads();                                                    // (the candidate counts of the front are no longer read)
    if (lane == 0) wcnt[wave] = __popcll(cm);
    __syncthreads();
    int base = 0, nfin = 0;
#pragma unroll
    for (int q = 0; q < kStepThreads / 64; ++q) { const int v = wcnt[q]; base += (q < wave) ? v : 0; nfin += v; }
    // ---- global stores last
    if (tid < B && bn > 0 && !offload) A.corr_out[tid] = has_col ? corr : 0.f;
    if (changed) {
        const int e = base + __popcll(cm & ((1ull << lane) - 1ull));
        const float d = ao - an_own;
        A.ev_out->idx[e] = (int32_t)(j0 + c);
        A.ev_out->delta[0][e] = d;
        if (e < 7) { A.ev_out->hidx[e] = (int32_t)(j0 + c); A.ev_out->hdelta[e] = d; }
        A.alpha[j0 + c] = an_own;
    }
    if (own) { A.beta[j0 + c] = an_own; reinterpret_cast<float*>(A.delta)[j0 + c] = 1.f; }
    if (tid == 0) {
        A.ev_out->count = (int32_t)nfin;
        atomicAdd(&A.counters[0], (unsigned long long)nfin);
        atomicAdd(&A.counters[5], (unsigned long long)(tk4 - tk0));      // (diagnostics: front + walk)
        atomicAdd(&A.counters[7], (unsigned long long)b);
    }
    (void)astart;
}


// ---------------------------------------------------------------------------------------------
// COMPACT CHAIN (round 4).  Single-pass sweeps of 256- / 512- / 1024-marker blocks with a sparse prior and MANY markers in the
// model (BayesR, a fixed pi, the first sweeps of a chain): the speculative rounds cost ~850 cycles per committed change plus
// ~1.8 k per 64-marker sub-block, and almost every change is a marker that was a candidate at block entry (it is in the
// model, or its entry rhs already passes its threshold).  When all nc <= 64 candidates have their Gram rows staged
// (slot i = the i-th candidate in marker order), wave 0 walks THE CANDIDATES ONLY, dense-walk style: lane i owns candidate
// i and evaluates it against its own running rhs at every step; at step i lane i's alpha_old - alpha_new is broadcast with
// one v_readlane and applied to the running rhs of the candidates after it with G[cand_i][cand_l] from the staged rows --
// per element the sequential chain's own fmaf sequence in commit order.  That is the exact chain PROVIDED no other marker
// of the block changes; so afterwards all threads bring the rhs of every non-candidate up to the value it has when the
// chain reaches it (entry rhs + the changes of the candidates before it, same fmaf sequence) and test it against its
// threshold.  Nobody crosses (the usual case: a few percent of the blocks hold a "surprise"): the walk's results are the
// chain's, bit for bit, and are committed -- effects, classes and the change log, all at once.  Somebody crosses: nothing
// has been written, the block runs through the speculative rounds from its untouched entry state.
// Reference chain: BayesABC.jl:153-179, BayesR.jl:150-189.
// ---------------------------------------------------------------------------------------------
// Candidates at block entry: markers in the model, markers whose entry rhs passes their threshold -- and markers whose entry
// rhs comes within 1/16 of it: the changes committed inside the block move a later marker's rhs by a few percent of its
// threshold, and a marker that crosses it without having been a candidate ("surprise") costs the compact chain its block
// (fixed pi = 0.95: 17 % of the blocks without the margin).  A candidate that does not move is an exact no-op everywhere
// (its row is staged, it is walked, its change is 0): the margin changes speed only.
constexpr float kCandMargin = 0.9375f;
constexpr float kCandMarginLate = 0.90f;       // ping-pong blocks that stage before their right-hand side is finished (sampler_role_st, pp_late)
constexpr int kCompactMin = 6;          // fewer candidates: the speculative rounds are as fast
constexpr int kCompactMax = 64;         // one lane per candidate

// scratch of the compact chain: the overflow Gram-row slot (B >= 256 floats; rewritten by the general path before it reads it)
struct CompactScratch {
    int2* log;        // [64] {local column, bits(alpha_old - alpha_new)} of candidate i
    float* an;        // [64] its new effect
    int* cls;         // [64] BayesR: its class 0..3
    __device__ __forceinline__ CompactScratch(char* smem, const StepSmem& SM)
    {
        float* base = reinterpret_cast<float*>(smem + SM.rows_off) + SM.max_cand * SM.B;
        log = reinterpret_cast<int2*>(base); an = base + 128; cls = reinterpret_cast<int*>(base + 192);
    }
};

// wave 0.  Returns false when a BayesR evaluation lay too close to a class threshold to be decided from the thresholds
// (practically never): the caller falls back.
template <int METHOD>
__device__ __forceinline__ bool compact_walk(char* smem, const StepSmem& SM, int nc, float ie, int lane)
{
    constexpr bool kR = (METHOD == kBayesR);
    const int B = SM.B;
    const short* cand_list = reinterpret_cast<const short*>(smem + SM.cand_off);
    const float* rows = reinterpret_cast<const float*>(smem + SM.rows_off);
    const float* rhs_lds = reinterpret_cast<const float*>(smem + SM.rhs_off);
    const float* acur = reinterpret_cast<const float*>(smem + SM.acur_off);
    const double* lpd = reinterpret_cast<const double*>(smem + SM.prepd_off);
    const float* lpf = reinterpret_cast<const float*>(smem + SM.prepf_off);
    const CompactScratch CS(smem, SM);
    const bool act = lane < nc;
    const int col = (int)cand_list[act ? lane : 0];
    float rhs = rhs_lds[col];
    const float a_cur = acur[col];
    float c_lo = 0.f, c_hi = 0.f, c_il = 0.f, c_d = 0.f;
    double c_zs = 0.0;
    double r_il1 = 0.0, r_il2 = 0.0, r_il3 = 0.0, r_zs1 = 0.0, r_zs2 = 0.0, r_zs3 = 0.0, r_T0 = 0.0, r_T1 = 0.0, r_T2 = 0.0;
    if constexpr (kR) {
        c_d = lpf[col];
        r_il1 = lpd[0 * B + col]; r_il2 = lpd[1 * B + col]; r_il3 = lpd[2 * B + col];
        r_zs1 = lpd[3 * B + col]; r_zs2 = lpd[4 * B + col]; r_zs3 = lpd[5 * B + col];
        r_T0 = lpd[6 * B + col]; r_T1 = lpd[7 * B + col]; r_T2 = lpd[8 * B + col];
    } else {
        c_il = lpf[col]; c_d = lpf[2 * B + col]; c_lo = lpf[3 * B + col]; c_hi = lpf[4 * B + col];
        c_zs = lpd[col];
    }
    int cls = 0;
    bool sure = true;
    auto eval = [&](float x) -> float {
        float an;
        if constexpr (kR) cls = bayesr_eval_thr(x, a_cur, ie, c_d, r_il1, r_il2, r_il3, r_zs1, r_zs2, r_zs3, r_T0, r_T1, r_T2, an, sure);
        else an = abc_alpha_new(x, a_cur, c_d, ie, c_il, c_zs, abc_included(x, c_lo, c_hi));
        return an;
    };
    // G[cand_i][cand_lane] = rows[i * B + col]; lane l takes the changes of the candidates BEFORE it only (i < l), so its
    // running rhs stops moving at its own step and its last evaluation is the one of its own step
    const float* gcol = rows + col;
    auto step = [&](int i, float g) {
        const float Dl = a_cur - eval(rhs);
        const float D = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Dl), i));
        rhs = fmaf(D, (i < lane) ? g : 0.f, rhs);
    };
    constexpr int kBatch = 8;
    float gn[kBatch];
    int i = 0;
    if (nc >= kBatch) {
#pragma unroll
        for (int u = 0; u < kBatch; ++u) gn[u] = gcol[u * B];
    }
#pragma unroll 1
    for (; i + kBatch <= nc; i += kBatch) {
        float g[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) g[u] = gn[u];
        if (i + 2 * kBatch <= nc) {
#pragma unroll
            for (int u = 0; u < kBatch; ++u) gn[u] = gcol[(i + kBatch + u) * B];
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) step(i + u, g[u]);
    }
#pragma unroll 1
    for (; i < nc; ++i) step(i, gcol[i * B]);
    const float an = eval(rhs);                            // the evaluation of the lane's own step (its rhs is final)
    if (act) {
        CS.log[lane] = make_int2(col, __float_as_int(a_cur - an));
        CS.an[lane] = an;
        CS.cls[lane] = cls;
    }
    return __all(sure || !act);
}

// all threads, after the walk: does a marker that was NO candidate at block entry change when the chain reaches it?
// cand[q]: marker c = tid + q * kStepThreads was a candidate (walked).
template <int METHOD>
__device__ __forceinline__ bool compact_surprise(char* smem, const StepSmem& SM, int nc, int b, const bool (&cand)[2])
{
    constexpr bool kR = (METHOD == kBayesR);
    const int B = SM.B;
    const int tid = threadIdx.x, lane = tid & 63;
    const float* rows = reinterpret_cast<const float*>(smem + SM.rows_off);
    const float* rhs_lds = reinterpret_cast<const float*>(smem + SM.rhs_off);
    const float* lpf = reinterpret_cast<const float*>(smem + SM.prepf_off);
    const CompactScratch CS(smem, SM);
    bool surprise = false;
    const int mycol = CS.log[lane < nc ? lane : 0].x;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (q * kStepThreads >= B) break;
        const int c = tid + q * kStepThreads;
        if ((c & ~63) >= b) continue;                       // (wave-uniform: none of the wave's columns is a marker)
        const int cmax = (c | 63) + 1;                      // the wave's columns are [cmax - 64, cmax)
        // candidates before the wave's last column (the list is in marker order): wave-uniform loop bound
        const int nlim = __popcll(__ballot(lane < nc && mycol < cmax));
        // ... and the candidates before the wave's FIRST column reach every lane: whole batches of them need no per-lane test
        const int nf8 = __popcll(__ballot(lane < nc && mycol < cmax - 64)) & ~7;
        float r = rhs_lds[c < B ? c : 0];
#pragma unroll 1
        for (int i0 = 0; i0 < nf8; i0 += 8) {
            float d[8], g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { d[u] = __int_as_float(CS.log[i0 + u].y); g[u] = rows[(i0 + u) * B + (c < B ? c : 0)]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) r = fmaf(d[u], g[u], r);
        }
#pragma unroll 1
        for (int i0 = nf8; i0 < nlim; i0 += 8) {
            int2 e[8];
            float g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u < nlim ? i0 + u : nlim - 1;
                e[u] = CS.log[i];
                g[u] = rows[i * B + (c < B ? c : 0)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u < nlim) r = fmaf((e[u].x < c) ? __int_as_float(e[u].y) : 0.f, g[u], r);
        }
        bool ev;
        if constexpr (kR) ev = fabsf(r) >= lpf[B + (c < B ? c : 0)];
        else ev = abc_included(r, lpf[3 * B + (c < B ? c : 0)], lpf[4 * B + (c < B ? c : 0)]);
        surprise = surprise || (ev && c < b && !cand[q]);
    }
    return surprise;
}

__host__ __device__ constexpr int st_park_nd(int method) { return method == kBayesR ? BayesRMarker::kFastD : 1; }
__host__ __device__ constexpr int st_park_nf(int method, bool dense = false) { (void)dense; return method == kBayesR ? 2 : 5; }

// DENSE: the instantiation for sweeps under a UNIFORM PRIOR pi = 0 (single-trait BayesA/B/C: RR-BLUP, BayesA, BayesL, the
// reference's benchmark setting), selected by the host: every marker follows Rule D (AbcMarker::rule_d) on every path of
// it, and full 256- / 512-marker blocks take dense_big_st.  The steady-state kernel is compiled without any of it.
// GROUP (k_group_step): the block is one of a group of consecutive blocks sampled by ONE launch -- three lookahead corrections
// instead of one, its changes appended to the group's merged list at ev_base.  Returns the length of that list behind the block.
// PP: the grouped kernel's ping-pong instantiation (hand-over words, late hand-over); without it none of that code exists in the kernel
// (the steady-state grouped launches: its mere presence cost the 2-bit packed sweep 7 %).
template <int METHOD, bool DENSE = false, bool GROUP = false, bool PP = false>
__device__ __forceinline__ int sampler_role_st(char* smem, const SamplerArgs& A, int ev_base = 0)
{
    static_assert(GROUP || !PP, "ping-pong samplers exist in grouped launches only");
    constexpr bool kPP = GROUP && PP;
    static_assert(!(GROUP && DENSE), "grouped launches run the general single-trait sampler");
    constexpr bool kR = (METHOD == kBayesR);
    constexpr int ND = st_park_nd(METHOD), NF = st_park_nf(METHOD, DENSE);
    const StepSmem SM(A.bsz, 1, ND, NF);
    const int B = SM.B;
    const DevParams* P = A.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = A.b;
    const int64_t j0 = A.j0, p = A.p;
    const float ie = 1.0f / P->vare[0];
    double* lpd = reinterpret_cast<double*>(smem + SM.prepd_off);     // [ND][B]
    float* lpf = reinterpret_cast<float*>(smem + SM.prepf_off);       // [NF][B]
    float* rhs_lds = reinterpret_cast<float*>(smem + SM.rhs_off);
    float* acur = reinterpret_cast<float*>(smem + SM.acur_off);
    float* astart = reinterpret_cast<float*>(smem + SM.astart_off);
    float* bpark0 = reinterpret_cast<float*>(smem + SM.bcur_off);      // [B] beta / [B] delta of the block (single trait:
    float* dpark0 = reinterpret_cast<float*>(smem + SM.dcur_off);      // the multi-trait slots are free)
    const long long tk0 = clock64();

    // ---- phase A (all threads, ONE memory latency): for its marker every thread issues, back to back, the
    // loads of alpha, the sweep constants, the row-group partials and the lookahead correction; then
    //   rhs = fl32(sum of partials) + corr
    // and decides candidacy (does the effect change if evaluated against the entry rhs?) with the marker's
    // thresholds: two float compares.  Under full-rate streaming by the update role a dependent global load costs
    // microseconds, so nothing here waits twice.
    // Small blocks (B <= 128: the host's choice for dense priors): the whole Gram block fits the row slots, and it does
    // not depend on anything this launch computes -- fetch it with the very first loads instead of after the candidates
    // are known (one dependent memory latency less per block).  Slot of marker c = c.
    const bool prestage = (B <= 128) && (B <= SM.max_cand);
    // full blocks: the Gram block (and the cross-Gram rows X_this'X_next, when they have their own LDS room and the next
    // block is full too) go straight to LDS with direct loads issued before anything else: a handful of instructions
    // instead of ~200 lines of cold unrolled code (instruction fetch after a dispatch runs at memory latency)
    const bool gram_dma = prestage && b == B;
    const bool cross_dma = gram_dma && SM.has_cross && A.b_next == B;
    if (gram_dma) dma_copy_to_lds(A.gram, reinterpret_cast<float*>(smem + SM.rows_off), B * B);
    // 256- / 512-marker blocks under a prior that includes every marker (Pi = 0 without per-marker pi: known before any
    // load): dense_big_st.  Its diagonal Gram tiles (64 x 64 floats per 64-marker section: wave w fetches tile w) go to
    // LDS with direct loads issued before anything else; whether every marker really is "always included" (thresholds
    // lo = hi) is voted below, and a block that fails the vote runs the general path (which re-stages the row slots).
    bool dense_big_try = false;
    if constexpr (!kR && DENSE) {
        // (full blocks followed by any block or none; also a ragged LAST block -- its last section is walked over the markers
        // it has: without this the sweep's last two blocks, 672 markers at p = 100 000, went through the speculative rounds with
        // ~550 Gram rows fetched on demand, 1.5 of the 9.4 ms of the reference benchmark shape)
        dense_big_try = (B == 256 || B == 512) && (b == B || (A.b_next == 0 && (b & 3) == 0 && b >= 64)) &&
                        (P->nreps == 1) && P->pi == 0.0 && P->pi_vec == nullptr && !A.dense_big_off;
        if (dense_big_try && wave < ((b + 63) >> 6)) {
            typedef __attribute__((address_space(3))) void lds_void;
            float* tile = reinterpret_cast<float*>(smem + SM.rows_off) + wave * 4096;
            const float* src = A.gram + (int64_t)(64 * wave + (lane >> 4)) * b + 64 * wave + (lane & 15) * 4;
#pragma unroll 1
            for (int r4 = 0; r4 < 16; ++r4)       // 4 rows of 64 floats per instruction: lane -> row lane / 16, float4 column lane % 16
                __builtin_amdgcn_global_load_lds(src + (int64_t)(4 * r4) * b, (lds_void*)(tile + r4 * 256), 16, 0, 0);
        }
    }
    bool always_mine = true;
    // (the cross-Gram rows are only needed after the walk: waves 1..7 fetch them while wave 0 walks)
    float4 gpre[8];
    if (prestage && !gram_dma) {
        // B*B/4 float4 elements over 512 threads: <= 8 per thread; element e -> row e / (B/4), float4 column e % (B/4)
        const int per_row = B >> 2, total = b * per_row;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + u * kStepThreads;
            const int ec = e < total ? e : 0;
            const int row = ec / per_row, c4 = (ec - row * per_row) * 4;
            // rows are b floats apart in global memory (b may be < B for the last block): element-wise clamped loads
            const float* src = A.gram + (int64_t)row * b;
            if (b == B) gpre[u] = *reinterpret_cast<const float4*>(src + c4);       // full block: rows are 16-byte aligned
            else {
                gpre[u].x = src[c4 < b ? c4 : 0]; gpre[u].y = src[c4 + 1 < b ? c4 + 1 : 0];
                gpre[u].z = src[c4 + 2 < b ? c4 + 2 : 0]; gpre[u].w = src[c4 + 3 < b ? c4 + 3 : 0];
            }
        }
    }
    bool cand[2] = {false, false};
    // PING-PONG (SamplerArgs::pp_cw_in / pp_cp_in: a block sampled by its own workgroup while the blocks before it are still being
    // walked): cW and / or cP do not exist yet.  Everything else of the front -- one memory latency -- happens now; the right-hand side
    // and the candidacy are finished below, when the workgroups before this one have posted them.
    bool pp_wait = false;
    if constexpr (kPP) pp_wait = A.pp_cw_in != nullptr || A.pp_cp_in != nullptr;
    // LATE hand-over (single-pass sweeps of >= 256-marker blocks): the candidacy is decided PROVISIONALLY from the right-hand side
    // without the terms that are still on their way, with a wider margin (kCandMarginLate), and the slots, the candidates' Gram rows
    // and the cross-Gram pieces are staged BEFORE the block waits -- slot assignment and the rows' memory latency leave the chain of the
    // group's blocks too.  When the terms arrive the right-hand side is finished; a marker that is a candidate only now (rare: it
    // needs a shift of > 4 % of its threshold) sends the block through the speculative rounds from its first sub-block, where every
    // marker is evaluated against the finished right-hand side (rows that are not staged are fetched on demand) -- exact either way,
    // the provisional set only decides what is walked and staged.  (Final form: such a block is simply staged a second time with the
    // union of the two candidate sets -- what it cost before -- and goes on as ever.)
    const bool pp_late = kPP && pp_wait && !prestage && ((P->nreps > 0 ? P->nreps : b) == 1) && !(A.compact_off & 4);
    const float cmarg = pp_late ? kCandMarginLate : kCandMargin;
    float pp_sum[2] = {0.f, 0.f}, pp_c1[2] = {0.f, 0.f}, pp_c2[2] = {0.f, 0.f}, pp_c3[2] = {0.f, 0.f};
    // TWO STEPS (round 6): the loads of the front -- state, constants and lookahead corrections of BOTH columns of a thread when the
    // block has more than 512 markers, and the first column's row-group partial sums -- are issued before anything waits for one of
    // them; the second column's partial sums follow when the first column's have been summed (all of them at once do not fit the
    // registers: the compiler then spills the first column's values, i.e. waits for them, before it issues the second's).  Written
    // as one loop over the columns with the sums inside, the compiler finished column 0 (loads, sums, LDS stores) before it issued
    // column 1's loads, and split 28 partial sums into two batches with a full wait between them: four memory latencies at the start
    // of every 1024-marker block (n = 50 000), each of them microseconds under the update role's stream.  N = partial sums loaded
    // per batch (nrg <= N, else the tall-matrix loop of sum_loaded_n), NQ = columns per thread; uniform branches pick the instantiation.
    auto front = [&](auto Nc, auto NQc) {
        constexpr int N = decltype(Nc)::value, NQ = decltype(NQc)::value;
        float a0q[NQ], djq[NQ], co1q[NQ], co2q[NQ], co3q[NQ];
        float thrxq[NQ], invLhsq[NQ], bexq[NQ], loq[NQ], hiq[NQ], c1q[NQ], c0q[NQ];
        double zsq[NQ], pv0[N];
        BayesRMarker bmq[NQ];
        // ---- step 1: loads only
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = tid + q * kStepThreads;
            const int cc = c < b ? c : 0, cx = c < B ? c : 0;       // (a column the block does not have: marker 0's values, never used)
            const int64_t j = j0 + cc;
            a0q[q] = A.alpha[j];
            djq[q] = A.xpx[j];
            co1q[q] = A.corr_in[cx]; co2q[q] = 0.f; co3q[q] = 0.f;
            // (grouped launches: unconditional loads from always-valid buffers, the VALUE selected below: a load under a condition on
            // a pointer makes hipcc choose between address spaces)
            if constexpr (GROUP) { co2q[q] = A.corr_in2[cx]; co3q[q] = A.corr_in3[cx]; }
            thrxq[q] = 0.f; invLhsq[q] = 0.f; bexq[q] = 0.f; loq[q] = 0.f; hiq[q] = 0.f; c1q[q] = 0.f; c0q[q] = 0.f; zsq[q] = 0.0;
            if constexpr (kR) {
                bmq[q].load_fast_global(A.prep_d, p, j, djq[q], ie);
                thrxq[q] = A.prep_f[j];
            } else {
                zsq[q] = A.prep_d[3 * p + j];
                invLhsq[q] = A.prep_f[j]; bexq[q] = A.prep_f[2 * p + j]; loq[q] = A.prep_f[3 * p + j]; hiq[q] = A.prep_f[4 * p + j];
                if constexpr (DENSE) { c1q[q] = A.prep_f[5 * p + j]; c0q[q] = A.prep_f[6 * p + j]; }
            }
            if (q == 0) load_partials_n<N>(A.partials + cc, A.nrg, A.bstride, pv0);
        }
        // (nothing of step 2 may be scheduled in between: the first addition of a sum needs one load only, and hoisted above the
        // other loads it makes the wait that follows it a wait for everything issued so far)
        __builtin_amdgcn_sched_barrier(0);
        // ---- step 2: sums, right-hand sides, candidacy, LDS
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = tid + q * kStepThreads;
            if (c >= B) continue;                     // (B may be smaller than the workgroup)
            const int cc = c < b ? c : 0;
            const float a0 = a0q[q], dj = djq[q];
            float co = 0.f;
            if constexpr (GROUP) {
                const float co2 = co2q[q];
                float co1 = co1q[q], co3 = co3q[q];
                if constexpr (kPP) {
                    if (A.pp_cw_in != nullptr) co1 = 0.f;
                    if (A.pp_cp_in != nullptr) co3 = 0.f;
                    pp_c1[q] = co1; pp_c2[q] = co2; pp_c3[q] = co3;
                }
                co = (co1 + co2) + co3;                                       // (final unless a term is still on its way: pp_wait)
            } else co = co1q[q];
            double sum;
            if (q == 0) sum = sum_loaded_n<N>(pv0, A.partials + cc, A.nrg, A.bstride);
            else {
                double pv1[N];
                load_partials_n<N>(A.partials + cc, A.nrg, A.bstride, pv1);
                __builtin_amdgcn_sched_barrier(0);
                sum = sum_loaded_n<N>(pv1, A.partials + cc, A.nrg, A.bstride);
            }
            if constexpr (kPP) pp_sum[q] = (float)sum;
            const float rhs0 = (float)sum + co;       // + lookahead correction formed by the previous block's sampler
            rhs_lds[c] = rhs0;
            const float a_in = (c < b) ? a0 : 0.f;
            acur[c] = a_in; astart[c] = a_in;
            if constexpr (kR) {
                const float thrx = thrxq[q];
                bmq[q].store_fast(lpd, B, c);
                lpf[c] = dj; lpf[B + c] = thrx;
                cand[q] = (c < b) && ((a_in != 0.f) || (fabsf(rhs0) >= thrx * cmarg));
                dpark0[c] = 1.f;                      // a marker that stays out: class 1 (see the prefix skip below)
            } else {
                const float invLhs = invLhsq[q], bex = bexq[q], lo = loq[q], hi = hiq[q];
                lpd[c] = zsq[q];
                lpf[c] = invLhs; lpf[B + c] = bex; lpf[2 * B + c] = dj; lpf[3 * B + c] = lo; lpf[4 * B + c] = hi;
                // Rule D sweeps: nothing is ever excluded, so the slots of the "excluded" draw and of the lower threshold carry
                // c1 and c0 instead (512-marker blocks leave no LDS for two more rows next to dense_big_st's 128 KB of tiles)
                if constexpr (DENSE) { lpf[B + c] = c1q[q]; lpf[3 * B + c] = c0q[q]; }
                cand[q] = (c < b) && ((a_in != 0.f) || abc_included(rhs0, lo * cmarg, hi * cmarg));      // (alpha = 0: lo <= 0 <= hi)
                always_mine = always_mine && ((c >= b) || (lo == hi));           // thresholds(): lo = hi <=> always included
                bpark0[c] = bex; dpark0[c] = 0.f;     // a marker that stays out: delta 0, beta = its excluded draw
            }
        }
    };
    {
        using std::integral_constant;
        typedef integral_constant<int, 1> Q1; typedef integral_constant<int, 2> Q2;
        if (B > kStepThreads) {
            if (A.nrg <= 8) front(integral_constant<int, 8>{}, Q2{});
            else if (A.nrg <= 16) front(integral_constant<int, 16>{}, Q2{});
            else front(integral_constant<int, 32>{}, Q2{});
        } else {
            if (A.nrg <= 8) front(integral_constant<int, 8>{}, Q1{});
            else if (A.nrg <= 16) front(integral_constant<int, 16>{}, Q1{});
            else front(integral_constant<int, 32>{}, Q1{});
        }
    }
    if constexpr (kPP) {
        if (pp_wait && !pp_late) {
            // the hand-over: every thread polls the tagged words of ITS markers (block 0's sampler posts them the moment its
            // correction chain is done), then   rhs = fl32(sum) + ((cW + cG) + cP)   and the candidacy, exactly as above
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = tid + q * kStepThreads;
                if (c >= B) continue;
                const float cw = (A.pp_cw_in != nullptr) ? __uint_as_float(pp_wait_word(A.pp_cw_in + c, A.pp_tag, A.counters)) : pp_c1[q];
                const float cp = (A.pp_cp_in != nullptr) ? __uint_as_float(pp_wait_word(A.pp_cp_in + c, A.pp_tag, A.counters)) : pp_c3[q];
                const float co = (cw + pp_c2[q]) + cp;
                const float rhs0 = pp_sum[q] + co;
                rhs_lds[c] = rhs0;
                const float a_in = acur[c];
                if constexpr (kR) cand[q] = (c < b) && ((a_in != 0.f) || (fabsf(rhs0) >= lpf[B + c] * kCandMargin));
                else cand[q] = (c < b) && ((a_in != 0.f) || abc_included(rhs0, lpf[3 * B + c] * kCandMargin, lpf[4 * B + c] * kCandMargin));
            }
        }
    }
    if (prestage) {
        float* rows_p = reinterpret_cast<float*>(smem + SM.rows_off);
        short* slot_p = reinterpret_cast<short*>(smem + SM.slot_off);
        short* cand_p = reinterpret_cast<short*>(smem + SM.cand_off);
        if (!gram_dma) {
            const int per_row = B >> 2, total = b * per_row;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + u * kStepThreads;
                if (e < total) {
                    const int row = e / per_row, c4 = (e - row * per_row) * 4;
                    *reinterpret_cast<float4*>(rows_p + row * B + c4) = gpre[u];
                }
            }
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the direct loads have landed (barrier below)
        for (int c = tid; c < B; c += kStepThreads) { slot_p[c] = (short)(c < b ? c : -1); cand_p[c] = (short)c; }
    }
    // number of markers whose effect changes against the entry rhs (block-wide count through the wave-count slots;
    // __syncthreads_count would add static LDS on top of the 160 KB dynamic carve)
    {
        int* wc = reinterpret_cast<int*>(smem + SM.wcnt_off);
        const unsigned long long mb = __ballot(cand[0] || cand[1]);
        // bits 16 / 17: this wave's markers of sub-block `wave` / `8 + wave` contain a candidate
        const int f0 = __any(cand[0]) ? 1 << 16 : 0, f1 = __any(cand[1]) ? 1 << 17 : 0;     // (votes outside the lane-0 branch)
        const int f2 = __all(always_mine) ? 1 << 18 : 0;                 // bit 18: every marker of this wave is always included
        if (lane == 0) wc[wave] = __popcll(mb) | f0 | f1 | f2;
    }
    if (dense_big_try) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the diagonal tiles have landed (barrier below)
    __syncthreads();
    int ncand_all = 0;
    // PREFIX SKIP: until the first candidate of the block commits, the running rhs IS the entry rhs, so the evaluation
    // every thread just did is final for all markers before it -- they stay out of the model (beta / delta parked
    // above) and the serial wave starts at the first sub-block that holds a candidate.  With a sparse prior that is
    // half of the sub-blocks on average, and all of them in the blocks without a candidate.
    int first_sub = 16;
    {
        const int* wc = reinterpret_cast<const int*>(smem + SM.wcnt_off);
        unsigned mask = 0u;
#pragma unroll
        for (int q = 0; q < kStepThreads / 64; ++q) {
            const int v = wc[q];
            ncand_all += v & 0xffff;
            mask |= ((v >> 16) & 1u) << q | ((v >> 17) & 1u) << (8 + q);
        }
        if (mask) first_sub = __builtin_ctz(mask);
    }
    if constexpr (!kR && DENSE) {
        if (dense_big_try) {
            const int* wc = reinterpret_cast<const int*>(smem + SM.wcnt_off);
            bool all_in = true;
#pragma unroll
            for (int q = 0; q < kStepThreads / 64; ++q) all_in = all_in && ((wc[q] >> 18) & 1);
            if (all_in) { dense_big_st<METHOD>(smem, SM, A, ie, tk0); return 0; }
            // (a helper workgroup is waiting for this block's sections: tell it that the general path forms the correction itself)
            if (A.xch != nullptr && tid < 64)
                __hip_atomic_store(A.xch + tid, (unsigned long long)(kXchAbort | ((unsigned)(A.xch_epoch + 1) & 0x7fffffffu)) << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // single-pass sweeps with a next block: waves 1..4 accumulate its lookahead correction while the serial wave runs
    const bool stream_corr = ((P->nreps > 0 ? P->nreps : b) == 1) && !prestage && A.b_next > 0;
    if (tid == 0) { int* wc0 = reinterpret_cast<int*>(smem + SM.wcnt_off); wc0[8] = 0; wc0[9] = 0; wc0[10] = 0; wc0[12] = 0; wc0[13] = 0; wc0[14] = 0; }
    if (stream_corr) for (int c = tid; c < B; c += kStepThreads) reinterpret_cast<int2*>(smem + SM.log_off)[c] = make_int2(-1, 0);
    // small dense blocks (most markers are candidates): the serial wave walks them section by section (below) -- on strictly
    // upper diagonal tiles (mask_diagonal_tile); the rows are not read again after the walk (single pass)
    const bool dense_walk = !kR && ((P->nreps > 0 ? P->nreps : b) == 1) && prestage && 5 * ncand_all >= 3 * b;
    if (dense_walk) {
        float* rows_m = reinterpret_cast<float*>(smem + SM.rows_off);
        for (int q = 0; q < (B >> 6); ++q) mask_diagonal_tile(rows_m + (64 * q) * B + 64 * q, B, tid, kStepThreads);
    }
    __syncthreads();                               // (stage_rows reuses the slots)
    const long long tk1 = clock64();
    const long long tk2 = clock64();
    // (a block without any candidate -- about a third of them with a sparse prior -- has nothing to stage or to walk)
    long long tss[3] = {tk2, tk2, tk2};
    // (a block without any candidate has nothing to stage or to walk -- in a SINGLE pass.  With within-block repetitions a
    // later repetition draws anew and may move a marker that was no candidate at entry: stage_rows must then have marked
    // every marker "not staged" (slot -1), or the winner's row would be looked up through a stale slot.)
    const bool single_pass_st = (P->nreps > 0 ? P->nreps : b) == 1;
    int ncand_total = 0;
    // COMPACT CHAIN (below): the lookahead correction of the NEXT block needs the cross-Gram rows of the markers that moved --
    // candidates all.  With a full next block waves 1..7 fetch the candidates' rows into REGISTERS (float4 pieces, task
    // T = k * 448 + (wave - 1) * 64 + lane -> row T / (B/4), piece T % (B/4): <= 19 per lane), issued together with the Gram
    // rows' own loads (one memory latency for both); once the walk is verified the staged Gram rows are dead and the pieces
    // take their place in LDS, so the correction is a chain of LDS reads -- no global access after the walk.  (Left to the
    // log-consuming helper waves it was 21 k cycles of L2 / HBM round trips per block behind the walk.)
    constexpr int kXR = 19, kXL = kStepThreads - 64;
    // (1024-marker blocks: from ONE candidate on -- the speculative rounds walk every 64-marker sub-block behind the first candidate,
    // ~8 rounds of ~1.1 k cycles per block in a sparse steady state; 2-bit packed config 2 10.74 -> 10.26 ms per sweep, NOTES R5.
    // JWAS_HIP_COMPACT_OFF = 256 n: experiments -- the chain from n candidates on)
    const int cmin = (A.compact_off >> 8) ? (A.compact_off >> 8) : (B == 1024 ? 1 : kCompactMin);
    const bool xreg_geom = !DENSE && single_pass_st && !prestage && !(A.compact_off & 1) &&
                           (A.b_next == B) && (b == B) && (B == 256 || B == 512 || B == 1024);
    const int xsh = (B == 1024) ? 8 : (B == 512 ? 7 : 6);
    const int xlid = tid - 64;
    typedef float xr_v4f __attribute__((ext_vector_type(4)));      // (a native vector: an array of HIP's float4 struct stayed in scratch memory)
    xr_v4f xr[kXR];
    bool xreg = false;
    int nstaged = prestage ? b : 0;
    bool pp_newc = false;
    // (two passes at most: the second one only for a ping-pong block whose finished right-hand side shows a candidate the
    // provisional staging did not see -- it is staged again with the union, i.e. at the cost the block had before pp_late)
#pragma unroll 1
    for (int pass = 0; pass < (kPP ? 2 : 1); ++pass) {
    if (!prestage && !(first_sub >= 16 && single_pass_st)) {
        nstaged = stage_assign(smem, SM, A, cand, ncand_total);
        xreg = xreg_geom && ncand_total >= cmin && ncand_total <= kCompactMax && ncand_total <= SM.max_cand;
        // the Gram rows' direct loads FIRST, the cross-Gram pieces behind them: the walk needs the rows, the pieces are needed
        // after it -- the wait below is for the rows only (vmcnt counts in order), the pieces arrive while wave 0 walks
        // (waiting for both cost 5.8 k cycles per block: ~190 KB through one CU's memory pipe under the stream's load)
        const bool split = xreg && stage_load(smem, SM, A, nstaged, tss, 1);
        // The wait of phase 2 counts on EXACTLY the kXR loads below being younger than the direct loads: the two compiler
        // barriers pin them between phase 1 and the wait (no global load may be hoisted above the direct loads or sunk below
        // the wait); a 16-byte load is the widest there is, so they cannot be merged into fewer, and all kXR results are used.
        // (Splitting one into several would only make the wait longer.)
        asm volatile("" ::: "memory");
        if (xreg && wave != 0) {
            const short* cl = reinterpret_cast<const short*>(smem + SM.cand_off);
            const int xtask = ncand_total << xsh;
#pragma unroll
            for (int k = 0; k < kXR; ++k) {
                const int T = k * kXL + xlid, Tc = T < xtask ? T : xtask - 1;
                xr[k] = *reinterpret_cast<const xr_v4f*>(A.cross_next + ((int64_t)cl[Tc >> xsh] * B + 4 * (Tc & ((1 << xsh) - 1))));
            }
        }
        static_assert(kXR == kStageYounger, "the number of loads stage_load leaves in flight");
        asm volatile("" ::: "memory");
        if (split) stage_load(smem, SM, A, nstaged, tss, 2, true);
        else stage_load(smem, SM, A, nstaged, tss);
    }
    if (!kPP || pass == 1) break;
    // ---- the LATE hand-over (pp_late, above): the block has staged everything it could; now it waits
    bool again = false;
    if constexpr (kPP) {
        if (pp_late) {
            bool nw = false;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = tid + q * kStepThreads;
                if (c >= B) continue;
                const float cw = (A.pp_cw_in != nullptr) ? __uint_as_float(pp_wait_word(A.pp_cw_in + c, A.pp_tag, A.counters)) : pp_c1[q];
                const float cp = (A.pp_cp_in != nullptr) ? __uint_as_float(pp_wait_word(A.pp_cp_in + c, A.pp_tag, A.counters)) : pp_c3[q];
                const float co = (cw + pp_c2[q]) + cp;
                const float rhs0 = pp_sum[q] + co;            // rhs = fl32(sum) + ((cW + cG) + cP), as ever
                rhs_lds[c] = rhs0;
                const float a_in = acur[c];
                bool ct;
                if constexpr (kR) ct = (c < b) && ((a_in != 0.f) || (fabsf(rhs0) >= lpf[B + c] * kCandMargin));
                else ct = (c < b) && ((a_in != 0.f) || abc_included(rhs0, lpf[3 * B + c] * kCandMargin, lpf[4 * B + c] * kCandMargin));
                nw = nw || (ct && !cand[q]);
                cand[q] = cand[q] || ct;                      // (the union: what was staged stays a candidate -- a no-op where it does not move)
            }
            int* wc = reinterpret_cast<int*>(smem + SM.wcnt_off);
            if (__any(nw) && lane == 0) wc[10] = 1;
            // (the sub-blocks that hold a candidate, once more: wave slots 0..7 of the vote are free again)
            const int f0 = __any(cand[0]) ? 1 : 0, f1 = __any(cand[1]) ? 2 : 0;
            __syncthreads();                                  // (every wave has read the first vote long ago; rhs_lds is complete)
            pp_newc = wc[10] != 0;
            if (pp_newc) {
                // a candidate the provisional evaluation did not see: slots, rows and pieces once more for the union, and the first
                // sub-block with a candidate from the finished right-hand side
                if (lane == 0) wc[wave] = f0 | f1;
                __syncthreads();
                unsigned mask = 0u;
#pragma unroll
                for (int q = 0; q < kStepThreads / 64; ++q) { const int v = wc[q]; mask |= (unsigned)(v & 1) << q | (unsigned)((v >> 1) & 1) << (8 + q); }
                first_sub = mask ? __builtin_ctz(mask) : 16;
                if (tid == 0) atomicAdd(&A.counters[29], 1ull);
                __syncthreads();                              // (stage_assign reuses the wave slots)
                again = true;
            }
        }
    }
    if (!again) break;
    }   // staging passes
    // ---- COMPACT CHAIN (see compact_walk): all candidates staged, one lane each
    bool compact_done = false, compact_corr = false;      // compact_corr: ... and the next block's lookahead correction is in corr_cd
    bool pp_posted = false;                                // ping-pong, first block: cW has been posted to the second block's workgroup
    float corr_cd[2] = {0.f, 0.f};
    long long tkc[3] = {0, 0, 0}, tkx[3] = {0, 0, 0};
    if constexpr (!DENSE) {
        const bool compact_try = single_pass_st && !prestage && !(A.compact_off & 1) && ncand_total >= cmin &&
                                 ncand_total <= kCompactMax && ncand_total <= SM.max_cand;
        if (compact_try) {
            int* wc = reinterpret_cast<int*>(smem + SM.wcnt_off);
            tkc[0] = clock64();
            const int nc = ncand_total;
            const int xtask = nc << xsh;
            if (wave == 0) {
                const bool ok = compact_walk<METHOD>(smem, SM, nc, ie, lane);
                if (lane == 0) { wc[8] = ok ? 1 : 0; __hip_atomic_store(&wc[12], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            } else {
                if (!xreg && A.b_next > 0) prefetch_cross_rows(smem, SM, A, nstaged, 1, wc + 12);     // (ragged next block: L2 only)
                if (!(A.compact_off & 2)) prefetch_next_gram(A, false, 1, wc + 12);             // the next block's staging becomes an L2 hit
            }
            __syncthreads();
            tkc[1] = clock64();
            const bool surprise = compact_surprise<METHOD>(smem, SM, nc, b, cand);
            if (__any(surprise) && lane == 0) wc[9] = 1;
            __syncthreads();
            compact_done = (wc[8] != 0) && (wc[9] == 0);
            tkc[2] = clock64();
            if (compact_done && xreg) {
                float* crossL = reinterpret_cast<float*>(smem + SM.rows_off);        // rows 0 .. nc-1 (the scratch row lies behind them)
                if (wave != 0) {
#pragma unroll
                    for (int k = 0; k < kXR; ++k) {
                        const int T = k * kXL + xlid;
                        if (T < xtask) *reinterpret_cast<xr_v4f*>(crossL + (T >> xsh) * B + 4 * (T & ((1 << xsh) - 1))) = xr[k];
                    }
                }
                tkx[0] = clock64();
                __syncthreads();
                tkx[1] = clock64();
                const CompactScratch CS(smem, SM);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int c = tid + q * kStepThreads;
                    if (c >= B) break;
                    float corr = 0.f;
                    for (int e0 = 0; e0 < nc; e0 += 8) {
                        float dd[8], g[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int e = e0 + u < nc ? e0 + u : nc - 1;
                            dd[u] = __int_as_float(CS.log[e].y);
                            g[u] = crossL[e * B + c];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) if (e0 + u < nc) corr = fmaf(dd[u], g[u], corr);     // (a marker that did not move: d = 0, an exact no-op)
                    }
                    corr_cd[q] = corr;
                    // (ping-pong: block 1's workgroup is waiting for exactly this)
                    if constexpr (kPP) { if (A.pp_cw_out != nullptr) pp_post_word(A.pp_cw_out + c, A.pp_tag, __float_as_uint(corr)); }
                }
                if constexpr (kPP) pp_posted = A.pp_cw_out != nullptr;
                compact_corr = true;
                tkx[2] = clock64();
            }
        }
    }
    const bool cross_lds = prestage && SM.has_cross;
    float4 corr_mine{0.f, 0.f, 0.f, 0.f};
    if (compact_corr) {
        // (nothing left for the other waves: the correction is done, the prefetches went out during the walk)
    } else if (stream_corr) {
        if (is_corr_helper(wave)) corr_mine = stream_corr_role(smem, SM, A);      // returns when the serial wave is done
        else if (wave >= 5) {                                           // waves 5, 6, 7
            const int* stop = reinterpret_cast<const int*>(smem + SM.wcnt_off) + 13;
            prefetch_cross_rows(smem, SM, A, nstaged, 5, stop);         // the helpers' loads become L2 hits
            prefetch_next_gram(A, prestage, 5, stop);
        }
    } else {
        if (cross_lds) {                                     // waves 1..7, while wave 0 runs the serial phase
            if (cross_dma) {
                dma_copy_to_lds(A.cross_next, reinterpret_cast<float*>(smem + SM.cross_off), B * B, 1);
                if (wave != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // landed before the barrier after the walk
            } else copy_cross_rows(smem, SM, A);
        }
        else prefetch_cross_rows(smem, SM, A, nstaged);      // waves 1..7: pull the rows into L2 for corr_phase at the end
        prefetch_next_gram(A, prestage);                     // waves 1..7: the next block's staging becomes an L2 hit
    }
    int* wcnt_s = reinterpret_cast<int*>(smem + SM.wcnt_off);
    long long tk3 = 0, tk4 = 0, tk5 = 0;
    int nrounds = 0, nslow = 0;
    if (wave == 0) {
    tk3 = clock64();

    // wave 0: lane l owns marker c = 64*s + l of sub-block s
    const short* slot_of = reinterpret_cast<const short*>(smem + SM.slot_off);
    const float* rows = reinterpret_cast<const float*>(smem + SM.rows_off);
    int2* evlog = reinterpret_cast<int2*>(smem + SM.log_off);
    float* bpark = bpark0;
    float* dpark = dpark0;
    const int nsub = (b + 63) / 64;
    const int nreps = P->nreps > 0 ? P->nreps : b;
    const bool lazy = (nreps == 1);     // single pass: corrections reach a sub-block when it becomes active
    RngKey key{P->seed_lo, P->seed_hi, P->iter, 0u};
    int nlog = 0;

    // The serial wave reads Gram rows from LDS ONLY (a value that may come from LDS or global compiles to
    // flat loads whose vmcnt(0) wait also drains the prefetch of the next sub-block).  A committed change
    // whose row was not staged is copied into a free slot first; when the slots are exhausted it goes to the
    // overflow row and is applied to the remaining sub-blocks at once instead of being logged.
    float* rows_w = reinterpret_cast<float*>(smem + SM.rows_off);
    auto fetch_row = [&](int ce, int slot) {
        const float* grow = A.gram + (int64_t)ce * b;
        for (int c0 = 0; c0 < B; c0 += 512) {          // 8 loads in flight per lane
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int col = c0 + 64 * u + lane; v[u] = grow[col < b ? col : 0]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int col = c0 + 64 * u + lane; if (col < B) rows_w[slot * B + col] = v[u]; }
        }
    };

    // ---- DENSE blocks (most markers of the block are candidates: Pi = 0, BayesA, the reference benchmark's setting):
    // speculation buys nothing -- every round commits exactly one marker -- so the wave walks the block sequentially.
    // Every lane evaluates ITS OWN marker against its own running rhs at every step (two float compares for the
    // decision, six operations for the new effect: no operand is broadcast); the step's marker is lane l, whose
    // alpha_old - alpha_new is broadcast with ONE v_readlane and applied to the running rhs of the whole block (two
    // registers per lane) with the marker's Gram row from LDS (all rows are staged; the read is issued a step ahead).
    // A lane's result is final at its own step: it keeps the rhs it was evaluated against and recomputes its update
    // after the walk.  Same arithmetic, same order, same results as the speculative rounds;
    // ~18 instructions per marker on a dependent chain of 11.
    bool dense_done = false;
    if constexpr (!kR) {
        if (dense_walk) {
            float lo[2], hi[2], il[2], da[2], ao[2], rhsq[2], rev[2], bex[2], kc1[2] = {0.f, 0.f}, kc0[2] = {0.f, 0.f};
            double zs[2];
            bool always = true;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = (64 * q + lane < B) ? 64 * q + lane : 0;
                il[q] = lpf[c]; bex[q] = lpf[B + c]; lo[q] = lpf[3 * B + c]; hi[q] = lpf[4 * B + c];
                if constexpr (DENSE) { kc1[q] = lpf[B + c]; kc0[q] = lpf[3 * B + c]; }
                zs[q] = lpd[c];
                rhsq[q] = rhs_lds[c]; ao[q] = acur[c]; rev[q] = rhsq[q];
                da[q] = lpf[2 * B + c] * ao[q];                                       // d * alpha_old (BayesABC.jl:36)
                always = always && (DENSE || lo[q] == hi[q]);                         // thresholds(): lo = hi <=> always included
            }
            // Pi = 0 / BayesA / RR-BLUP: every marker of the block is included whatever its rhs -- no decision on the chain
            const bool all_in = __all(always);
            if constexpr (DENSE) {
                // Rule D (uniform pi = 0): every marker is always included and its new effect is fmaf(kc1, x, kc0)
                if (B > 64) {
                    dense_section<0, true, true, true>(rows, B, b < 64 ? b : 64, lane, ie, lo[0], hi[0], il[0], da[0], ao[0], zs[0], rhsq[0], rhsq[1], rev[0], kc1[0], kc0[0]);
                    if (b > 64) dense_section<1, true, true, true>(rows + 64 * B, B, b - 64, lane, ie, lo[1], hi[1], il[1], da[1], ao[1], zs[1], rhsq[0], rhsq[1], rev[1], kc1[1], kc0[1]);
                } else dense_section<0, false, true, true>(rows, B, b, lane, ie, lo[0], hi[0], il[0], da[0], ao[0], zs[0], rhsq[0], rhsq[1], rev[0], kc1[0], kc0[0]);
            } else
            if (B > 64) {
                if (all_in) {
                    dense_section<0, true, true>(rows, B, b < 64 ? b : 64, lane, ie, lo[0], hi[0], il[0], da[0], ao[0], zs[0], rhsq[0], rhsq[1], rev[0]);
                    if (b > 64) dense_section<1, true, true>(rows + 64 * B, B, b - 64, lane, ie, lo[1], hi[1], il[1], da[1], ao[1], zs[1], rhsq[0], rhsq[1], rev[1]);
                } else {
                    dense_section<0, true, false>(rows, B, b < 64 ? b : 64, lane, ie, lo[0], hi[0], il[0], da[0], ao[0], zs[0], rhsq[0], rhsq[1], rev[0]);
                    if (b > 64) dense_section<1, true, false>(rows + 64 * B, B, b - 64, lane, ie, lo[1], hi[1], il[1], da[1], ao[1], zs[1], rhsq[0], rhsq[1], rev[1]);
                }
            } else {
                if (all_in) dense_section<0, false, true>(rows, B, b, lane, ie, lo[0], hi[0], il[0], da[0], ao[0], zs[0], rhsq[0], rhsq[1], rev[0]);
                else dense_section<0, false, false>(rows, B, b, lane, ie, lo[0], hi[0], il[0], da[0], ao[0], zs[0], rhsq[0], rhsq[1], rev[0]);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = 64 * q + lane;
                const bool inc = DENSE ? true : abc_included(rev[q], lo[q], hi[q]);
                const float an = DENSE ? fmaf(kc1[q], rev[q], kc0[q]) : dense_alpha_new(rev[q], da[q], ie, il[q], zs[q], inc);
                if (c < B) {
                    acur[c] = (c < b) ? an : 0.f; bpark0[c] = inc ? an : bex[q]; dpark0[c] = inc ? 1.f : 0.f;
                    rhs_lds[c] = (c < b) ? ao[q] - an : 0.f;                          // alpha_old - alpha_new, for the dense correction
                }
            }
            if (lane == 0) wcnt_s[14] = 1;
            nrounds += b;
            dense_done = true;
        }
    }

    const int s_first = (lazy && !dense_done) ? (first_sub < nsub ? first_sub : nsub) : 0;      // prefix skip (single pass only)

    // ---- SINGLE PASS (nreps = 1: the exact non-block chain; the hot path).  Everything comes from LDS; the log of
    // committed changes {row offset, D} lives in two VGPRs (lane e = entry e, written with v_writelane, read back with
    // v_readlane), so bringing a later sub-block up to date costs one LDS read per entry and no dependent second one.
    if (lazy && !dense_done && compact_done) {
        // the compact chain's results ARE the block's: effects, classes, and the change log {column, alpha_old - alpha_new} of
        // the effects that moved, in marker order -- published at once (the correction helpers consume it from here)
        int2* plog = reinterpret_cast<int2*>(smem + SM.log_off);
        const CompactScratch CS(smem, SM);
        const bool act = lane < ncand_total;
        const int2 e = CS.log[act ? lane : 0];
        const bool moved = act && (__int_as_float(e.y) != 0.f);
        const unsigned long long mm = __ballot(moved);
        if (act) {
            acur[e.x] = CS.an[lane];
            if constexpr (kR) dpark0[e.x] = (float)(CS.cls[lane] + 1);
        }
        if (moved) plog[__popcll(mm & ((1ull << lane) - 1ull))] = e;
        nrounds += ncand_total;
        if (lane == 0) {
            wcnt_s[11] = __popcll(mm);
            __hip_atomic_store(&wcnt_s[13], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else
    if (lazy && !dense_done) {
        int2* plog = reinterpret_cast<int2*>(smem + SM.log_off);
        int npub = 0;
        int log_off = 0;            // lane e: sl*B of entry e
        float log_D = 0.f;          // lane e: alpha_old - alpha_new of entry e
        // apply the logged changes (in commit order) to the rhs of the sub-blocks after `s` and empty the log: needed before
        // a change is applied eagerly (log full, or a row that only lives in the overflow slot) so that every rhs element
        // still sees its corrections in commit order
        auto flush_log = [&](int s) {
            for (int s2 = s + 1; s2 < nsub; ++s2) {
                const int c2 = 64 * s2 + lane;
                float r2 = rhs_lds[c2];
                for (int e = 0; e < nlog; ++e)
                    r2 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(log_D), e)),
                              rows[__builtin_amdgcn_readlane(log_off, e) + c2], r2);
                rhs_lds[c2] = r2;
            }
            nlog = 0;
        };
#pragma unroll 1
        for (int s = s_first; s < nsub; ++s) {
            const int c = 64 * s + lane;
            const bool valid = c < b;
            const int cl = valid ? c : 0;
            const float a_cur = acur[c];                  // (fixed while the marker is pending; a committed lane leaves `pending`)
            float rhs = rhs_lds[c];
            const int my_slot = slot_of[c];
            float c_lo = 0.f, c_hi = 0.f, c_il = 0.f, c_bex = 0.f, c_d = 0.f, c_thrx = 0.f, c_k1 = 0.f, c_k0 = 0.f;
            double c_zs = 0.0;
            double r_il1 = 0.0, r_il2 = 0.0, r_il3 = 0.0, r_zs1 = 0.0, r_zs2 = 0.0, r_zs3 = 0.0, r_T0 = 0.0, r_T1 = 0.0, r_T2 = 0.0;
            if constexpr (kR) {
                c_d = lpf[cl]; c_thrx = lpf[B + cl];
                r_il1 = lpd[0 * B + cl]; r_il2 = lpd[1 * B + cl]; r_il3 = lpd[2 * B + cl];
                r_zs1 = lpd[3 * B + cl]; r_zs2 = lpd[4 * B + cl]; r_zs3 = lpd[5 * B + cl];
                r_T0 = lpd[6 * B + cl]; r_T1 = lpd[7 * B + cl]; r_T2 = lpd[8 * B + cl];
            } else {
                c_il = lpf[cl]; c_bex = lpf[B + cl]; c_d = lpf[2 * B + cl]; c_lo = lpf[3 * B + cl]; c_hi = lpf[4 * B + cl];
                c_zs = lpd[cl];
                if constexpr (DENSE) { c_k1 = lpf[B + cl]; c_k0 = lpf[3 * B + cl]; }
            }
            // bring this sub-block up to date: the changes committed so far, in commit order (the same fmaf sequence
            // per element as an immediate update); 8 independent LDS reads in flight
            for (int e0 = 0; e0 < nlog; e0 += 8) {
                float gv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ee = e0 + u < nlog ? e0 + u : nlog - 1;
                    gv[u] = rows[__builtin_amdgcn_readlane(log_off, ee) + c];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u < nlog) rhs = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(log_D), e0 + u)), gv[u], rhs);
            }
            unsigned long long pending = __ballot(valid);
            const bool nz = a_cur != 0.f;
            // this sub-block's entries of the block's change list {local column, alpha_old - alpha_new} (marker order = commit
            // order; read by the correction helpers while it grows and by the final stores) collect in two VGPRs -- lane e =
            // the sub-block's e-th change, a lane wins at most once -- and go to LDS with ONE write when the sub-block is done
            // (a round is ~70 instructions of one wave at ~8 cycles each: the per-change LDS write under a lane mask was 8 of them)
            const int npub0 = npub;
            int pub_col = 0, pub_D = 0;
            // speculative rounds: every pending lane tests ITS marker against the current rhs (float compares only);
            // the first lane whose effect changes commits, the rest are re-tested after its Gram row corrected the rhs.
            // Lanes before the winner stay out of the model with the values parked for them; only the winner writes.
            while (true) {
                ++nrounds;
                bool inc = false, ev = false;
                float an = 0.f;
                if constexpr (kR) ev = nz || (fabsf(rhs) >= c_thrx);
                else {
                    inc = DENSE ? true : abc_included(rhs, c_lo, c_hi); ev = inc || nz;
                    // every lane's new effect BEFORE the vote: the six dependent operations run beside the vote's
                    // compare / ballot / find-first chain instead of after it (wasted only in a sub-block's last round)
                    if constexpr (DENSE) an = fmaf(c_k1, rhs, c_k0);         // Rule D (uniform pi = 0: inc is always true)
                    else an = abc_alpha_new(rhs, a_cur, c_d, ie, c_il, c_zs, inc);
                    asm volatile("" : "+v"(an));         // (keeps the compiler from sinking it below the vote's branch)
                }
                const unsigned long long m = __builtin_amdgcn_ballot_w64(ev) & pending;
                if (m == 0ull) break;                     // no further change in this sub-block
                const int k = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
                if constexpr (kR) {
                    bool sure = true;
                    int cls = bayesr_eval_thr(rhs, a_cur, ie, c_d, r_il1, r_il2, r_il3, r_zs1, r_zs2, r_zs3, r_T0, r_T1, r_T2, an, sure);
                    if (__builtin_amdgcn_readlane(sure ? 0 : 1, k)) {       // (practically never: s within 1e-6 of a class threshold)
                        BayesRMarker bm;
                        bm.load(A.prep_d, A.prep_f, p, j0 + cl, c_d, ie);   // full constants from global
                        cls = bm.evaluate(rhs, a_cur, ie, an);
                        ++nslow;
                    }
                    if (cls == 0) an = 0.f;
                    if (lane == k) { acur[c] = an; dpark[c] = (float)(cls + 1); }             // stored as class 1..4
                } else {
                    if (lane == k) acur[c] = an;          // (beta / delta follow from alpha at the end: derive_bd)
                }
                const float Dl = a_cur - an;
                pending = (k == 63) ? 0ull : (pending & ~((2ull << k) - 1ull));
                const float D = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Dl), k));
                if (D != 0.f) {
                    pub_col = jw_llvm_amdgcn_writelane_i32(64 * s + k, npub - npub0, pub_col);
                    pub_D = jw_llvm_amdgcn_writelane_i32(__float_as_int(D), npub - npub0, pub_D);
                    ++npub;
                    // rhs += D * G[ce][:] (BayesABC.jl:169,172).  The active sub-block is corrected in its register
                    // copy -- the only value the next round waits for.
                    int sl = __builtin_amdgcn_readlane(my_slot, k);
                    bool overflow = false;
                    if (sl < 0) {                                            // rare: not staged at entry
                        if (nstaged < SM.max_cand) sl = nstaged++; else { sl = SM.max_cand; overflow = true; }
                        fetch_row(64 * s + k, sl);
                        if (lane == 0) atomicAdd(&A.counters[1], 1ull);      // diagnostic
                    }
                    const int off = sl * B;
                    rhs = fmaf(D, rows[off + c], rhs);
                    if (overflow || nlog >= 64) {
                        flush_log(s);
                        for (int s2 = s + 1; s2 < nsub; ++s2) {
                            const int c2 = 64 * s2 + lane;
                            rhs_lds[c2] = fmaf(D, rows[off + c2], rhs_lds[c2]);
                        }
                    } else {
                        log_off = jw_llvm_amdgcn_writelane_i32(off, nlog, log_off);
                        log_D = __int_as_float(jw_llvm_amdgcn_writelane_i32(__float_as_int(D), nlog, __float_as_int(log_D)));
                        ++nlog;
                    }
                }
                if (pending == 0ull) break;
            }
            if (lane < npub - npub0) plog[npub0 + lane] = make_int2(pub_col, pub_D);
        }
        if (lane == 0) {
            wcnt_s[11] = npub;                                              // (read after the barrier)
            __hip_atomic_store(&wcnt_s[13], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);     // the helpers may finish
        }
    }

    for (int rep = 0; rep < ((dense_done || lazy) ? 0 : nreps); ++rep) {
        key.rep = (uint32_t)rep;
#pragma unroll 1
        for (int s = s_first; s < nsub; ++s) {
            const int c = 64 * s + lane;
            const bool valid = c < b;
            const int cl = valid ? c : 0;
            const int64_t j = j0 + cl;
            const uint32_t marker = P->marker0 + (uint32_t)j;
            unsigned long long pending = __ballot(valid);
            const float a_cur = acur[c];                  // (fixed while the marker is pending; a committed lane leaves `pending`)
            float rhs = rhs_lds[c];                       // register copy of the active sub-block's rhs
            const int my_slot = slot_of[c];
            if (lazy) {
                // bring this sub-block up to date: the changes committed so far, in commit order
                // (same fmaf sequence per entry as an immediate update)
                // two-phase chunks so the dependent LDS reads (log entry -> row element) are pipelined
                for (int e0 = 0; e0 < nlog; e0 += 8) {
                    int2 le[8];
                    float gv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) le[u] = evlog[e0 + u < nlog ? e0 + u : nlog - 1];
#pragma unroll
                    for (int u = 0; u < 8; ++u) gv[u] = rows[le[u].x * B + c];
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (e0 + u < nlog) rhs = fmaf(__int_as_float(le[u].y), gv[u], rhs);
                }
            }

            // the marker's constants: parked in LDS by the parallel phase (rep 0) or recomputed for a later repetition
            float c_lo = 0.f, c_hi = 0.f, c_il = 0.f, c_bex = 0.f, c_d = 0.f, c_thrx = 0.f, c_k1 = 0.f, c_k0 = 0.f;
            double c_zs = 0.0;
            BayesRMarker bm;
            double r_il1 = 0.0, r_il2 = 0.0, r_il3 = 0.0, r_zs1 = 0.0, r_zs2 = 0.0, r_zs3 = 0.0, r_T0 = 0.0, r_T1 = 0.0, r_T2 = 0.0;
            if (rep == 0) {
                if constexpr (kR) {
                    c_d = lpf[cl]; c_thrx = lpf[B + cl];
                    r_il1 = lpd[0 * B + cl]; r_il2 = lpd[1 * B + cl]; r_il3 = lpd[2 * B + cl];
                    r_zs1 = lpd[3 * B + cl]; r_zs2 = lpd[4 * B + cl]; r_zs3 = lpd[5 * B + cl];
                    r_T0 = lpd[6 * B + cl]; r_T1 = lpd[7 * B + cl]; r_T2 = lpd[8 * B + cl];
                } else {
                    c_il = lpf[cl]; c_bex = lpf[B + cl]; c_d = lpf[2 * B + cl]; c_lo = lpf[3 * B + cl]; c_hi = lpf[4 * B + cl];
                    c_zs = lpd[cl];
                    if constexpr (DENSE) { c_k1 = lpf[B + cl]; c_k0 = lpf[3 * B + cl]; }
                }
            } else {
                const float dj = A.xpx[j];
                const double u = draw_uniform(key, marker, 0u);
                const double z = draw_normal(key, marker, 0u);
                c_d = dj;
                if constexpr (kR) {
                    double pj[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) pj[k] = P->pi_mat ? P->pi_mat[4 * j + k] : P->pi4[k];
                    bm.prepare(dj, P->var_effect[0], pj, P->gamma, ie, u, z);
                    r_il1 = bm.invLhs[1]; r_il2 = bm.invLhs[2]; r_il3 = bm.invLhs[3];
                    r_zs1 = bm.zs[1]; r_zs2 = bm.zs[2]; r_zs3 = bm.zs[3];
                    r_T0 = bm.T[0]; r_T1 = bm.T[1]; r_T2 = bm.T[2];
                    c_thrx = (a_cur != 0.f) ? 0.f : bayesr_candidate_threshold(bm, ie);
                } else {
                    float var_j = P->var_effect[0];
                    if constexpr (METHOD == kBayesB) var_j = P->var_vec[j];
                    double pi_j = P->pi;
                    if (P->pi_vec) pi_j = P->pi_vec[j];
                    AbcMarker am;
                    am.prepare(dj, var_j, pi_j, ie, u, z);
                    am.thresholds(a_cur, ie, c_lo, c_hi);
                    c_il = am.invLhs; c_bex = am.beta_excl; c_zs = am.zs;
                    if constexpr (DENSE) am.rule_d(a_cur, ie, c_k1, c_k0);       // Rule D with this repetition's alpha_old and draw
                }
                // a marker that is not touched in this repetition gets the repetition's "out of the model" draw
                if (valid) { if constexpr (kR) dpark[c] = 1.f; else { bpark[c] = c_bex; dpark[c] = 0.f; } }
            }
            const bool nz = a_cur != 0.f;
            // speculative rounds: every pending lane tests ITS marker against the current rhs (float compares only);
            // the first lane whose effect changes commits, the rest are re-tested after its Gram row corrected the rhs.
            // Lanes before the winner stay out of the model with the values parked for them; only the winner writes.
            while (true) {
                ++nrounds;
                bool inc = false, ev = false;
                if constexpr (kR) ev = nz || (fabsf(rhs) >= c_thrx);
                else { inc = DENSE ? true : abc_included(rhs, c_lo, c_hi); ev = inc || nz; }
                const unsigned long long m = __ballot(ev && valid) & pending;
                if (m == 0ull) break;                     // no further change in this sub-block
                const int k = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
                float an = 0.f;
                if constexpr (kR) {
                    bool sure = true;
                    int cls = bayesr_eval_thr(rhs, a_cur, ie, c_d, r_il1, r_il2, r_il3, r_zs1, r_zs2, r_zs3, r_T0, r_T1, r_T2, an, sure);
                    if (__builtin_amdgcn_readlane(sure ? 0 : 1, k)) {       // (practically never: s within 1e-6 of a class threshold)
                        if (rep == 0) bm.load(A.prep_d, A.prep_f, p, j, c_d, ie);          // full constants from global
                        cls = bm.evaluate(rhs, a_cur, ie, an);
                        ++nslow;
                    }
                    if (cls == 0) an = 0.f;
                    if (lane == k) { acur[c] = an; dpark[c] = (float)(cls + 1); }             // stored as class 1..4
                } else {
                    if constexpr (DENSE) an = fmaf(c_k1, rhs, c_k0);         // Rule D (inc is always true)
                    else an = abc_alpha_new(rhs, a_cur, c_d, ie, c_il, c_zs, inc);
                    if (lane == k) { acur[c] = an; bpark[c] = inc ? an : c_bex; dpark[c] = inc ? 1.f : 0.f; }
                }
                const float Dl = a_cur - an;
                pending = (k == 63) ? 0ull : (pending & ~((2ull << k) - 1ull));
                const float D = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Dl), k));
                if (D != 0.f) {
                    // rhs += D * G[ce][:] (BayesABC.jl:169,172).  The active sub-block is corrected in its register
                    // copy -- the only value the next round waits for.
                    const int ce = 64 * s + k;
                    int sl = __builtin_amdgcn_readlane(my_slot, k);
                    bool overflow = false;
                    if (sl < 0) {                                            // rare: not staged at entry
                        if (nstaged < SM.max_cand) sl = nstaged++; else { sl = SM.max_cand; overflow = true; }
                        fetch_row(ce, sl);
                        if (lane == 0) atomicAdd(&A.counters[1], 1ull);      // diagnostic
                    }
                    const float* grow = rows + sl * B;
                    rhs = fmaf(D, grow[c], rhs);
                    if (lazy && !overflow) {
                        if (lane == 0) evlog[nlog] = make_int2(sl, __float_as_int(D));
                        ++nlog;
                    } else {
                        for (int s2 = lazy ? s + 1 : 0; s2 < nsub; ++s2) {
                            const int c2 = 64 * s2 + lane;
                            if (s2 != s) rhs_lds[c2] = fmaf(D, grow[c2], rhs_lds[c2]);
                        }
                    }
                }
                if (pending == 0ull) break;
            }
            rhs_lds[c] = rhs;
        }
    }
    // the net changes of this block as a compact list in LDS (nothing changed before the first candidate's sub-block);
    // single-pass sweeps: the published change log IS that list
    tk4 = clock64();
    if (!(lazy && !dense_done)) {
        int base = 0;
#pragma unroll 1
        for (int s = s_first; s < nsub; ++s) {
            const int c = 64 * s + lane;
            const bool changed = (c < b) && (astart[c] != acur[c]);
            const unsigned long long cm = __ballot(changed);
            if (changed) reinterpret_cast<int*>(smem + SM.log_off)[base + __popcll(cm & ((1ull << lane) - 1ull))] = c;
            base += __popcll(cm);
        }
        if (lane == 0) { wcnt_s[15] = base; wcnt_s[11] = -1; }
    }
    tk5 = clock64();
    }   // wave 0
    const long long tk5w = clock64();
    __syncthreads();
    const bool from_log = wcnt_s[11] >= 0;                  // single pass: {column, d} pairs published by the serial wave
    const int nfin = from_log ? wcnt_s[11] : wcnt_s[15];
    const long long tk6 = clock64();
    if (A.b_next > 0 && !stream_corr && cross_lds && wcnt_s[14] != 0) {
        // dense walk: every marker of the block is an entry (alpha_old - alpha_new left in rhs_lds; 0 = exact no-op), the
        // cross-Gram rows are in LDS: one thread per column of the next block, a chain of b fused multiply-adds in marker
        // order fed by broadcast reads of four changes and conflict-free reads of the rows
        const float* crossL = reinterpret_cast<const float*>(smem + SM.cross_off);
        if (tid < B) {
            float corr = 0.f;
            if (tid < A.b_next) {
                int e = 0;
#pragma unroll 1
                for (; e + 8 <= b; e += 8) {
                    const float4 d0 = *reinterpret_cast<const float4*>(rhs_lds + e), d1 = *reinterpret_cast<const float4*>(rhs_lds + e + 4);
                    float g[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) g[u] = crossL[(e + u) * B + tid];
                    corr = fmaf(d0.x, g[0], corr); corr = fmaf(d0.y, g[1], corr); corr = fmaf(d0.z, g[2], corr); corr = fmaf(d0.w, g[3], corr);
                    corr = fmaf(d1.x, g[4], corr); corr = fmaf(d1.y, g[5], corr); corr = fmaf(d1.z, g[6], corr); corr = fmaf(d1.w, g[7], corr);
                }
                for (; e < b; ++e) corr = fmaf(rhs_lds[e], crossL[e * B + tid], corr);
            }
            A.corr_out[tid] = corr;
        }
    } else if (A.b_next > 0 && !stream_corr) {
        if (from_log) {                                     // (small blocks with a sparse prior: corr_phase wants plain columns)
            const int2* plog = reinterpret_cast<const int2*>(smem + SM.log_off);
            int cols[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) cols[q] = (tid + q * kStepThreads < nfin) ? plog[tid + q * kStepThreads].x : 0;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q) if (tid + q * kStepThreads < nfin) reinterpret_cast<int*>(smem + SM.log_off)[tid + q * kStepThreads] = cols[q];
            __syncthreads();
        }
        corr_phase<1>(smem, SM, A, nfin, cross_lds);
    }
    const long long tk7 = clock64();
    // ping-pong, second block: where its changes go in the pair's merged list = how many block 0 had (posted when block 0's list
    // entries were acknowledged by the memory side -- long ago by now)
    int evb = ev_base;
    if constexpr (kPP) {
        if (A.pp_cnt_in != nullptr) {
            if (tid == 0) wcnt_s[10] = (int)pp_wait_word(A.pp_cnt_in, A.pp_tag, A.counters);
            __syncthreads();
            evb = wcnt_s[10];
        }
    }
    // ---- global stores LAST (nothing in this launch waits for them; a barrier after a global store waits for the store):
    // the change list for the next update role, alpha of the changed markers, beta / delta of the whole block
    if (compact_corr) {
#pragma unroll
        for (int q = 0; q < 2; ++q) if (tid + q * kStepThreads < B) A.corr_out[tid + q * kStepThreads] = corr_cd[q];
    } else
    if (stream_corr && is_corr_helper(wave)) {              // the lookahead correction accumulated by this helper lane
        const int col = (corr_helper_index(wave) * 64 + lane) * 4, bn = A.b_next;
        const float cv[4] = {corr_mine.x, corr_mine.y, corr_mine.z, corr_mine.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) if (col + i < B) A.corr_out[col + i] = (col + i < bn) ? cv[i] : 0.f;
    }
    {
        const int* fin = reinterpret_cast<const int*>(smem + SM.log_off);
        const bool pairs = from_log && (stream_corr || A.b_next <= 0);          // (else the list was turned into plain columns)
        int32_t* eidx; float* edel;
        if constexpr (GROUP) { eidx = A.ev_idx + evb; edel = A.ev_delta + evb; }
        else { eidx = A.ev_out->idx; edel = A.ev_out->delta[0]; }
        const int hb = GROUP ? evb : 0;                                         // (entries of the list in front of this block's)
        bool pp_first = false;                                                  // (ping-pong, first block: the second block's workgroup
        if constexpr (kPP) pp_first = A.pp_cnt_out != nullptr;                  //  reads these entries in THIS launch: write-through stores)
        for (int e = tid; e < nfin; e += kStepThreads) {
            const int ce = pairs ? fin[2 * e] : fin[e];
            const float d = astart[ce] - acur[ce];
            if (pp_first) {
                __hip_atomic_store(eidx + e, (int32_t)(j0 + ce), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(edel + e, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else { eidx[e] = (int32_t)(j0 + ce); edel[e] = d; }
            if (hb + e < 7) { A.ev_out->hidx[hb + e] = (int32_t)(j0 + ce); A.ev_out->hdelta[hb + e] = d; }
            A.alpha[j0 + ce] = acur[ce];
        }
        // single-pass BayesA/B/C: a marker is in the model iff its effect is nonzero, beta = the effect, else its
        // "excluded" draw (parked at entry) -- the serial wave only wrote alpha
        const bool derive_bd = !kR && from_log;
        for (int c = tid; c < b; c += kStepThreads) {
            if constexpr (kR) reinterpret_cast<int32_t*>(A.delta)[j0 + c] = (int32_t)dpark0[c];
            else if (derive_bd) {
                const float a = acur[c];
                A.beta[j0 + c] = (a != 0.f) ? a : bpark0[c];
                reinterpret_cast<float*>(A.delta)[j0 + c] = (a != 0.f) ? 1.f : 0.f;
            }
            else { A.beta[j0 + c] = bpark0[c]; reinterpret_cast<float*>(A.delta)[j0 + c] = dpark0[c]; }
        }
    }
    if constexpr (kPP) {
        if (A.pp_cnt_out != nullptr) {
            // the hand-over to the next block's workgroup: every store of this workgroup -- cW through corr_out on the paths that did
            // not post it themselves, the list entries -- has been acknowledged by the memory side (vmcnt counts a store out then)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (A.pp_cw_out != nullptr && !pp_posted)
                for (int c = tid; c < B; c += kStepThreads)
                    pp_post_word(A.pp_cw_out + c, A.pp_tag, __hip_atomic_load(reinterpret_cast<const unsigned*>(A.corr_out) + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (tid == 0) pp_post_word(A.pp_cnt_out, A.pp_tag, (unsigned)(evb + nfin));
        }
    }
    if (tid == 0) {
        if (!(kPP && A.pp_cnt_out != nullptr)) A.ev_out->count = (int32_t)((GROUP ? evb : 0) + nfin);      // (a split group's count: its last block's)
        atomicAdd(&A.counters[0], (unsigned long long)nfin);
        atomicAdd(&A.counters[2], (unsigned long long)(tk1 - tk0));      // phase cycle counts (diagnostics)
        atomicAdd(&A.counters[3], (unsigned long long)(tk2 - tk1));
        atomicAdd(&A.counters[4], (unsigned long long)(tk3 - tk2));
        atomicAdd(&A.counters[5], (unsigned long long)(tk4 - tk3));
        atomicAdd(&A.counters[6], (unsigned long long)(tk5 - tk4));
        atomicAdd(&A.counters[7], (unsigned long long)nrounds);
        if (nslow) atomicAdd(&A.counters[8], (unsigned long long)nslow);      // BayesR: rounds that needed the double-precision evaluation
        atomicAdd(&A.counters[9], (unsigned long long)(tk7 - tk6));           // the lookahead-correction phase
        atomicAdd(&A.counters[10], (unsigned long long)(tss[0] - tk2));       // staging: slot assignment | load issue | LDS stores
        atomicAdd(&A.counters[11], (unsigned long long)(tss[1] - tss[0]));
        atomicAdd(&A.counters[12], (unsigned long long)(tss[2] - tss[1]));
        atomicAdd(&A.counters[20], (unsigned long long)(clock64() - tk0));    // the whole role
        atomicAdd(&A.counters[21], (unsigned long long)(tk6 - tk5w));         // serial wave done -> every helper wave done
        if (tkc[0] != 0) {                                                    // compact chain: blocks tried / fallen back, walk / verification cycles
            atomicAdd(&A.counters[16], 1ull);
            if (!compact_done) atomicAdd(&A.counters[17], 1ull);
            atomicAdd(&A.counters[18], (unsigned long long)(tkc[1] - tkc[0]));
            atomicAdd(&A.counters[19], (unsigned long long)(tkc[2] - tkc[1]));
            if (tkx[0] != 0) {
                atomicAdd(&A.counters[22], (unsigned long long)(tkx[1] - tkc[2]));      // cross-Gram pieces to LDS (+ barrier)
                atomicAdd(&A.counters[23], (unsigned long long)(tkx[2] - tkx[1]));      // the correction chain
            }
        }
    }
    return evb + nfin;                                     // (grouped launches: the length of the merged list behind this block)
}


}  // namespace jw
