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
// Launch k depends only on launch k-1 (stream order): no in-kernel inter-workgroup communication.
#pragma once
#include "kernels.hpp"

namespace jw {

constexpr int kStepThreads = 512;
constexpr int kRowsBytes = 96 * 1024;          // LDS budget for staged Gram rows (sampler role)

// Dynamic-LDS carve of one step workgroup (bytes); B = block size, NT = traits.
struct StepSmem {
    int B, NT, max_cand;
    int rhs_off, acur_off, bcur_off, dcur_off, slot_off, cand_off, wcnt_off, rows_off, bytes;
    __host__ __device__ StepSmem(int B_, int NT_) : B(B_), NT(NT_)
    {
        max_cand = kRowsBytes / (4 * B);
        if (max_cand > B) max_cand = B;
        rhs_off  = 0;                               // float [NT][B]  running block RHS
        acur_off = rhs_off + NT * B * 4;            // float [NT][B]  current alpha
        bcur_off = acur_off + NT * B * 4;           // float [NT][B]  current beta   (multi-trait)
        dcur_off = bcur_off + NT * B * 4;           // float [NT][B]  current delta  (multi-trait)
        slot_off = dcur_off + NT * B * 4;           // int16 [B]      LDS slot of a marker's Gram row, -1 = not staged
        cand_off = slot_off + B * 2;                // int16 [max_cand]
        wcnt_off = (cand_off + max_cand * 2 + 15) / 16 * 16;   // int [16]
        rows_off = wcnt_off + 64;                   // float [max_cand][B]
        const int samp = rows_off + max_cand * B * 4;
        const int red = kRowGroupSlices * kColChunk * NT * 8;   // update role: double [8][64][NT]
        bytes = samp > red ? samp : red;
    }
};

// ---------------------------------------------------------------------------------------------
// UPDATE/PARTIAL role
// ---------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void update_role(char* smem, int rg, int g,
                                            const float* __restrict__ X, int64_t ld,
                                            const float* __restrict__ r_in, float* __restrict__ r_out,
                                            const Events* __restrict__ ev,
                                            int64_t j0, int b, int nslices, int nrg, int ncg,
                                            double* __restrict__ partials, int bstride)
{
    typedef double RedT[kColChunk][NT];
    RedT* red = reinterpret_cast<RedT*>(smem);                 // [kRowGroupSlices][kColChunk][NT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slice = rg * kRowGroupSlices + wave;
    const bool active = slice < nslices;
    // an inactive wave (slice beyond the matrix) aliases slice 0 for addressing and contributes 0
    const int64_t row = (int64_t)(active ? slice : 0) * kSliceRows + lane * 4;
    const int ncols = (b > g) ? (b - g + ncg - 1) / ncg : 0;

    // Loads are unconditional from clamped, always-valid addresses: a select between a load and a
    // constant makes hipcc pick between pointers and emit flat/scratch accesses.
    const float* xcol = X + (j0 + (ncols > 0 ? g : 0)) * ld + row;
    const int64_t cstride = (int64_t)ncg * ld;
    const int nc1 = ncols > 0 ? ncols - 1 : 0;
    auto load_batch = [&](float4 (&dst)[kU], int ib) {
#pragma unroll
        for (int u = 0; u < kU; ++u)
            dst[u] = *reinterpret_cast<const float4*>(xcol + (ib + u < ncols ? ib + u : nc1) * cstride);
    };

    // (1) the first batch of column loads does not depend on r: issue it before the update.
    float4 xa[kU], xb[kU];
    load_batch(xa, 0);

    // (2) sparse exit update: sequential fmaf in marker order, bit-identical to the oracle's per-marker
    //     axpy sequence.  Every column group recomputes it (reads r_in only); group 0 stores r_out.
    float4 rv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) rv[t] = *reinterpret_cast<const float4*>(r_in + t * ld + row);
    const int ne = ev->count;
#pragma unroll 8
    for (int e = 0; e < ne; ++e) {
        const float4 x = *reinterpret_cast<const float4*>(X + (int64_t)ev->idx[e] * ld + row);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float d = ev->delta[t][e];
            rv[t].x = fmaf(d, x.x, rv[t].x); rv[t].y = fmaf(d, x.y, rv[t].y);
            rv[t].z = fmaf(d, x.z, rv[t].z); rv[t].w = fmaf(d, x.w, rv[t].w);
        }
    }
    if (active && g == 0)
#pragma unroll
        for (int t = 0; t < NT; ++t) *reinterpret_cast<float4*>(r_out + t * ld + row) = rv[t];
    if (ncols == 0) return;
    const float keep = active ? 1.f : 0.f;
    double rd[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        rd[t][0] = rv[t].x * keep; rd[t][1] = rv[t].y * keep; rd[t][2] = rv[t].z * keep; rd[t][3] = rv[t].w * keep;
    }

    // (3) partial block RHS.
    auto consume = [&](const float4 (&xv)[kU], int ib, int i0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            double acc[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                acc[u] = (double)xv[u].x * rd[t][0];
                acc[u] = fma((double)xv[u].y, rd[t][1], acc[u]);
                acc[u] = fma((double)xv[u].z, rd[t][2], acc[u]);
                acc[u] = fma((double)xv[u].w, rd[t][3], acc[u]);
            }
            const double s = butterfly8(acc, lane);
            const int u = lane >> 3;                       // column of this 8-lane group
            if ((lane & 7) == 0 && ib + u < ncols) red[wave][ib + u - i0][t] = s;
        }
    };

    for (int i0 = 0; i0 < ncols; i0 += kColChunk) {
        const int iend = (i0 + kColChunk < ncols) ? i0 + kColChunk : ncols;
        // two batches per trip so both register sets are statically indexed
        for (int ib = i0; ib < iend; ib += 2 * kU) {
            if (ib + kU < ncols) load_batch(xb, ib + kU);
            consume(xa, ib, i0);
            if (ib + 2 * kU < ncols) load_batch(xa, ib + 2 * kU);
            if (ib + kU < iend) consume(xb, ib + kU, i0);
        }
        __syncthreads();
        for (int q = tid; q < (iend - i0) * NT; q += kStepThreads) {
            const int i = q / NT, t = q - i * NT;
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < kRowGroupSlices; ++w) s += red[w][i][t];
            const int c = g + (i0 + i) * ncg;
            partials[((int64_t)t * nrg + rg) * bstride + c] = s;
        }
        __syncthreads();
    }
}

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
    int64_t j0_prev;              // first column of the previous block
    int bsz;                      // nominal block size (LDS strides)
    const float* xpx;
    const float* gram;            // b x b, this block
    const float* cross;           // bprev x b: X_prev' X_this (row = marker of the previous block)
    const double* prep_d; const float* prep_f;
    float* alpha; float* beta; void* delta;
    const Events* ev_prev;        // changes of the previous block (for the lookahead correction)
    Events* ev_out;
    unsigned long long* counters;
};

template <int NT>
__device__ __forceinline__ void sampler_front(char* smem, const StepSmem& SM, const SamplerArgs& A)
{
    const int B = SM.B;
    float* rhs_lds = reinterpret_cast<float*>(smem + SM.rhs_off);
    float* acur = reinterpret_cast<float*>(smem + SM.acur_off);
    const int tid = threadIdx.x;
    const int b = A.b;
    // rhs_b[c] = sum over row groups (fp64, fixed order), rounded once to fp32; then the lookahead
    // correction for the previous block's changed markers, in marker order.
    const int ne = A.ev_prev->count;
    for (int c = tid; c < B; c += kStepThreads) {
        const int cc = c < b ? c : 0;
        float rv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            double s = 0.0;
            for (int rg = 0; rg < A.nrg; ++rg) s += A.partials[((int64_t)t * A.nrg + rg) * A.bstride + cc];
            rv[t] = (float)s;
        }
#pragma unroll 8
        for (int e = 0; e < ne; ++e) {
            const float g = A.cross[(int64_t)(A.ev_prev->idx[e] - A.j0_prev) * b + cc];
#pragma unroll
            for (int t = 0; t < NT; ++t) rv[t] = fmaf(A.ev_prev->delta[t][e], g, rv[t]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            rhs_lds[t * B + c] = rv[t];
            acur[t * B + c] = (c < b) ? A.alpha[(int64_t)t * A.p + A.j0 + cc] : 0.f;
        }
    }
    __syncthreads();
}

// Stage the Gram rows of the candidate markers (cand[q] for marker c = tid + q*kStepThreads) in LDS.
__device__ __forceinline__ void stage_rows(char* smem, const StepSmem& SM, const SamplerArgs& A, const bool (&cand)[2])
{
    const int B = SM.B;
    short* slot_of = reinterpret_cast<short*>(smem + SM.slot_off);
    short* cand_list = reinterpret_cast<short*>(smem + SM.cand_off);
    int* wcnt = reinterpret_cast<int*>(smem + SM.wcnt_off);
    float* rows = reinterpret_cast<float*>(smem + SM.rows_off);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = A.b;
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
    const int ncand = base < SM.max_cand ? base : SM.max_cand;
    for (int s = wave; s < ncand; s += kStepThreads / 64) {
        const float* grow = A.gram + (int64_t)cand_list[s] * b;
        for (int c = lane; c < B; c += 64) rows[s * B + c] = grow[c < b ? c : 0];
    }
    __syncthreads();
}

// rhs[t][:] += D[t] * G[ce][:]  for the committed marker ce (BayesABC.jl:169,172); one wave.
template <int NT>
__device__ __forceinline__ void apply_gram_row(char* smem, const StepSmem& SM, const SamplerArgs& A, int ce,
                                               const float (&D)[NT], int lane)
{
    const int B = SM.B, b = A.b;
    float* rhs_lds = reinterpret_cast<float*>(smem + SM.rhs_off);
    const short* slot_of = reinterpret_cast<const short*>(smem + SM.slot_off);
    const float* rows = reinterpret_cast<const float*>(smem + SM.rows_off);
    const int sl = slot_of[ce];
    const float* grow = A.gram + (int64_t)ce * b;                 // symmetric: row = column
    for (int c2 = lane; c2 < B; c2 += 64) {
        const float g = (sl >= 0) ? rows[sl * B + c2] : grow[c2 < b ? c2 : 0];
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (D[t] != 0.f) rhs_lds[t * B + c2] = fmaf(D[t], g, rhs_lds[t * B + c2]);
    }
}

// ---------------------------------------------------------------------------------------------
// SAMPLER role, single trait.  METHOD in {kBayesC, kBayesB, kBayesR}.
// ---------------------------------------------------------------------------------------------
template <int METHOD>
__device__ __forceinline__ void sampler_role_st(char* smem, const SamplerArgs& A)
{
    const StepSmem SM(A.bsz, 1);
    const int B = SM.B;
    const DevParams* P = A.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = A.b;
    const int64_t j0 = A.j0, p = A.p;
    const float ie = 1.0f / P->vare[0];
    float* rhs_lds = reinterpret_cast<float*>(smem + SM.rhs_off);
    float* acur = reinterpret_cast<float*>(smem + SM.acur_off);

    sampler_front<1>(smem, SM, A);

    // candidates: markers whose effect changes if evaluated against the entry rhs (all threads)
    bool cand[2] = {false, false};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = tid + q * kStepThreads;
        if (c < b) {
            const int64_t j = j0 + c;
            const float a0 = acur[c];
            if (a0 != 0.f) cand[q] = true;
            else if constexpr (METHOD == kBayesR) {
                BayesRMarker bm; float an;
                bm.load(A.prep_d, A.prep_f, p, j, A.xpx[j], ie);
                cand[q] = bm.evaluate(rhs_lds[c], 0.f, ie, an) != 0;
            } else {
                AbcMarker am; float gh;
                am.load(A.prep_d, A.prep_f, p, j, A.xpx[j]);
                cand[q] = am.evaluate(rhs_lds[c], 0.f, ie, gh);
            }
        }
    }
    stage_rows(smem, SM, A, cand);
    if (wave != 0) return;

    // wave 0: lane l owns marker c = 64*s + l of sub-block s
    float* delta_f = reinterpret_cast<float*>(A.delta);
    int32_t* delta_i = reinterpret_cast<int32_t*>(A.delta);
    const int nsub = (b + 63) / 64;
    const int nreps = P->nreps > 0 ? P->nreps : b;
    RngKey key{P->seed_lo, P->seed_hi, P->iter, 0u};

    for (int rep = 0; rep < nreps; ++rep) {
        key.rep = (uint32_t)rep;
#pragma unroll 1
        for (int s = 0; s < nsub; ++s) {
            const int c = 64 * s + lane;
            const bool valid = c < b;
            const int64_t j = j0 + (valid ? c : 0);
            const uint32_t marker = P->marker0 + (uint32_t)j;
            unsigned long long pending = __ballot(valid);
            const float dj = A.xpx[j];
            float a_cur = acur[c];
            float b_out = 0.f, d_out = 0.f;

            AbcMarker am; BayesRMarker bm;
            if (rep == 0) {
                if constexpr (METHOD == kBayesR) bm.load(A.prep_d, A.prep_f, p, j, dj, ie);
                else am.load(A.prep_d, A.prep_f, p, j, dj);
            } else {
                const double u = draw_uniform(key, marker, 0u);
                const double z = draw_normal(key, marker, 0u);
                if constexpr (METHOD == kBayesR) {
                    double pj[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) pj[k] = P->pi_mat ? P->pi_mat[4 * j + k] : P->pi4[k];
                    bm.prepare(dj, P->var_effect[0], pj, P->gamma, ie, u, z);
                } else {
                    float var_j = P->var_effect[0];
                    if constexpr (METHOD == kBayesB) var_j = P->var_vec[j];
                    double pi_j = P->pi;
                    if (P->pi_vec) pi_j = P->pi_vec[j];
                    am.prepare(dj, var_j, pi_j, ie, u, z);
                }
            }
            // speculative rounds
            while (true) {
                const float rhs = rhs_lds[c];
                bool is_event = false, incl = false;
                float a_new = 0.f, gHat = 0.f;
                int cls = 0;
                const bool live = valid && ((pending >> lane) & 1ull);
                if (live) {
                    if constexpr (METHOD == kBayesR) {
                        cls = bm.evaluate(rhs, a_cur, ie, a_new);
                        is_event = (cls != 0) || (a_cur != 0.f);
                    } else {
                        incl = am.evaluate(rhs, a_cur, ie, gHat);
                        is_event = incl || (a_cur != 0.f);
                    }
                }
                const unsigned long long m = __ballot(is_event) & pending;
                const int k = m ? __builtin_ctzll(m) : 64;
                // lanes before k (and k itself) are final with the values just computed
                float Dl = 0.f;
                if (live && lane <= k) {
                    if constexpr (METHOD == kBayesR) {
                        d_out = (float)(cls + 1);                          // stored as class 1..4
                        const float an = (cls == 0) ? 0.f : a_new;
                        Dl = a_cur - an;
                        a_cur = an;
                    } else {
                        if (incl) { const float an = am.alpha_incl(gHat); d_out = 1.f; b_out = an; Dl = a_cur - an; a_cur = an; }
                        else      { d_out = 0.f; b_out = am.beta_excl; Dl = a_cur; a_cur = 0.f; }
                    }
                }
                if (k == 64) break;
                pending = (k == 63) ? 0ull : (pending & ~((2ull << k) - 1ull));
                const float D[1] = {__shfl(Dl, k, 64)};
                if (D[0] != 0.f) apply_gram_row<1>(smem, SM, A, 64 * s + k, D, lane);
                if (pending == 0ull) break;
            }
            acur[c] = a_cur;
            if (valid) {
                if constexpr (METHOD == kBayesR) delta_i[j] = (int32_t)d_out;
                else { A.beta[j] = b_out; delta_f[j] = d_out; }
            }
        }
    }

    // write back alpha and the net changes of this block
    int base = 0;
#pragma unroll 1
    for (int s = 0; s < nsub; ++s) {
        const int c = 64 * s + lane;
        const bool valid = c < b;
        const int64_t j = j0 + (valid ? c : 0);
        const float a_start = A.alpha[j];
        const float a_fin = acur[c];
        const bool changed = valid && (a_start != a_fin);
        const unsigned long long cm = __ballot(changed);
        if (changed) {
            const int pos = base + __popcll(cm & ((1ull << lane) - 1ull));
            A.ev_out->idx[pos] = (int32_t)j;
            A.ev_out->delta[0][pos] = a_start - a_fin;
            A.alpha[j] = a_fin;
        }
        base += __popcll(cm);
    }
    if (lane == 0) {
        A.ev_out->count = base;
        atomicAdd(&A.counters[0], (unsigned long long)base);
    }
}

// ---------------------------------------------------------------------------------------------
// SAMPLER role, multi-trait BayesC Gibbs sampler I (MTBayesABC.jl:57-127, block form :243-333)
// ---------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void sampler_role_mt1(char* smem, const SamplerArgs& A)
{
    const StepSmem SM(A.bsz, NT);
    const int B = SM.B;
    const DevParams* P = A.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = A.b;
    const int64_t j0 = A.j0, p = A.p;
    float* rhs_lds = reinterpret_cast<float*>(smem + SM.rhs_off);
    float* acur = reinterpret_cast<float*>(smem + SM.acur_off);
    float* bcur = reinterpret_cast<float*>(smem + SM.bcur_off);
    float* dcur = reinterpret_cast<float*>(smem + SM.dcur_off);
    float* delta = reinterpret_cast<float*>(A.delta);

    sampler_front<NT>(smem, SM, A);
    for (int c = tid; c < B; c += kStepThreads) {
        const int cc = c < b ? c : 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            bcur[t * B + c] = A.beta[(int64_t)t * p + j0 + cc];
            dcur[t * B + c] = delta[(int64_t)t * p + j0 + cc];
        }
    }
    // candidates: markers already in the model for some trait (their effects always change)
    bool cand[2] = {false, false};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = tid + q * kStepThreads;
        if (c < b)
#pragma unroll
            for (int t = 0; t < NT; ++t) cand[q] = cand[q] || (acur[t * B + c] != 0.f);
    }
    stage_rows(smem, SM, A, cand);      // (its barriers also publish bcur/dcur)
    if (wave != 0) return;

    float Rinv[NT][NT], Ginv[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c) { Rinv[a][c] = P->Rinv[a * NT + c]; Ginv[a][c] = P->Ginv[a * NT + c]; }

    const int nsub = (b + 63) / 64;
    const int nreps = P->nreps > 0 ? P->nreps : b;
    RngKey key{P->seed_lo, P->seed_hi, P->iter, 0u};

    for (int rep = 0; rep < nreps; ++rep) {
        key.rep = (uint32_t)rep;
#pragma unroll 1
        for (int s = 0; s < nsub; ++s) {
            const int c = 64 * s + lane;
            const bool valid = c < b;
            const int64_t j = j0 + (valid ? c : 0);
            const uint32_t marker = P->marker0 + (uint32_t)j;
            unsigned long long pending = __ballot(valid);
            const float dj = A.xpx[j];
            double thr[NT], z[NT];
            float a_cur[NT], b_cur[NT], d_cur[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                a_cur[t] = acur[t * B + c]; b_cur[t] = bcur[t * B + c]; d_cur[t] = dcur[t * B + c];
                if (rep == 0) { thr[t] = A.prep_d[(int64_t)t * p + j]; z[t] = A.prep_d[(int64_t)(NT + t) * p + j]; }
                else {
                    const double u = draw_uniform(key, marker, (uint32_t)t);
                    thr[t] = log((1.0 - u) / u);
                    z[t] = draw_normal(key, marker, (uint32_t)t);
                }
            }
            while (true) {
                const bool live = valid && ((pending >> lane) & 1ull);
                float an[NT], bn[NT], dn[NT], Dl[NT];
                bool is_event = false;
#pragma unroll
                for (int t = 0; t < NT; ++t) { an[t] = a_cur[t]; bn[t] = b_cur[t]; dn[t] = d_cur[t]; Dl[t] = 0.f; }
                if (live) {
                    float w[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) w[t] = rhs_lds[t * B + c] + dj * a_cur[t];           // :82
#pragma unroll
                    for (int k = 0; k < NT; ++k) {                                                  // :85
                        const float Ginv11 = Ginv[k][k];
                        const float C11 = Ginv11 + Rinv[k][k] * dj;                                 // :89
                        float rhs0 = 0.f, c12b = 0.f, wR = 0.f;
#pragma unroll
                        for (int m = 0; m < NT; ++m) {
                            wR = wR + w[m] * Rinv[m][k];
                            if (m == k) continue;
                            const float C12m = Ginv[k][m] + (dj * dn[m]) * Rinv[k][m];              // :90
                            rhs0 = rhs0 + Ginv[k][m] * bn[m];
                            c12b = c12b + C12m * bn[m];
                        }
                        rhs0 = -rhs0;                                                               // :93
                        const float invLhs0 = 1.0f / Ginv11;
                        const float gHat0 = rhs0 * invLhs0;
                        const float invLhs1 = 1.0f / C11;
                        const float rhs1 = wR - c12b;                                               // :96
                        const float gHat1 = rhs1 * invLhs1;
                        unsigned s0 = 0u;
#pragma unroll
                        for (int m = 0; m < NT; ++m) if (m != k && dn[m] != 0.f) s0 |= 1u << m;
                        const unsigned s1 = s0 | (1u << k);
                        const float in0 = logf_via_double(Ginv11) - (gHat0 * gHat0) * Ginv11;       // :104
                        const float in1 = logf_via_double(C11) - (gHat1 * gHat1) * C11;             // :105
                        const double* lpr = P->log_prior;
                        const double logDelta0 = -0.5 * (double)in0 + lpr[s0];
                        const double logDelta1 = -0.5 * (double)in1 + lpr[s1];
                        if ((logDelta0 - logDelta1) < thr[k]) {                                     // :107-111
                            dn[k] = 1.f;
                            bn[k] = (float)((double)gHat1 + z[k] * (double)sqrtf(invLhs1));
                            Dl[k] = an[k] - bn[k];
                            an[k] = bn[k];
                        } else {                                                                    // :112-119
                            bn[k] = (float)((double)gHat0 + z[k] * (double)sqrtf(invLhs0));
                            dn[k] = 0.f;
                            Dl[k] = an[k];
                            an[k] = 0.f;
                        }
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) is_event = is_event || (Dl[t] != 0.f);
                }
                const unsigned long long m = __ballot(is_event) & pending;
                const int k = m ? __builtin_ctzll(m) : 64;
                if (live && lane <= k) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) { a_cur[t] = an[t]; b_cur[t] = bn[t]; d_cur[t] = dn[t]; }
                }
                if (k == 64) break;
                pending = (k == 63) ? 0ull : (pending & ~((2ull << k) - 1ull));
                float D[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) D[t] = __shfl(Dl[t], k, 64);
                apply_gram_row<NT>(smem, SM, A, 64 * s + k, D, lane);                               // :311,317
                if (pending == 0ull) break;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) { acur[t * B + c] = a_cur[t]; bcur[t * B + c] = b_cur[t]; dcur[t * B + c] = d_cur[t]; }
        }
    }

    int base = 0;
#pragma unroll 1
    for (int s = 0; s < nsub; ++s) {
        const int c = 64 * s + lane;
        const bool valid = c < b;
        const int64_t j = j0 + (valid ? c : 0);
        bool changed = false;
        float dd[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float a0 = A.alpha[(int64_t)t * p + j];
            dd[t] = a0 - acur[t * B + c];
            changed = changed || (valid && a0 != acur[t * B + c]);
        }
        if (valid) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                A.alpha[(int64_t)t * p + j] = acur[t * B + c];
                A.beta[(int64_t)t * p + j]  = bcur[t * B + c];
                delta[(int64_t)t * p + j]   = dcur[t * B + c];
            }
        }
        const unsigned long long cm = __ballot(changed);
        if (changed) {
            const int pos = base + __popcll(cm & ((1ull << lane) - 1ull));
            A.ev_out->idx[pos] = (int32_t)j;
#pragma unroll
            for (int t = 0; t < NT; ++t) A.ev_out->delta[t][pos] = dd[t];
        }
        base += __popcll(cm);
    }
    if (lane == 0) {
        A.ev_out->count = base;
        atomicAdd(&A.counters[0], (unsigned long long)base);
    }
}

// ---------------------------------------------------------------------------------------------
// The fused step.  grid = 1 + nrg*ncg, block = 512.
// ---------------------------------------------------------------------------------------------
struct UpdateArgs {
    const float* X; int64_t ld;
    const float* r_in; float* r_out;
    const Events* ev;             // changes to apply (block k-2)
    int64_t j0; int b;            // block whose partial RHS is formed (b = 0: none)
    int nslices, nrg, ncg;
    double* partials; int bstride;
};

template <int METHOD, int NT>
__global__ __launch_bounds__(kStepThreads) void k_block_step(UpdateArgs U, SamplerArgs S, int do_sample)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (blockIdx.x == 0) {
        if (!do_sample) return;
        if constexpr (METHOD == kMTBayesC1) sampler_role_mt1<NT>(smem, S);
        else sampler_role_st<METHOD>(smem, S);
        return;
    }
    const int w = blockIdx.x - 1;
    update_role<NT>(smem, w % U.nrg, w / U.nrg, U.X, U.ld, U.r_in, U.r_out, U.ev, U.j0, U.b,
                    U.nslices, U.nrg, U.ncg, U.partials, U.bstride);
}

// Cross-Gram of consecutive blocks, exact (fp64-accumulated): C[a][c] = x_{jp+a}' x_{j0+c}.
// grid = (bsize, nblocks-1), block = 256; workgroup (a, i) writes row a of cross block i+1.
__global__ __launch_bounds__(256) void k_cross_f64(const float* __restrict__ X, int64_t ld, int64_t p, int bsize,
                                                   float* __restrict__ cross)
{
    const int64_t blk = (int64_t)blockIdx.y + 1;
    const int64_t j0 = blk * bsize, jp = j0 - bsize;
    const int b = (int)((j0 + bsize <= p) ? bsize : (p - j0));
    const int a = blockIdx.x;
    float* C = cross + blk * (int64_t)bsize * bsize;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* xa = X + (jp + a) * ld;
    for (int c = wave; c < b; c += 4) {
        const float* xc = X + (j0 + c) * ld;
        double s = 0.0;
        for (int64_t i = (int64_t)lane * 4; i < ld; i += 256) {
            const float4 qa = *reinterpret_cast<const float4*>(xa + i);
            const float4 qc = *reinterpret_cast<const float4*>(xc + i);
            s = fma((double)qa.x, (double)qc.x, s);
            s = fma((double)qa.y, (double)qc.y, s);
            s = fma((double)qa.z, (double)qc.z, s);
            s = fma((double)qa.w, (double)qc.w, s);
        }
        s = wave_sum(s);
        if (lane == 0) C[(int64_t)a * b + c] = (float)s;
    }
}

}  // namespace jw
