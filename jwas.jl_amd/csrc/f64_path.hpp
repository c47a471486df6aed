// f64_path.hpp -- the Float64 marker sweep (runMCMC(double_precision=true), JWAS.jl:349-366, readgenotypes.jl:298,345).
//
// In the reference double_precision=true makes EVERYTHING Float64: genotypes, residual, effects, x'x, the scalar kernels'
// arithmetic.  This file is that mode on the device: the same exact block form of the single-site chain as the Float32 path
// (x_j'(r - x_k D_k) = rhs_j - G_jk D_k; BayesABC.jl:118-188, BayesR.jl:111-193, MTBayesABC.jl:243-333) with every
// quantity in double, written for clarity first:
//
//   per block k (b <= 1024 markers, any partition), two stream-ordered launches:
//     k64_update_partial   grid = (256-row slices, column groups): apply block k-1's changes to the slice of r (registers;
//                          column group 0 writes the other residual buffer -- the two alternate, so no group ever reads a
//                          row another is writing), then the slice's partial right-hand side X_k[rows,:]' (R^-1 r) for its
//                          share of the block's columns -- X is read once per sweep (+ once per changed column);
//     k64_sample           one workgroup: rhs = sum of the slice partials (fixed order); the block's Gram staged in LDS when
//                          it fits (b <= 128: <= 128 KB of doubles), else its rows read from L2; then one wave runs the
//                          block's single-site chain by speculative parallel evaluation (all 64 lanes test their marker
//                          against the current rhs; the first lane whose effect changes commits, its Gram row corrects the
//                          rhs, the rest are re-tested), 64-marker sub-block after sub-block, the markers' running state in
//                          LDS, the draws fixed by the counter RNG -- the Float32 path's scheme without its fast paths.
//   independent_blocks (BayesABC.jl:190-255): every block's partial sums from the SAME residual (no apply between the
//   launches), every block sampled, then k64_finish reconciles r += sum_b X_b (alpha_old - alpha_new) in (block, marker) order.
//
// There is no lookahead here (launch k+1 starts when block k's sampler is done): the Float64 mode is the reference's
// "more digits" switch, not its throughput mode -- ~2 launches of a few microseconds per block on top of streaming
// 8 n p bytes.  Methods: single-trait BayesA/B/C (RR-BLUP, BayesL via the host), BayesR, multi-trait sampler I; dense
// storage; residual weights (x'R^-1 x, X_b'R^-1 X_b, X_b'R^-1 r, r'R^-1 r); uniform blocks of any size <= 1024 and explicit
// (ragged) partitions with within-block repetitions (fast_blocks); block size x traits <= 2048.  Arithmetic: operation for
// operation the scalar kernels of the reference with T = Float64 (bayesabc_update_marker! BayesABC.jl:24-58, BayesR! :56-96,
// _MTBayesABC_samplerI! :57-127); inner products are plain double sums (order: 4 x 64 lanes per slice, slices in order), so
// the chain agrees with a sequential double chain to ~1e-13 relative, not bit for bit.  Oracle: oracle/jwas_oracle_f64.c.
#pragma once
#include "kernels.hpp"

namespace jw64 {
using namespace jw;

constexpr int kMaxBlock64 = 1024;           // markers per block (the change list's capacity)
constexpr int kGramLds64 = 128;             // blocks up to this size keep their Gram in LDS

struct Events64 {
    int32_t count;
    int32_t idx[kMaxBlock64];               // global marker index
    double  delta[kMaxT][kMaxBlock64];      // alpha_old - alpha_new per trait
};

struct Params64 {
    int32_t method, ntraits, nreps;
    uint32_t iter, seed_lo, seed_hi, marker0, pad0;
    double vare[16], var_effect[16], Rinv[16], Ginv[16];
    double pi, pi4[4], gamma[4], log_prior[kMaxStates];
    const double* var_vec;      // p (BayesB)
    const double* pi_vec;       // p
    const double* pi_mat;       // p x 4
};

// ---- x'R^-1 x ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k64_xpx(const double* __restrict__ X, int64_t ld, const double* __restrict__ w, double* __restrict__ xpx)
{
    __shared__ double red[4];
    const double* x = X + (int64_t)blockIdx.x * ld;
    double v[1] = {0.0};
    for (int64_t i = threadIdx.x; i < ld; i += 256) v[0] += (x[i] * x[i]) * w[i];
    block_sum<1>(v, red, 4);
    if (threadIdx.x == 0) xpx[blockIdx.x] = v[0];
}

// ---- Gram of ONE block: G[a][c] = x_a' R^-1 x_c, row stride b.  grid = b, block = 256: workgroup a writes row a. ---------
__global__ __launch_bounds__(256) void k64_gram(const double* __restrict__ X, int64_t ld, const double* __restrict__ w, int64_t j0, int b,
                                                double* __restrict__ G)
{
    __shared__ double red[4];
    const int a = blockIdx.x;
    const double* xa = X + (j0 + a) * ld;
    double* out = G + (int64_t)a * b;
    for (int c = 0; c < b; ++c) {
        const double* xc = X + (j0 + c) * ld;
        double v[1] = {0.0};
        for (int64_t i = threadIdx.x; i < ld; i += 256) v[0] += (xa[i] * xc[i]) * w[i];
        block_sum<1>(v, red, 4);
        if (threadIdx.x == 0) out[c] = v[0];
        __syncthreads();
    }
}

// ---- update / partial: one 256-row slice x one column group per workgroup, one row per thread -----------------------------
// grid = (nslices, ncg).  Column group g forms the partial sums of columns [g * cpg, (g+1) * cpg) of the block (cpg a multiple
// of 8); EVERY group applies the previous block's changes to its copy of the slice (registers), group 0 writes the slice to
// r_out (r_in and r_out alternate from launch to launch: nobody reads what somebody else is writing).  r_out = NULL: no
// write (independent blocks: every block's sums from the same residual).
template <int NT>
__global__ __launch_bounds__(256) void k64_update_partial(const double* __restrict__ X, int64_t ld, const double* __restrict__ w,
                                                          const double* __restrict__ r_in /* [NT][ld] */, double* __restrict__ r_out,
                                                          const Events64* __restrict__ ev, int64_t j0, int b, int cpg,
                                                          double* __restrict__ partials /* [NT][nslices][bstride] */, int bstride)
{
    __shared__ double red[2][4][8 * NT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = (int64_t)blockIdx.x * 256 + tid;
    const int nslices = gridDim.x;
    double rv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) rv[t] = r_in[(int64_t)t * ld + row];
    const int nev = ev ? ev->count : 0;
    for (int e0 = 0; e0 < nev; e0 += 8) {                     // 8 changed columns in flight, the axpy chain per row in list order
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = X[(int64_t)ev->idx[e0 + u < nev ? e0 + u : nev - 1] * ld + row];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + u < nev) {
#pragma unroll
                for (int t = 0; t < NT; ++t) rv[t] = fma(ev->delta[t][e0 + u], x[u], rv[t]);        // axpy!(oldAlpha - alpha, x, yCorr)
            }
    }
    if (r_out != nullptr && blockIdx.y == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) r_out[(int64_t)t * ld + row] = rv[t];
    }
    const int c_lo = (int)blockIdx.y * cpg, c_hi = (c_lo + cpg < b) ? c_lo + cpg : b;
    if (c_lo >= b) return;
    const double wr = w[row];
    double rw[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) rw[t] = rv[t] * wr;          // X_b' (R^-1 r): block_rhs!, tools4genotypes.jl:59-78
    // partial right-hand sides, 8 columns at a time (transposed butterfly: 8 wave sums for ~10 shuffle-adds); the next
    // batch's loads are in flight while this one is reduced, and the cross-wave scratch is double-buffered (one barrier per
    // batch)
    double xn[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xn[u] = (c_lo + u < c_hi) ? X[(j0 + c_lo + u) * ld + row] : 0.0;
    int ph = 0;
    for (int c0 = c_lo; c0 < c_hi; c0 += 8, ph ^= 1) {
        double xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xv[u] = xn[u];
        if (c0 + 8 < c_hi) {
#pragma unroll
            for (int u = 0; u < 8; ++u) xn[u] = (c0 + 8 + u < c_hi) ? X[(j0 + c0 + 8 + u) * ld + row] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = xv[u] * rw[t];
            const double s = butterfly8(v, lane);                 // lane l: column ((l>>5)&1)*4 + ((l>>4)&1)*2 + ((l>>3)&1)
            if ((lane & 7) == 0) red[ph][wave][(lane >> 3) * NT + t] = s;
        }
        __syncthreads();
        if (tid < 8 * NT) {
            const int u = tid / NT, t = tid - u * NT;
            if (c0 + u < c_hi)
                partials[((int64_t)t * nslices + blockIdx.x) * bstride + c0 + u] = (red[ph][0][tid] + red[ph][1][tid]) + (red[ph][2][tid] + red[ph][3][tid]);
        }
    }
}

// ---- per-marker evaluation, all double.  Each returns the new effect(s); "changed" = the effect differs from the old one.
// BayesA/B/C: bayesabc_update_marker! (BayesABC.jl:24-58) with T = Float64.
struct Abc64 {
    double d, ie, iv, lv, sv, lp0, lp1, u, z;     // x'x, 1/vare, 1/var_j, log var_j, sqrt var_j, log pi, log(1-pi), draws
    __device__ __forceinline__ void eval(double x, double a_old, double& a_new, double& b_new, double& d_new) const
    {
        const double rhs = (x + d * a_old) * ie;                                  // :36
        const double lhs = d * ie + iv;                                           // :37
        const double invLhs = 1.0 / lhs;                                          // :38
        const double gHat = rhs * invLhs;                                         // :39
        const double logDelta1 = -0.5 * (log(lhs) + lv - gHat * rhs) + lp1;       // :40
        const double probDelta1 = 1.0 / (1.0 + exp(lp0 - logDelta1));             // :41
        if (u < probDelta1) { d_new = 1.0; b_new = gHat + z * sqrt(invLhs); a_new = b_new; }      // :44-48
        else { d_new = 0.0; b_new = z * sv; a_new = 0.0; }                        // :50-56
    }
};

// BayesR: BayesR! (BayesR.jl:56-96) with T = Float64; classes 1..4 (delta as stored), class 1 = zero effect.
struct R64 {
    double d, ie, sigma_sq, lpi[4], gamma[4], u, z;
    __device__ __forceinline__ void eval(double x, double a_old, double& a_new, int& cls_new) const
    {
        const double rhs = (x + d * a_old) * ie;                                  // :60
        double lp[4], probs[4];
        lp[0] = lpi[0];                                                           // :64
#pragma unroll
        for (int k = 1; k < 4; ++k) {                                             // :65-72
            const double varEffect = gamma[k] * sigma_sq;
            const double invVarEffect = 1.0 / varEffect;
            const double lhs = d * ie + invVarEffect;
            const double invLhs = 1.0 / lhs;
            const double betaHat = invLhs * rhs;
            lp[k] = 0.5 * (log(invLhs) - log(varEffect) + betaHat * rhs) + lpi[k];
        }
        double mx = lp[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) mx = lp[k] > mx ? lp[k] : mx;
        double se = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) se += exp(lp[k] - mx);
        const double log_norm = mx + log(se);                                     // bayesr_logsumexp :1-4
#pragma unroll
        for (int k = 0; k < 4; ++k) probs[k] = exp(lp[k] - log_norm);             // :75-77
        int cls = 0;                                                              // rand(Categorical(probs)) :79: CDF walk while cp <= u
        double cp = probs[0];
        while (cp <= u && cls < 3) { ++cls; cp += probs[cls]; }
        cls_new = cls + 1;
        if (cls == 0) { a_new = 0.0; return; }                                    // :82-86
        const double varEffect = gamma[cls] * sigma_sq;                           // :88-94
        const double lhs = d * ie + 1.0 / varEffect;
        const double invLhs = 1.0 / lhs;
        a_new = invLhs * rhs + z * sqrt(invLhs);
    }
};

// ---- sampler: one workgroup of 256 threads; wave 0 runs the chain -----------------------------------------------------
// LDS: gram [b][b] doubles (GLDS: blocks of <= 128 markers), then per marker and trait, stride bsz: rhs, the current effect,
// the effect at block entry, beta, delta.
struct Smem64 {
    int gram_off, rhs_off, acur_off, astart_off, bcur_off, dcur_off, bytes;
    __host__ __device__ Smem64(int b, int bsz, int NT, bool glds)
    {
        gram_off = 0; rhs_off = glds ? b * b * 8 : 0;
        const int arr = NT * bsz * 8;
        acur_off = rhs_off + arr; astart_off = acur_off + arr; bcur_off = astart_off + arr; dcur_off = bcur_off + arr;
        bytes = dcur_off + arr;
    }
};

template <int METHOD, int NT, bool GLDS>
__global__ __launch_bounds__(256) void k64_sample(const Params64* __restrict__ P, const double* __restrict__ gram /* b x b */,
                                                  const double* __restrict__ partials, int nslices, int bsz /* partials / LDS stride */,
                                                  int64_t j0, int b, int64_t p,
                                                  const double* __restrict__ xpx, double* __restrict__ alpha, double* __restrict__ beta,
                                                  void* __restrict__ delta, Events64* __restrict__ ev_out, unsigned long long* __restrict__ counters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Smem64 SM(b, bsz, NT, GLDS);
    double* G = reinterpret_cast<double*>(smem + SM.gram_off);       // [b][b]  (GLDS)
    double* rhs = reinterpret_cast<double*>(smem + SM.rhs_off);      // [NT][bsz]
    double* acur = reinterpret_cast<double*>(smem + SM.acur_off);
    double* astart = reinterpret_cast<double*>(smem + SM.astart_off);
    double* bcur = reinterpret_cast<double*>(smem + SM.bcur_off);
    double* dcur = reinterpret_cast<double*>(smem + SM.dcur_off);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (GLDS) for (int i = tid; i < b * b; i += 256) G[i] = gram[i];
    for (int i = tid; i < NT * b; i += 256) {
        const int t = i / b, c = i - t * b;
        const double* pp = partials + (int64_t)t * nslices * bsz + c;
        double s = 0.0;
        for (int sl0 = 0; sl0 < nslices; sl0 += 16) {          // 16 independent loads in flight, summed in slice order (fixed order)
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = pp[(int64_t)(sl0 + u < nslices ? sl0 + u : nslices - 1) * bsz];
#pragma unroll
            for (int u = 0; u < 16; ++u) if (sl0 + u < nslices) s += v[u];
        }
        rhs[t * bsz + c] = s;
        const double a0 = alpha[(int64_t)t * p + j0 + c];
        acur[t * bsz + c] = a0; astart[t * bsz + c] = a0;
        bcur[t * bsz + c] = (METHOD != kBayesR) ? beta[(int64_t)t * p + j0 + c] : 0.0;
        if (METHOD == kBayesR) dcur[t * bsz + c] = (double)reinterpret_cast<const int32_t*>(delta)[j0 + c];
        else dcur[t * bsz + c] = reinterpret_cast<const double*>(delta)[(int64_t)t * p + j0 + c];
    }
    __syncthreads();
    if (wave != 0) return;

    const int nreps = P->nreps > 0 ? P->nreps : b;
    const int nsub = (b + 63) / 64;
    RngKey key{P->seed_lo, P->seed_hi, P->iter, 0u};
    const double ie = 1.0 / P->vare[0];

    for (int rep = 0; rep < nreps; ++rep) {
        key.rep = (uint32_t)rep;
#pragma unroll 1
        for (int s = 0; s < nsub; ++s) {
            const int c = 64 * s + lane;
            const bool valid = c < b;
            const int cl = valid ? c : 0;
            const int64_t j = j0 + cl;
            const uint32_t marker = P->marker0 + (uint32_t)j;
            const double dj = xpx[j];
            unsigned long long pending = __ballot(valid);
            double a_cur[NT], b_cur[NT], d_cur[NT];              // the lane's marker: current effect, beta, delta
#pragma unroll
            for (int t = 0; t < NT; ++t) { a_cur[t] = acur[t * bsz + cl]; b_cur[t] = bcur[t * bsz + cl]; d_cur[t] = dcur[t * bsz + cl]; }
            // the marker's sweep constants and draws (fixed for this repetition)
            Abc64 am;
            R64 rm;
            double u_t[NT], z_t[NT];
            if constexpr (METHOD == kBayesC || METHOD == kBayesB) {
                const double var_j = (METHOD == kBayesB) ? P->var_vec[j] : P->var_effect[0];
                const double pi_j = P->pi_vec ? P->pi_vec[j] : P->pi;
                am.d = dj; am.ie = ie; am.iv = 1.0 / var_j; am.lv = log(var_j); am.sv = sqrt(var_j);
                am.lp0 = log(pi_j); am.lp1 = log(1.0 - pi_j);
                am.u = draw_uniform(key, marker, 0u); am.z = draw_normal(key, marker, 0u);
            } else if constexpr (METHOD == kBayesR) {
                rm.d = dj; rm.ie = ie; rm.sigma_sq = P->var_effect[0];
#pragma unroll
                for (int k = 0; k < 4; ++k) { rm.lpi[k] = log(P->pi_mat ? P->pi_mat[4 * j + k] : P->pi4[k]); rm.gamma[k] = P->gamma[k]; }
                rm.u = draw_uniform(key, marker, 0u); rm.z = draw_normal(key, marker, 0u);
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) { u_t[t] = draw_uniform(key, marker, (uint32_t)t); z_t[t] = draw_normal(key, marker, (uint32_t)t); }
            }
            while (pending) {
                // every pending lane evaluates ITS marker against the current rhs
                double an[NT], bn[NT], dn[NT];
                bool ev = false;
                if constexpr (METHOD == kBayesC || METHOD == kBayesB) {
                    am.eval(rhs[cl], a_cur[0], an[0], bn[0], dn[0]);
                    ev = an[0] != a_cur[0];
                } else if constexpr (METHOD == kBayesR) {
                    int cls;
                    rm.eval(rhs[cl], a_cur[0], an[0], cls);
                    bn[0] = 0.0; dn[0] = (double)cls;
                    ev = an[0] != a_cur[0];
                } else {
                    // _MTBayesABC_samplerI! (MTBayesABC.jl:76-121), T = Float64
                    double w[NT], bb[NT], dl[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) { w[t] = rhs[t * bsz + cl] + dj * a_cur[t]; bb[t] = b_cur[t]; dl[t] = d_cur[t]; an[t] = a_cur[t]; }
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        const double Ginv11 = P->Ginv[k * NT + k];                               // :86
                        const double C11 = Ginv11 + P->Rinv[k * NT + k] * dj;                    // :89
                        double rhs0 = 0.0, c12b = 0.0, wR = 0.0;
#pragma unroll
                        for (int m = 0; m < NT; ++m) {
                            wR = wR + w[m] * P->Rinv[m * NT + k];                                // :96
                            if (m == k) continue;
                            const double C12m = P->Ginv[k * NT + m] + (dj * dl[m]) * P->Rinv[k * NT + m];      // :90
                            rhs0 = rhs0 + P->Ginv[k * NT + m] * bb[m];                           // :93
                            c12b = c12b + C12m * bb[m];
                        }
                        rhs0 = -rhs0;
                        const double invLhs0 = 1.0 / Ginv11, gHat0 = rhs0 * invLhs0;             // :92,:94
                        const double invLhs1 = 1.0 / C11, rhs1 = wR - c12b, gHat1 = rhs1 * invLhs1;       // :95-97
                        unsigned s0 = 0u;
#pragma unroll
                        for (int m = 0; m < NT; ++m) if (m != k && dl[m] != 0.0) s0 |= 1u << m;
                        const unsigned s1 = s0 | (1u << k);
                        const double logDelta0 = -0.5 * (log(Ginv11) - gHat0 * gHat0 * Ginv11) + P->log_prior[s0];      // :104
                        const double logDelta1 = -0.5 * (log(C11) - gHat1 * gHat1 * C11) + P->log_prior[s1];            // :105
                        const double probDelta1 = 1.0 / (1.0 + exp(logDelta0 - logDelta1));                          // :107
                        if (u_t[k] < probDelta1) { dl[k] = 1.0; bb[k] = gHat1 + z_t[k] * sqrt(invLhs1); an[k] = bb[k]; }      // :108-111
                        else { bb[k] = gHat0 + z_t[k] * sqrt(invLhs0); dl[k] = 0.0; an[k] = 0.0; }                         // :112-119
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) { bn[t] = bb[t]; dn[t] = dl[t]; ev = ev || (an[t] != a_cur[t]); }
                }
                // a lane whose effect does not change still takes its new beta / delta when it becomes final (below)
                const unsigned long long m = __ballot(ev && valid) & pending;
                const int k = m ? (int)__builtin_ctzll(m) : 64;
                // lanes before the winner are final with what they just evaluated (their effect is unchanged)
                const unsigned long long done = (k >= 64) ? pending : (pending & ((k == 63) ? ~0ull : ((2ull << k) - 1ull)));
                if ((done >> lane) & 1ull) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) { b_cur[t] = bn[t]; d_cur[t] = dn[t]; }
                }
                if (k >= 64) break;
                // the winner commits; its Gram row corrects the rhs of the whole block (BayesABC.jl:169,172)
                const int ce = 64 * s + k;
                const double* grow = GLDS ? nullptr : gram + (int64_t)ce * b;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const double Dl = a_cur[t] - an[t];
                    const double D = __shfl(Dl, k, 64);
                    if (lane == k) a_cur[t] = an[t];
                    if (D != 0.0) {
                        if constexpr (GLDS) { for (int c2 = lane; c2 < b; c2 += 64) rhs[t * bsz + c2] = fma(D, G[ce * b + c2], rhs[t * bsz + c2]); }
                        else { for (int c2 = lane; c2 < b; c2 += 64) rhs[t * bsz + c2] = fma(D, grow[c2], rhs[t * bsz + c2]); }
                    }
                }
                pending &= ~done;
            }
            if (valid) {
#pragma unroll
                for (int t = 0; t < NT; ++t) { acur[t * bsz + c] = a_cur[t]; bcur[t * bsz + c] = b_cur[t]; dcur[t * bsz + c] = d_cur[t]; }
            }
        }
    }
    // write back + the block's change list (marker order)
    int base = 0;
#pragma unroll 1
    for (int s = 0; s < nsub; ++s) {
        const int c = 64 * s + lane;
        const bool valid = c < b;
        const int cl = valid ? c : 0;
        bool changed = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) changed = changed || (acur[t * bsz + cl] != astart[t * bsz + cl]);
        changed = changed && valid;
        const unsigned long long cm = __ballot(changed);
        if (changed) {
            const int e = base + __popcll(cm & ((1ull << lane) - 1ull));
            ev_out->idx[e] = (int32_t)(j0 + c);
#pragma unroll
            for (int t = 0; t < NT; ++t) ev_out->delta[t][e] = astart[t * bsz + c] - acur[t * bsz + c];
        }
        base += __popcll(cm);
        if (valid) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                alpha[(int64_t)t * p + j0 + c] = acur[t * bsz + c];
                if (METHOD == kBayesR) reinterpret_cast<int32_t*>(delta)[j0 + c] = (int32_t)dcur[c];
                else { beta[(int64_t)t * p + j0 + c] = bcur[t * bsz + c]; reinterpret_cast<double*>(delta)[(int64_t)t * p + j0 + c] = dcur[t * bsz + c]; }
            }
        }
    }
    if (lane == 0) { ev_out->count = base; atomicAdd(&counters[0], (unsigned long long)base); }
}

// ---- epilogue: apply the last block's changes (independent blocks: EVERY block's, in (block, marker) order -- the reconcile of
// BayesABC.jl:251-253), r'R^-1 r and sum(R^-1 r) per slice --------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void k64_finish(const double* __restrict__ X, int64_t ld, int64_t n, const double* __restrict__ w,
                                                  const double* r_in, double* r_out /* may be the same buffer: one row per thread */,
                                                  const Events64* __restrict__ ev, int64_t nlists, double* __restrict__ out /* [nslices][NT*NT+NT] */)
{
    __shared__ double red[4 * (NT * NT + NT)];
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double rv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) rv[t] = r_in[(int64_t)t * ld + row];
    for (int64_t q = 0; q < nlists; ++q) {
        const Events64* L = ev + q;
        const int nev = L->count;
        for (int e0 = 0; e0 < nev; e0 += 8) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = X[(int64_t)L->idx[e0 + u < nev ? e0 + u : nev - 1] * ld + row];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + u < nev) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) rv[t] = fma(L->delta[t][e0 + u], x[u], rv[t]);
                }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) r_out[(int64_t)t * ld + row] = rv[t];
    double v[NT * NT + NT];
    const double live = row < n ? w[row] : 0.0;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
#pragma unroll
        for (int c = 0; c < NT; ++c) v[a * NT + c] = rv[a] * rv[c] * live;
        v[NT * NT + a] = rv[a] * live;
    }
    block_sum<NT * NT + NT>(v, red, 4);
    if (threadIdx.x == 0)
#pragma unroll
        for (int i = 0; i < NT * NT + NT; ++i) out[(int64_t)blockIdx.x * (NT * NT + NT) + i] = v[i];
}

// ---- marker statistics (one workgroup per kStat slot, fixed order) ------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void k64_marker_stats(int method, int64_t p, const double* __restrict__ alpha, const double* __restrict__ beta,
                                                        const void* __restrict__ delta, const double* __restrict__ gamma, double* __restrict__ out)
{
    __shared__ double red[4 * kNStat];
    double v[kNStat];
#pragma unroll
    for (int i = 0; i < kNStat; ++i) v[i] = 0.0;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < p; j += (int64_t)gridDim.x * 256) {
        double a[NT], b[NT];
        unsigned st = 0u;
#pragma unroll
        for (int t = 0; t < NT; ++t) { a[t] = alpha[(int64_t)t * p + j]; b[t] = beta[(int64_t)t * p + j]; }
        if (method == kBayesR) {
            const int cls = reinterpret_cast<const int32_t*>(delta)[j];
            v[36 + (cls - 1)] += 1.0;
            if (cls > 1) { v[40] += (a[0] * a[0]) / gamma[cls - 1]; v[41] += 1.0; }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) { const double d = reinterpret_cast<const double*>(delta)[(int64_t)t * p + j]; v[t] += d; if (d != 0.0) st |= 1u << t; }
            v[42 + st] += 1.0;
        }
#pragma unroll
        for (int x = 0; x < NT; ++x)
#pragma unroll
            for (int y = 0; y < NT; ++y) { v[4 + x * NT + y] += a[x] * a[y]; v[20 + x * NT + y] += b[x] * b[y]; }
    }
    block_sum<kNStat>(v, red, 4);
    if (threadIdx.x == 0)
        for (int i = 0; i < kNStat; ++i) out[(int64_t)blockIdx.x * kNStat + i] = v[i];
}

// ---- running posterior means (output.jl:556-560) -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void k64_accumulate(int64_t count, int delta_is_class, double k, const double* __restrict__ alpha,
                                                      const void* __restrict__ delta, double* __restrict__ ma, double* __restrict__ ma2, double* __restrict__ md)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const double a = alpha[i];
    const double d = delta_is_class ? (reinterpret_cast<const int32_t*>(delta)[i] > 1 ? 1.0 : 0.0) : reinterpret_cast<const double*>(delta)[i];
    ma[i] += (a - ma[i]) / k;
    ma2[i] += (a * a - ma2[i]) / k;
    md[i] += (d - md[i]) / k;
}

// ---- out = X alpha (getEBV, output.jl:281-306); r -= X alpha (initial ycorr) ---------------------------------------------
__global__ __launch_bounds__(256) void k64_mul_alpha(const double* __restrict__ X, int64_t ld, int64_t p, const double* __restrict__ alpha,
                                                     double* __restrict__ out, double sign, const double* __restrict__ base)
{
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double acc = base ? base[row] : 0.0;
    for (int64_t j = 0; j < p; ++j) {
        const double a = alpha[j];
        if (a != 0.0) acc = fma(sign * a, X[j * ld + row], acc);
    }
    out[row] = acc;
}

}  // namespace jw64
