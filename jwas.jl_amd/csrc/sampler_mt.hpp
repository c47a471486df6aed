// sampler_mt.hpp -- SAMPLER role, multi-trait: per-marker evaluations of samplers I (with Rule L) / II / megaBayesABC, the per-marker
// inverse-Wishart draws of multi-trait BayesA/B, k_prepare_mt2 and sampler_role_mt.  Included by sweep.hpp.
#pragma once
#include "kernels.hpp"

namespace jw {

// ---------------------------------------------------------------------------------------------
// Per-marker evaluation of the multi-trait samplers.  Inputs: w[k] = rhs_k + d*alpha_k (fp32), the
// marker's current (alpha, beta, delta), its draws.  Outputs: new (an, bn, dn) and the axpy
// coefficients Dl[k] (alpha_old - alpha_new; 0 = no change).  Operation for operation the oracle's
// mt1_update / mt2_update / mega_update.
// ---------------------------------------------------------------------------------------------
template <int NT>
struct MtConsts {
    float Rinv[NT][NT], Ginv[NT][NT];
    float Rm[NT][NT];                                 // the residual covariance itself (Rule T: the rhs a solved effect implies)
    float invG[NT], lG[NT], sG[NT];                   // sampler I: 1/Ginv_kk, log Ginv_kk, sqrt(1/Ginv_kk)
    // mega (constraint = true): per-trait single-trait BayesC constants
    float ie[NT], var[NT], iv[NT], lv[NT], sv[NT];   // sv = sqrt(var)
    double lp0[NT], lp1[NT];
};

// Per-marker quantities that depend only on x'x (not on the running rhs): computed once per marker, SIMD across the
// markers of a sub-block, instead of inside every evaluation (each holds a double-precision log).
template <int NT>
struct MtPre {
    float C11[NT], invLhs1[NT], lC11[NT], s1[NT];     // sampler I: C11, 1/C11, log C11, sqrt(1/C11)
                                                      // mega:      lhs, 1/lhs, log(lhs) + log(var), sqrt(1/lhs)
};
// lc = the logs k_prepare took for this marker (prep_f rows 0..NT-1)
template <int METHOD, int NT>
__device__ __forceinline__ MtPre<NT> mt_precompute(const MtConsts<NT>& K, float dj, const float (&lc)[NT])
{
    MtPre<NT> R;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        if constexpr (is_mega(METHOD)) {
            R.C11[k] = dj * K.ie[k] + K.iv[k];                                     // BayesABC.jl:37
            R.invLhs1[k] = 1.0f / R.C11[k];                                        // :38
        } else {
            R.C11[k] = K.Ginv[k][k] + K.Rinv[k][k] * dj;                           // MTBayesABC.jl:89
            R.invLhs1[k] = 1.0f / R.C11[k];                                        // :95
        }
        R.lC11[k] = lc[k];
        R.s1[k] = sqrtf(R.invLhs1[k]);
    }
    return R;
}

// Log prior probabilities of the two joint states sampler I compares for trait k (delta_k = 0 / 1, the other traits as
// they are now).  PriorMem: a table in memory (LDS; stride ls between states: 1 = the shared table, block size = this
// marker's column of the marker-specific priors).
struct PriorMem {
    const double* lpr; int ls;
    template <int NT>
    __device__ __forceinline__ void pair(int k, const float (&dn)[NT], double& l0, double& l1) const
    {
        unsigned s0 = 0u;
#pragma unroll
        for (int m = 0; m < NT; ++m) if (m != k && dn[m] != 0.f) s0 |= 1u << m;
        l0 = lpr[s0 * ls];
        l1 = lpr[(s0 | (1u << k)) * ls];
    }
};
// ---- Rule L (sampler I): the LINEAR FORM of a marker that is in the model for every trait and stays there.
// With every delta = 1 before and after, the marker's NT conditionals are one small triangular system:
//     beta_k = 1/C11_k * ( sum_m Rinv[m][k] w_m - sum_{m<k} C12[k][m] beta_m - sum_{m>k} C12[k][m] beta_old_m ) + z_k sqrt(1/C11_k)
// i.e.  beta = A w + c  with A (NT x NT) and c (NT) functions of the marker's constants, its old beta and its draws only --
// NOT of the running rhs.  The dense walk (every marker in the model: the reference's default all-ones prior) precomputes
// A, c for all 64 markers of a section in parallel and is left with NT^2 fused multiply-adds per marker on the serial
// chain instead of the ~26 dependent operations of the conditional-by-conditional order (430 -> ~110 cycles per 3-trait
// marker).  So that every path (dense walk, speculative rounds, the oracle) produces the SAME numbers, the rule is part
// of the sampler's definition: whenever the exact evaluation (below, the reference's operation order) says that a marker
// which entered with every delta = 1 leaves with every delta = 1, its new effects are the linear form's
//     beta_k = fmaf(A[k][NT-1], w[NT-1], ... fmaf(A[k][0], w[0], c[k]))      (A, c: double recurrence, rounded to float)
// -- the same conditional means and the same draws, another association: <= a few ulp from the reference's order.  Every
// other marker keeps the exact order's values.  The oracle applies the same rule (orc mt1_update; the literal order stays
// available there: orc_set_mt_linear_form(0), compared in tests/test_oracle_kat.py).
template <int NT>
__device__ __forceinline__ void mt1_linear_coeffs(const MtConsts<NT>& K, const MtPre<NT>& Q, float dj, const float (&b_old)[NT],
                                                  const double (&z)[NT], float (&A)[NT][NT], float (&cc)[NT])
{
    double Ad[NT][NT], cd[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const double il = (double)Q.invLhs1[k];
        double C12[NT];
#pragma unroll
        for (int m = 0; m < NT; ++m) C12[m] = (double)(K.Ginv[k][m] + (dj * 1.f) * K.Rinv[k][m]);       // MTBayesABC.jl:90 with delta_m = 1
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            double acc = (double)K.Rinv[m][k];
#pragma unroll
            for (int j = 0; j < k; ++j) acc = acc - C12[j] * Ad[j][m];
            Ad[k][m] = il * acc;
        }
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < k; ++j) acc = acc - C12[j] * cd[j];
#pragma unroll
        for (int j = k + 1; j < NT; ++j) acc = acc - C12[j] * (double)b_old[j];
        cd[k] = il * acc + z[k] * (double)Q.s1[k];
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        cc[k] = (float)cd[k];
#pragma unroll
        for (int m = 0; m < NT; ++m) A[k][m] = (float)Ad[k][m];
    }
}
template <int NT>
__device__ __forceinline__ void mt1_linear_beta(const float (&A)[NT][NT], const float (&cc)[NT], const float (&w)[NT], float (&b)[NT])
{
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        float v = cc[k];
#pragma unroll
        for (int m = 0; m < NT; ++m) v = fmaf(A[k][m], w[m], v);
        b[k] = v;
    }
}

// The matrix A of Rule L alone (the part of mt1_linear_coeffs that does not depend on the marker's old effects or draws --
// only on its x'x and the sweep's covariances): the same operations, the same floats.
template <int NT>
__device__ __forceinline__ void mt1_linear_A(const MtConsts<NT>& K, float dj, float (&A)[NT][NT])
{
    double Ad[NT][NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const float C11 = K.Ginv[k][k] + K.Rinv[k][k] * dj;                                     // MTBayesABC.jl:89
        const double il = (double)(1.0f / C11);                                                 // :95
        double C12[NT];
#pragma unroll
        for (int m = 0; m < NT; ++m) C12[m] = (double)(K.Ginv[k][m] + (dj * 1.f) * K.Rinv[k][m]);
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            double acc = (double)K.Rinv[m][k];
#pragma unroll
            for (int j = 0; j < k; ++j) acc = acc - C12[j] * Ad[j][m];
            Ad[k][m] = il * acc;
        }
    }
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int m = 0; m < NT; ++m) A[k][m] = (float)Ad[k][m];
}

// ---------------------------------------------------------------------------------------------
// RULE T (round 5): the dense block chain as the TRIANGULAR SOLVE it is.
// With every marker of a 64-marker section in the model for every trait, Rule L makes the section's chain linear in the
// changes D_l = alpha_old_l - alpha_new_l (NT-vectors):
//     D_l = y_l - A_l * sum_{j<l} G_lj D_j ,      y_l = alpha_old_l - (A_l (rhs_l + d_l alpha_old_l) + c_l)
// (rhs_l: the section-entry right-hand side; G: the section's diagonal Gram tile) -- i.e. (I + L) D = y with the strictly
// lower block matrix L[(l,k),(j,m)] = A_l[k][m] G_lj.  A_l depends on the marker's x'x and the sweep's covariances only, so
// T_s = (I + L)^-1 (64 NT x 64 NT, lower block triangular) is formed ONCE PER SWEEP for all sections in parallel
// (k_section_inverse_mt: one workgroup per section, one thread per column, forward substitution) and a section's chain is
// one mat-vec  D~ = T_s y  spread over 2 NT waves instead of a 64-step walk of one wave.  The new effects are
// alpha_new = alpha_old - D~ (the linear form's own values up to rounding: the same conditional means and draws); the
// literal evaluation (MTBayesABC.jl:85-120) at the right-hand side those effects imply,
//     rhs~_l = rhs_l + R Lc_l (y_l - D~_l)      (Lc_l: the marker's lower-triangular C11 / C12 matrix, R: residual covariance),
// verifies that every indicator stays 1; if one does not, the section runs through the sequential chain (the walk) from its
// saved state.  Part of the sampler's definition when jwas_sweep_params.section_solve is set (the oracle restates it
// operation for operation: orc mt1_section_solve); off = the sequential chain everywhere, bit for bit as before.
// Layout of T_s in HBM: float [16 NT][64 NT][4] -- column group cg (columns 4 cg .. 4 cg + 3; column = m * 64 + j for trait m
// of marker j), row r = k * 64 + l, the row's four values of the group: one float4 per lane, 1 KB per wave load.
// Summation order of D~[(l,k)]: four fmaf chains from 0 over the columns [16 NT q, 16 NT (q + 1)), q = 0..3, in ascending order,
// added as (p0 + p1) + (p2 + p3).
// ---------------------------------------------------------------------------------------------
template <int NT> __host__ __device__ constexpr int64_t tsec_floats() { return (int64_t)(64 * NT) * (64 * NT); }
// EXCEPTIONS of a solved section (a marker that is not in the model for every trait at entry, or that the verification finds
// leaving it).  T_s is lower block triangular, so everything before the first exception e is final; the literal evaluation of e
// at the right-hand side the solve implies for it, rhs_e + R Lc_e (y_e - D~_e), IS the chain's evaluation of e (whatever y_e
// was: the row only recovers sum_{j<e} G_ej D_j from D~); its result v = alpha_old - alpha_new replaces D~_e and the rows
// behind it take the rank-NT correction  D~_r += sum_m T_s[r,(e,m)] (v_m - D~_(e,m))  -- the solve of the same system with row e
// replaced by "D_e = v".  Exceptions are taken in marker order until none is left.  A section with more than kSolveMaxOdd
// markers outside the model at entry is walked; one with more than kSolveMaxExc exceptions in all falls back to the walk.
// (oracle: ORC_SOLVE_MAX_ODD / ORC_SOLVE_MAX_EXC, mt1_section_solve)
constexpr int kSolveMaxOdd = 16, kSolveMaxExc = 24;      // (an exception costs ~1.9 k cycles, a walked marker outside the model ~1.5 k on top of the 16 k walk)

// One section's inverse.  grid = sections (4 per full 256-marker block), block = 256 NT threads: FOUR adjacent lanes per column
// (column (jc, mc) = thread / 4), lane q of the quad sums the terms j = jc + q, jc + q + 4, ... of a row's dot product, the quad adds
// its four partial sums as (p0 + p1) + (p2 + p3) with two lane exchanges (same bits in all four lanes) -- the column's 63 dependent
// steps are a quarter as long and the CU runs twelve waves instead of three (1.34 -> ~0.35 ms per sweep at 20k x 100k x 3).
// LDS: the tile's G (16 KB), A of the 64 markers, the inverse in packed lower-block-triangular form (NT^2 * 2080 floats).
template <int METHOD, int NT>
__global__ __launch_bounds__(256 * NT) void k_section_inverse_mt(const DevParams* __restrict__ P, const float* __restrict__ xpx,
                                                               const float* __restrict__ gram /* blocks at stride 256*256 */,
                                                               const float* __restrict__ ginv_mat, float* __restrict__ tsec)
{
    static_assert(is_sampler1(METHOD), "sampler I");
    constexpr int NR = 64 * NT;
    extern __shared__ __attribute__((aligned(16))) char smem_ti[];
    float* Gt = reinterpret_cast<float*>(smem_ti);                  // [64][64] the section's diagonal Gram tile
    float* Al = Gt + 4096;                                          // [64][NT*NT]
    float* Xp = Al + 64 * NT * NT;                                  // packed inverse: row (l,k), columns of markers <= l
    const int tid = threadIdx.x;
    const int64_t sec = blockIdx.x, blk = sec >> 2;
    const int s = (int)(sec & 3);
    const float* G = gram + blk * (int64_t)(256 * 256) + (int64_t)(64 * s) * 256 + 64 * s;
    constexpr int NTHR = 256 * NT;
    for (int e = tid; e < 4096 / 4; e += NTHR) {
        const int l = e >> 4, c4 = (e & 15) * 4;
        *reinterpret_cast<float4*>(Gt + l * 64 + c4) = *reinterpret_cast<const float4*>(G + (int64_t)l * 256 + c4);
    }
    if (tid < 64) {
        const int64_t j = blk * 256 + 64 * s + tid;
        MtConsts<NT> K;
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                K.Rinv[a][c] = P->Rinv[a * NT + c];
                K.Ginv[a][c] = has_marker_cov(METHOD) ? ginv_mat[j * (NT * NT) + a * NT + c] : P->Ginv[a * NT + c];
            }
        float A[NT][NT];
        mt1_linear_A<NT>(K, xpx[j], A);
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int c = 0; c < NT; ++c) Al[tid * NT * NT + a * NT + c] = A[a][c];
    }
    __syncthreads();
    // quad = column (jc, mc); rows above the column's marker are zero (not stored), its own marker's rows are the identity
    const int col = tid >> 2, q = tid & 3;
    const int mc = col >> 6, jc = col & 63;                         // column index c = mc * 64 + jc
    auto xoff = [](int l, int k) { return NT * NT * (l * (l + 1) / 2) + k * (l + 1) * NT; };      // row (l,k): columns (j <= l, m) at + j * NT + m
    if (q == 0) {
#pragma unroll
        for (int k = 0; k < NT; ++k) Xp[xoff(jc, k) + jc * NT + mc] = (k == mc) ? 1.f : 0.f;
    }
    // (a quad reads back only what its own wave wrote: LDS operations of one wave execute in order -- no barrier inside)
#pragma unroll 1
    for (int l = jc + 1; l < 64; ++l) {
        double u[NT];
#pragma unroll
        for (int m = 0; m < NT; ++m) u[m] = 0.0;
#pragma unroll 1
        for (int j = jc + q; j < l; j += 4) {
            const double g = (double)Gt[l * 64 + j];
#pragma unroll
            for (int m = 0; m < NT; ++m) u[m] = fma(g, (double)Xp[xoff(j, m) + jc * NT + mc], u[m]);
        }
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            u[m] = u[m] + __shfl_xor(u[m], 1, 64);                  // (p0 + p1), (p2 + p3): a + b = b + a, the same bits on both sides
            u[m] = u[m] + __shfl_xor(u[m], 2, 64);                  // (p0 + p1) + (p2 + p3)
        }
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            double v = 0.0;
#pragma unroll
            for (int m = 0; m < NT; ++m) v = fma((double)Al[l * NT * NT + k * NT + m], u[m], v);
            if (q == 0) Xp[xoff(l, k) + jc * NT + mc] = (float)(-v);
        }
    }
    __syncthreads();
    // coalesced write-out in the sampler's layout: [cg][r][4], r = k*64 + l, column = m*64 + j
    float* dst = tsec + sec * tsec_floats<NT>();
    for (int e = tid; e < (NR / 4) * NR; e += NTHR) {
        const int cg = e / NR, r = e - cg * NR;
        const int k = r >> 6, l = r & 63;
        float4 v;
        float* vv = reinterpret_cast<float*>(&v);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 4 * cg + q, m = c >> 6, j = c & 63;
            vv[q] = (j <= l) ? Xp[xoff(l, k) + j * NT + m] : 0.f;
        }
        *reinterpret_cast<float4*>(dst + (int64_t)e * 4) = v;
    }
}
template <int NT> __host__ __device__ constexpr int tsec_inverse_lds_bytes() { return 4 * (4096 + 64 * NT * NT + NT * NT * 2080); }

// Gibbs sampler I (MTBayesABC.jl:85-120); LIN: apply Rule L to the result (off only where the result's VALUES are not kept).
// Apre / cpre: Rule L's coefficients of this marker when the caller has formed them already (the dense walk: once per
// section) -- the same numbers mt1_linear_coeffs would give here (same constants, entry beta and draws).
template <int NT, bool LIN = true, class LP>
__device__ __forceinline__ void mt1_eval(const MtConsts<NT>& K, const MtPre<NT>& Q, const LP& lp, const float (&w)[NT], float dj,
                                         const double (&thr)[NT], const double (&z)[NT],
                                         float (&an)[NT], float (&bn)[NT], float (&dn)[NT], float (&Dl)[NT],
                                         const float (*Apre)[NT] = nullptr, const float* cpre = nullptr)
{
    float a_in[NT], b_in[NT];
    bool all1 = LIN;
#pragma unroll
    for (int k = 0; k < NT; ++k) { a_in[k] = an[k]; b_in[k] = bn[k]; all1 = all1 && (dn[k] == 1.f); }
#pragma unroll
    for (int k = 0; k < NT; ++k) {                                                  // :85
        const float Ginv11 = K.Ginv[k][k];
        const float C11 = Q.C11[k];                                                 // :89
        float rhs0 = 0.f, c12b = 0.f, wR = 0.f;
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            wR = wR + w[m] * K.Rinv[m][k];
            if (m == k) continue;
            const float C12m = K.Ginv[k][m] + (dj * dn[m]) * K.Rinv[k][m];          // :90
            rhs0 = rhs0 + K.Ginv[k][m] * bn[m];
            c12b = c12b + C12m * bn[m];
        }
        rhs0 = -rhs0;                                                               // :93
        const float invLhs0 = K.invG[k];
        const float gHat0 = rhs0 * invLhs0;
        const float invLhs1 = Q.invLhs1[k];
        const float rhs1 = wR - c12b;                                               // :96
        const float gHat1 = rhs1 * invLhs1;
        double lp0, lp1;
        lp.template pair<NT>(k, dn, lp0, lp1);
        const float in0 = K.lG[k] - (gHat0 * gHat0) * Ginv11;                       // :104
        const float in1 = Q.lC11[k] - (gHat1 * gHat1) * C11;                        // :105
        const double logDelta0 = -0.5 * (double)in0 + lp0;
        const double logDelta1 = -0.5 * (double)in1 + lp1;
        if ((logDelta0 - logDelta1) < thr[k]) {                                     // :107-111
            dn[k] = 1.f;
            bn[k] = (float)((double)gHat1 + z[k] * (double)Q.s1[k]);
            Dl[k] = an[k] - bn[k];
            an[k] = bn[k];
        } else {                                                                    // :112-119
            bn[k] = (float)((double)gHat0 + z[k] * (double)K.sG[k]);
            dn[k] = 0.f;
            Dl[k] = an[k];
            an[k] = 0.f;
        }
    }
    if constexpr (LIN) {
#pragma unroll
        for (int k = 0; k < NT; ++k) all1 = all1 && (dn[k] == 1.f);
        if (all1) {                                                                 // Rule L
            float A[NT][NT], cc[NT];
            if (Apre != nullptr) {
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    cc[k] = cpre[k];
#pragma unroll
                    for (int m = 0; m < NT; ++m) A[k][m] = Apre[k][m];
                }
            } else mt1_linear_coeffs<NT>(K, Q, dj, b_in, z, A, cc);
            mt1_linear_beta<NT>(A, cc, w, bn);
#pragma unroll
            for (int k = 0; k < NT; ++k) { an[k] = bn[k]; Dl[k] = a_in[k] - bn[k]; }
        }
    }
}

// ---- mt1_eval with everything that does not depend on the running right-hand side HOISTED out of it (round 6).  A dense walk
// through MIXED joint states -- the transition of a default-prior chain from "every marker in the model" to its sparse steady state,
// ~1 000 sweeps in which every marker needs the literal evaluation at its own step -- is one wave issuing mt1_eval once per marker:
// ~250 instructions, bound by their COUNT.  When trait k is sampled, the traits after it still hold the marker's ENTRY state, so
// every product and sum over m > k (MTBayesABC.jl:90,93,96), the draws' scaled normals, C12 with delta_m = 1, and for trait 0 the
// whole "excluded" branch are functions of the marker's entry state, draws and constants only: formed ONCE per marker (all lanes
// in parallel, before the walk), they leave ~140 instructions per step.  The SAME operations on the same numbers in the same
// association as mt1_eval<NT, true> (every sum keeps its order: the m < k terms first, then the m > k terms one by one) -- bit for
// bit the same decisions and values (dense_big_mt verifies every lane once more with mt1_eval itself after the walk).
template <int NT>
struct Mt1Hoist {
    float tg[NT][NT];          // m > k: Ginv[k][m] * b_entry[m]                                   (a term of rhs0, :93)
    float tc[NT][NT];          // m > k: (Ginv[k][m] + (dj * d_entry[m]) * Rinv[k][m]) * b_entry[m] (a term of C12' beta, :90,:96)
    float c12on[NT][NT];       // m < k: Ginv[k][m] + (dj * 1) * Rinv[k][m]   (C12 when trait m is in the model; out of it: Ginv[k][m])
    double zs1[NT], zs0[NT];   // z_k sqrt(1 / C11_k),  z_k sqrt(1 / Ginv_kk)                       (:109,:114)
    double ld0_0, lp1_0;       // trait 0: logDelta0 (nothing in it moves with the rhs), log prior of its "in the model" state
    float c12b_0, b0_excl;     // trait 0: C12' beta; its effect when it stays out
    bool all1_in;              // the marker entered with every indicator 1 (Rule L applies if it also leaves that way)
    // traits 1 and 2: the log prior probabilities of the two joint states the conditional compares, for every configuration of the
    // traits BELOW it (the ones above hold their entry state): [configuration of traits 0..k-1][delta_k] -- registers instead of an
    // LDS lookup whose address depends on the trait before it (two LDS latencies on the chain of every step)
    double lpt1[2][2], lpt2[4][2];
};
template <int NT, class LP>
__device__ __forceinline__ Mt1Hoist<NT> mt1_hoist(const MtConsts<NT>& K, const MtPre<NT>& Q, const LP& lp, float dj,
                                                  const float (&b_in)[NT], const float (&d_in)[NT], const double (&z)[NT])
{
    Mt1Hoist<NT> H;
    H.all1_in = true;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        H.all1_in = H.all1_in && (d_in[k] == 1.f);
        H.zs1[k] = z[k] * (double)Q.s1[k];
        H.zs0[k] = z[k] * (double)K.sG[k];
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            H.tg[k][m] = 0.f; H.tc[k][m] = 0.f; H.c12on[k][m] = 0.f;
            if (m > k) {
                const float C12m = K.Ginv[k][m] + (dj * d_in[m]) * K.Rinv[k][m];                   // :90
                H.tg[k][m] = K.Ginv[k][m] * b_in[m];
                H.tc[k][m] = C12m * b_in[m];
            } else if (m < k) H.c12on[k][m] = K.Ginv[k][m] + (dj * 1.f) * K.Rinv[k][m];
        }
    }
    float rhs0 = 0.f, c12b = 0.f;
#pragma unroll
    for (int m = 1; m < NT; ++m) { rhs0 = rhs0 + H.tg[0][m]; c12b = c12b + H.tc[0][m]; }
    rhs0 = -rhs0;                                                                                   // :93
    const float gHat0 = rhs0 * K.invG[0];
    double lp0, lp1;
    lp.template pair<NT>(0, d_in, lp0, lp1);
    const float in0 = K.lG[0] - (gHat0 * gHat0) * K.Ginv[0][0];                                     // :104
    H.ld0_0 = -0.5 * (double)in0 + lp0;
    H.lp1_0 = lp1;
    H.c12b_0 = c12b;
    H.b0_excl = (float)((double)gHat0 + H.zs0[0]);
#pragma unroll
    for (int cfg = 0; cfg < 4; ++cfg) {
        float dc[NT];
#pragma unroll
        for (int m = 0; m < NT; ++m) dc[m] = d_in[m];
        dc[0] = (cfg & 1) ? 1.f : 0.f;
        if (cfg < 2) {
            if constexpr (NT >= 2) lp.template pair<NT>(1, dc, H.lpt1[cfg][0], H.lpt1[cfg][1]);
            else { H.lpt1[cfg][0] = 0.0; H.lpt1[cfg][1] = 0.0; }
        }
        if constexpr (NT >= 3) { dc[1] = (cfg & 2) ? 1.f : 0.f; lp.template pair<NT>(2, dc, H.lpt2[cfg][0], H.lpt2[cfg][1]); }
        else { H.lpt2[cfg][0] = 0.0; H.lpt2[cfg][1] = 0.0; }
    }
    return H;
}
// a_in / d_in: the marker's entry alpha / delta; out: bn, dn, Dl = alpha_old - alpha_new (what the walk broadcasts)
template <int NT, class LP>
__device__ __forceinline__ void mt1_eval_hoisted(const MtConsts<NT>& K, const MtPre<NT>& Q, const Mt1Hoist<NT>& H, const LP& lp,
                                                 const float (&w)[NT], const float (&a_in)[NT], const float (&d_in)[NT],
                                                 const double (&thr)[NT], const float (&Al)[NT][NT], const float (&cl)[NT],
                                                 float (&bn)[NT], float (&dn)[NT], float (&Dl)[NT])
{
    float dcomb[NT];                                 // the indicators as the conditional of trait k sees them: new below k, entry above
#pragma unroll
    for (int k = 0; k < NT; ++k) dcomb[k] = d_in[k];
    bool all1 = H.all1_in;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        float wR = 0.f;
#pragma unroll
        for (int m = 0; m < NT; ++m) wR = wR + w[m] * K.Rinv[m][k];
        float c12b, gHat0 = 0.f;
        double ld0, lp1;
        if (k == 0) { c12b = H.c12b_0; ld0 = H.ld0_0; lp1 = H.lp1_0; }
        else {
            float rhs0 = 0.f;
            c12b = 0.f;
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                if (m < k) {
                    const float C12m = (dn[m] != 0.f) ? H.c12on[k][m] : K.Ginv[k][m];              // :90
                    rhs0 = rhs0 + K.Ginv[k][m] * bn[m];
                    c12b = c12b + C12m * bn[m];
                } else if (m > k) {
                    rhs0 = rhs0 + H.tg[k][m];
                    c12b = c12b + H.tc[k][m];
                }
            }
            rhs0 = -rhs0;                                                                           // :93
            gHat0 = rhs0 * K.invG[k];
            double lp0;
            if (k == 1) { const bool d0 = dn[0] != 0.f; lp0 = d0 ? H.lpt1[1][0] : H.lpt1[0][0]; lp1 = d0 ? H.lpt1[1][1] : H.lpt1[0][1]; }
            else if (k == 2) {
                const bool d0 = dn[0] != 0.f, d1 = dn[1] != 0.f;
                const double a0 = d0 ? H.lpt2[1][0] : H.lpt2[0][0], a1 = d0 ? H.lpt2[3][0] : H.lpt2[2][0];
                const double c0 = d0 ? H.lpt2[1][1] : H.lpt2[0][1], c1 = d0 ? H.lpt2[3][1] : H.lpt2[2][1];
                lp0 = d1 ? a1 : a0; lp1 = d1 ? c1 : c0;
            } else lp.template pair<NT>(k, dcomb, lp0, lp1);
            const float in0 = K.lG[k] - (gHat0 * gHat0) * K.Ginv[k][k];                             // :104
            ld0 = (double)(-0.5f * in0) + lp0;             // (-0.5 * (double)in0: the scaling by a power of two is exact in either type)
        }
        const float rhs1 = wR - c12b;                                                               // :96
        const float gHat1 = rhs1 * Q.invLhs1[k];
        const float in1 = Q.lC11[k] - (gHat1 * gHat1) * Q.C11[k];                                   // :105
        const double ld1 = (double)(-0.5f * in1) + lp1;
        const bool take = (ld0 - ld1) < thr[k];                                                     // :107-111
        if (k == 0) bn[k] = take ? (float)((double)gHat1 + H.zs1[k]) : H.b0_excl;
        else bn[k] = (float)((double)(take ? gHat1 : gHat0) + (take ? H.zs1[k] : H.zs0[k]));      // (one value chain: the branch taken)
        dn[k] = take ? 1.f : 0.f;
        dcomb[k] = dn[k];
        all1 = all1 && take;
    }
    if (all1) mt1_linear_beta<NT>(Al, cl, w, bn);                                                   // Rule L
#pragma unroll
    for (int k = 0; k < NT; ++k) Dl[k] = (dn[k] != 0.f) ? a_in[k] - bn[k] : a_in[k];
}

// megaBayesABC! (BayesABC.jl:1-8): trait k is an independent single-trait BayesC update (BayesABC.jl:24-58)
template <int NT>
__device__ __forceinline__ void mega_eval(const MtConsts<NT>& K, const MtPre<NT>& Q, const float (&w)[NT], float dj,
                                          const double (&thr)[NT], const double (&z)[NT],
                                          float (&an)[NT], float (&bn)[NT], float (&dn)[NT], float (&Dl)[NT])
{
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const float rhs    = w[k] * K.ie[k];                                        // :36
        const float invLhs = Q.invLhs1[k];                                          // :37-38
        const float gHat   = rhs * invLhs;                                          // :39
        const float inner  = Q.lC11[k] - gHat * rhs;                                // (log lhs + log var) - gHat*rhs
        const double l1    = -0.5 * (double)inner + K.lp1[k];                       // :40
        if ((K.lp0[k] - l1) < thr[k]) {                                             // :41,:44
            dn[k] = 1.f;
            bn[k] = (float)((double)gHat + z[k] * (double)Q.s1[k]);                 // :46
            Dl[k] = an[k] - bn[k];
            an[k] = bn[k];
        } else {
            dn[k] = 0.f;
            bn[k] = (float)(z[k] * (double)K.sv[k]);                                // :54
            Dl[k] = an[k];
            an[k] = 0.f;
        }
    }
    (void)dj;
}

// lower Cholesky factor of an SPD NT x NT matrix (fixed operation order, shared with the oracle's chol_lower)
template <int NT>
__device__ __forceinline__ void chol_lower(const double (&A)[NT][NT], double (&L)[NT][NT])
{
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        double s = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) s = s - L[j][k] * L[j][k];
        L[j][j] = sqrt(s);
#pragma unroll
        for (int i = j + 1; i < NT; ++i) {
            double v = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v = v - L[i][k] * L[j][k];
            L[i][j] = v / L[j][j];
        }
    }
}

// ---- multi-trait BayesA/B: one InverseWishart(df, scale + b_j b_j') draw per marker (variance_components.jl:181-186:
// sample_variance(data, 1, df, scale) per marker -- the host's 100 000 draws per iteration were ~35 ms of numpy and made
// the multi-trait BayesB iteration host-bound).  One thread per marker, Bartlett's decomposition on the counter RNG:
//   S = scale + b b' = C C' (chol_lower);  A lower-triangular, A_ii = sqrt(chi2(df - i)), A_ik ~ N(0,1) (k < i);
//   K' = A^-1 C' (forward substitution);  G = K K', symmetrised, rounded to float.
// W = C'^-1 A A' C^-1 ~ Wishart(df, S^-1) and G = W^-1.  Counter of a draw: (global marker, iteration, 0x80000000 | attempt,
// slot): slot 32 + 2i (+1) the chi-square of row i (Marsaglia-Tsang gamma: one normal + one uniform per attempt), slot
// 64 + 4i + k the normal A_ik -- disjoint from the sweep's own draws (repetition index < 2^31, slots 0 / 1 + 16 trait).
// Operation for operation the oracle's orc_sample_marker_covariances.
__device__ __forceinline__ double iw_chi2(uint32_t marker, uint32_t iter, uint32_t slot, uint32_t k0, uint32_t k1, double nu)
{
    double a = 0.5 * nu, boost = 1.0;
    if (a < 1.0) {                                   // gamma(a) = gamma(a + 1) * u^(1/a)
        const u32x4 w = philox4x32_10(marker, iter, 0x80000000u | 0xFFFFu, slot, k0, k1);
        boost = exp(log(u52(w.x, w.y)) / a);
        a = a + 1.0;
    }
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    double g = d;
    for (uint32_t attempt = 0; attempt < 64u; ++attempt) {
        const u32x4 w = philox4x32_10(marker, iter, 0x80000000u | attempt, slot, k0, k1);
        const u32x4 w2 = philox4x32_10(marker, iter, 0x80000000u | attempt, slot + 1u, k0, k1);
        const double x = sqrt(-2.0 * log(u52(w.x, w.y))) * cos(6.283185307179586476925286766559 * u52(w.z, w.w));
        const double u = u52(w2.x, w2.y);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        g = d * v;
        if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) break;
    }
    return 2.0 * g * boost;
}

struct IwParams { double df; double scale[kMaxT * kMaxT]; uint32_t seed_lo, seed_hi, iter, marker0; int diagonal; };

template <int NT>
__global__ __launch_bounds__(256) void k_sample_marker_covariances(IwParams Q, int64_t p, const float* __restrict__ beta, float* __restrict__ var_mat)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= p) return;
    const uint32_t marker = Q.marker0 + (uint32_t)j;
    double b[NT], S[NT][NT], C[NT][NT], A[NT][NT], Kt[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) b[a] = (double)beta[(int64_t)a * p + j];
    if (Q.diagonal) {       // constraint = true (variance_components.jl:112-117): G_kk = (scale_kk + b_k^2) / chi2(df), zeros elsewhere
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const double g = (Q.scale[a * NT + a] + b[a] * b[a]) / iw_chi2(marker, Q.iter, 32u + 2u * (uint32_t)a, Q.seed_lo, Q.seed_hi, Q.df);
#pragma unroll
            for (int c = 0; c < NT; ++c) var_mat[(j * NT + a) * NT + c] = (c == a) ? (float)g : 0.0f;
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c) { S[a][c] = Q.scale[a * NT + c] + b[a] * b[c]; C[a][c] = 0.0; A[a][c] = 0.0; }
    chol_lower<NT>(S, C);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        A[i][i] = sqrt(iw_chi2(marker, Q.iter, 32u + 2u * (uint32_t)i, Q.seed_lo, Q.seed_hi, Q.df - (double)i));
#pragma unroll
        for (int k = 0; k < i; ++k) {
            const u32x4 w = philox4x32_10(marker, Q.iter, 0x80000000u, 64u + 4u * (uint32_t)i + (uint32_t)k, Q.seed_lo, Q.seed_hi);
            A[i][k] = sqrt(-2.0 * log(u52(w.x, w.y))) * cos(6.283185307179586476925286766559 * u52(w.z, w.w));
        }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            double acc = C[c][i];                                       // C'[i][c]
#pragma unroll
            for (int k = 0; k < i; ++k) acc = acc - A[i][k] * Kt[k][c];
            Kt[i][c] = acc / A[i][i];
        }
    double G[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < NT; ++i) s = s + Kt[i][a] * Kt[i][c];
            G[a][c] = s;
        }
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c) var_mat[(j * NT + a) * NT + c] = (float)(0.5 * (G[a][c] + G[c][a]));
}

// Gibbs sampler II, one candidate state (MTBayesABC.jl:178-185).  st: bit k = trait k in the model.
// q = -0.5*(log det lhs - rhs'gHat); cand = gHat + chol(lhs^-1)*z only when want_cand.
// The evaluation of one state is split in three: the part that depends on the marker's x'x and the sweep's variances
// only (lhs, its inverse and log determinant -- both Cholesky factorisations' worth of divisions and square roots),
// the part that depends on the running rhs (a handful of multiply-adds), and the candidate effects of the chosen state.
// mt2_state = pre + post (+ cand): one operation order, shared with the oracle's mt2_state.
template <int NT>
__device__ __forceinline__ void mt2_state_pre(const MtConsts<NT>& K, unsigned st, float dj, double (&inv)[NT][NT], double& logdet)
{
    double lhs[NT][NT], L[NT][NT], M[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        const double Da = ((st >> a) & 1u) ? 1.0 : 0.0;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const double Dc = ((st >> c) & 1u) ? 1.0 : 0.0;
            const double rl = (Da * (double)K.Rinv[a][c]) * Dc;                     // D*Rinv*D  :159
            lhs[a][c] = rl * (double)dj + (double)K.Ginv[a][c];                     // :179
        }
    }
    chol_lower<NT>(lhs, L);
#pragma unroll
    for (int j = 0; j < NT; ++j) {                                                  // M = L^-1
        M[j][j] = 1.0 / L[j][j];
#pragma unroll
        for (int i = j + 1; i < NT; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = j; k < i; ++k) s = s + L[i][k] * M[k][j];
            M[i][j] = -s / L[i][i];
        }
    }
#pragma unroll
    for (int a = 0; a < NT; ++a)                                                    // inv(lhs) = M'M  :181
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            double s = 0.0;
#pragma unroll
            for (int k = (a > c ? a : c); k < NT; ++k) s = s + M[k][a] * M[k][c];
            inv[a][c] = s;                                                          // (bitwise symmetric: products commute)
        }
    double det = 1.0;
#pragma unroll
    for (int j = 0; j < NT; ++j) det = det * (L[j][j] * L[j][j]);
    logdet = log(det);
}
template <int NT>
__device__ __forceinline__ void mt2_state_post(const MtConsts<NT>& K, unsigned st, const float (&w)[NT],
                                               const double (&inv)[NT][NT], double logdet, double& q, double (&gHat)[NT])
{
    double rhs[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        const double Da = ((st >> a) & 1u) ? 1.0 : 0.0;
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < NT; ++m) s = s + ((double)K.Rinv[m][a] * Da) * (double)w[m];   // (Rinv*D)'w :180
        rhs[a] = s;
    }
    double quad = 0.0;
#pragma unroll
    for (int a = 0; a < NT; ++a) {                                                  // gHat = invLhs*rhs :183
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < NT; ++c) s = s + inv[a][c] * rhs[c];
        gHat[a] = s;
        quad = quad + rhs[a] * s;
    }
    q = -0.5 * (logdet - quad);                                                     // :184
}
template <int NT>
__device__ __forceinline__ void mt2_state_cand(const double (&inv)[NT][NT], const double (&gHat)[NT], const double (&z)[NT],
                                               double (&cand)[NT])
{
    double C[NT][NT];
    chol_lower<NT>(inv, C);                                                         // cholesky(Hermitian(invLhs)).L :182
#pragma unroll
    for (int a = 0; a < NT; ++a) {                                                  // gHat + L*z  :185
        double s = gHat[a];
#pragma unroll
        for (int c = 0; c <= a; ++c) s = s + C[a][c] * z[c];
        cand[a] = s;
    }
}
template <int NT>
__device__ __forceinline__ void mt2_state(const MtConsts<NT>& K, unsigned st, const float (&w)[NT], float dj,
                                          const double (&z)[NT], bool want_cand, double& q, double (&cand)[NT])
{
    double inv[NT][NT], gHat[NT], logdet;
    mt2_state_pre<NT>(K, st, dj, inv, logdet);
    mt2_state_post<NT>(K, st, w, inv, logdet, q, gHat);
    if (want_cand) mt2_state_cand<NT>(inv, gHat, z, cand);
}

// Per-marker table of the state-dependent, rhs-independent quantities (sampler II, NT <= 3): for each of the 2^NT
// states the NT(NT+1)/2 unique entries of inv(lhs) (row-major upper triangle) and log det lhs.  Filled once per sweep
// for all markers in parallel (k_prepare_mt2); layout [state][value][p].
template <int NT>
struct Mt2Tab {
    static constexpr int NS = 1 << NT, NV = NT * (NT + 1) / 2 + 1, kRows = NS * NV;
};
template <int NT>
__device__ __forceinline__ void mt2_unpack(const double (&row)[Mt2Tab<NT>::NV], double (&inv)[NT][NT], double& logdet)
{
    int v = 0;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int c = a; c < NT; ++c) { inv[a][c] = row[v]; inv[c][a] = row[v]; ++v; }
    logdet = row[v];
}
template <int NT>
__global__ __launch_bounds__(256) void k_prepare_mt2(const DevParams* __restrict__ P, int64_t p, const float* __restrict__ xpx,
                                                     double* __restrict__ tab)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= p) return;
    MtConsts<NT> K;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c) { K.Rinv[a][c] = P->Rinv[a * NT + c]; K.Ginv[a][c] = P->Ginv[a * NT + c]; }
    const float dj = xpx[j];
#pragma unroll 1
    for (int st = 0; st < Mt2Tab<NT>::NS; ++st) {
        double inv[NT][NT], logdet;
        mt2_state_pre<NT>(K, (unsigned)st, dj, inv, logdet);
        int v = 0;
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int c = a; c < NT; ++c) { tab[((int64_t)st * Mt2Tab<NT>::NV + v) * p + j] = inv[a][c]; ++v; }
        tab[((int64_t)st * Mt2Tab<NT>::NV + v) * p + j] = logdet;
    }
}
template <int NT>
__device__ __forceinline__ void mt2_load_tab(const double* __restrict__ tab, int64_t p, int64_t j,
                                             double (&T)[Mt2Tab<NT>::NS][Mt2Tab<NT>::NV])
{
#pragma unroll
    for (int st = 0; st < Mt2Tab<NT>::NS; ++st)
#pragma unroll
        for (int v = 0; v < Mt2Tab<NT>::NV; ++v) T[st][v] = tab[((int64_t)st * Mt2Tab<NT>::NV + v) * p + j];
}

// Gibbs sampler II, one marker, from its state table (same results as mt2_eval).
template <int NT>
__device__ __forceinline__ void mt2_eval_tab(const MtConsts<NT>& K, const double* lpr, int ls, const float (&w)[NT],
                                             const double (&T)[Mt2Tab<NT>::NS][Mt2Tab<NT>::NV],
                                             double u, const double (&z)[NT],
                                             float (&an)[NT], float (&bn)[NT], float (&dn)[NT], float (&Dl)[NT])
{
    constexpr int NS = Mt2Tab<NT>::NS, NV = Mt2Tab<NT>::NV;
    double ld[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        double inv[NT][NT], lg, q, gh[NT];
        mt2_unpack<NT>(T[s], inv, lg);
        mt2_state_post<NT>(K, (unsigned)s, w, inv, lg, q, gh);
        ld[s] = q + lpr[s * ls];
    }
    int which = NS - 1;
    {                                                                               // :188-198
        double mx = -INFINITY;
#pragma unroll
        for (int s = 0; s < NS; ++s) if (ld[s] > mx) mx = ld[s];
        double den = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) { ld[s] = exp(ld[s] - mx); den += ld[s]; }
        double cp = 0.0;
        bool found = false;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            cp += ld[s] / den;
            if (!found && u < cp) { which = s; found = true; }
        }
    }
    double row[NV];                                                                 // the chosen state's row: select chain
#pragma unroll
    for (int v = 0; v < NV; ++v) row[v] = T[0][v];
#pragma unroll
    for (int s = 1; s < NS; ++s)
#pragma unroll
        for (int v = 0; v < NV; ++v) row[v] = (which == s) ? T[s][v] : row[v];
    double inv[NT][NT], lg, q, gh[NT], cand[NT];
    mt2_unpack<NT>(row, inv, lg);
    mt2_state_post<NT>(K, (unsigned)which, w, inv, lg, q, gh);
    mt2_state_cand<NT>(inv, gh, z, cand);
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const double dk = ((which >> k) & 1) ? 1.0 : 0.0;
        const double a_new = dk * cand[k];                                          // diagm(delta)*beta :201
        Dl[k] = (float)((double)an[k] - a_new);                                     // oldα-newα -> axpy :204
        bn[k] = (float)cand[k];
        dn[k] = (float)dk;
        an[k] = (float)a_new;
    }
}

// Gibbs sampler II, one marker (MTBayesABC.jl:160-208).  u = the marker's uniform (slot 0).
template <int NT>
__device__ __forceinline__ void mt2_eval(const MtConsts<NT>& K, const double* lpr, int ls, const float (&w)[NT], float dj,
                                         double u, const double (&z)[NT],
                                         float (&an)[NT], float (&bn)[NT], float (&dn)[NT], float (&Dl)[NT])
{
    constexpr int NS = 1 << NT;
    double ld[NS], cand[NT];
#pragma unroll
    for (int s = 0; s < NS; ++s) ld[s] = 0.0;
    int which = NS - 1;
    // passes 0..NS-1 evaluate the states; pass NS re-evaluates the chosen one for its candidate effects
#pragma unroll 1
    for (int pass = 0; pass <= NS; ++pass) {
        const unsigned st = pass < NS ? (unsigned)pass : (unsigned)which;
        double q;
        mt2_state<NT>(K, st, w, dj, z, pass == NS, q, cand);
        if (pass < NS) {
            const double v = q + lpr[pass * ls];
#pragma unroll
            for (int s = 0; s < NS; ++s) ld[s] = (s == pass) ? v : ld[s];
        }
        if (pass == NS - 1) {                                                       // :188-198
            double mx = -INFINITY;
#pragma unroll
            for (int s = 0; s < NS; ++s) if (ld[s] > mx) mx = ld[s];
            double den = 0.0;
#pragma unroll
            for (int s = 0; s < NS; ++s) { ld[s] = exp(ld[s] - mx); den += ld[s]; }
            double cp = 0.0;
            bool found = false;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                cp += ld[s] / den;
                if (!found && u < cp) { which = s; found = true; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const double dk = ((which >> k) & 1) ? 1.0 : 0.0;
        const double a_new = dk * cand[k];                                          // diagm(delta)*beta :201
        Dl[k] = (float)((double)an[k] - a_new);                                     // oldα-newα -> axpy :204
        bn[k] = (float)cand[k];
        dn[k] = (float)dk;
        an[k] = (float)a_new;
    }
}

// ---------------------------------------------------------------------------------------------
// SAMPLER role, multi-trait: Gibbs sampler I (MTBayesABC.jl:57-127, block form :243-333), sampler II
// (:129-210) and megaBayesABC! (BayesABC.jl:1-8) share the schedule; only the per-marker evaluation differs.
// ---------------------------------------------------------------------------------------------
// DW: the instantiation for sweeps in which (nearly) every marker is in the model (selected by the host from the previous
// sweep's change count, for single-pass sweeps over <= 128-marker blocks): EVERY block takes the dense walk -- which is exact
// for any state, markers outside the model are simply evaluated the general way at their step -- so the candidacy evaluation
// of the front, the prefix skip, row staging and the speculative rounds are compiled out.  Same results as the general
// instantiation, bit for bit (the walk and the rounds are the same chain); the point is the CODE SIZE: the front is ~3 000
// cold instructions per launch, fetched at memory latency.
// ---- DENSE blocks of 256 markers, sampler I (one shared effect covariance, or one per marker: multi-trait BayesA/B) under a
// prior that keeps (nearly) every marker in the model for every trait -- the reference's default multi-trait prior.  As dense_big_st: the block chain is a forward substitution; 64-marker SECTION s is
// walked by wave s from its strictly-upper DIAGONAL Gram tile in LDS exactly as the 128-marker dense walk walks a section --
// Rule L's linear form speculatively, one full evaluation per lane afterwards that verifies the speculation and yields the
// final state, the section walked again from its saved rhs with the offending markers evaluated the general way on a miss --
// and everything off the diagonal runs in parallel: thread c (waves 0..3) owns marker c and its running rhs of every trait
// in registers, thread 256 + c' (waves 4..7) owns column c' of the NEXT block's lookahead correction, and after section s is
// done both apply its 64 changes per trait from Gram / cross-Gram values they prefetched into registers while the section was
// being walked (fmaf in marker order: the sequential chain's own sequence -- bit-identical to the general path).  Half the
// launches (and fronts) of the 128-marker blocks, and the walking wave updates its own section only.
// Reference: MTBayesABC.jl:243-333 (block form of _MTBayesABC_samplerI!).
template <int METHOD, int NT>
__device__ __forceinline__ void dense_big_mt(char* smem, const StepSmem& SM, const SamplerArgs& A, const MtConsts<NT>& K,
                                             const double* lpr, long long tk0, long long tk1)
{
    static_assert(is_sampler1(METHOD), "sampler I (K: the constants of THIS thread's marker -- the shared ones, or the marker's own under multi-trait BayesA/B)");
    constexpr int kB = 256, kSec = kB / 64;
    static_assert(kStepThreads == 2 * kB, "waves 0..3: markers, waves 4..7: columns of the next block");
    const int B = SM.B, b = A.b, bn = A.b_next;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t j0 = A.j0, p = A.p;
    float* rhs_lds = reinterpret_cast<float*>(smem + SM.rhs_off);       // entry rhs; reused as D[t][c] = alpha_old - alpha_new
    float* acur = reinterpret_cast<float*>(smem + SM.acur_off);
    const float* astart = reinterpret_cast<const float*>(smem + SM.astart_off);
    float* bcur = reinterpret_cast<float*>(smem + SM.bcur_off);
    float* dcur = reinterpret_cast<float*>(smem + SM.dcur_off);
    const double* lpd = reinterpret_cast<const double*>(smem + SM.prepd_off);
    const float* lpf = reinterpret_cast<const float*>(smem + SM.prepf_off);
    const float* tiles = reinterpret_cast<const float*>(smem + SM.rows_off);      // [4][64][64], strictly upper
    int* wcnt = reinterpret_cast<int*>(smem + SM.wcnt_off);
    float* delta = reinterpret_cast<float*>(A.delta);
    const bool rowthr = tid < kB;                       // marker c = tid
    const bool colthr = !rowthr && (tid - kB) < bn;     // column c' = tid - 256 of the next block
    const int c = rowthr ? tid : 0;
    const int cn = colthr ? tid - kB : 0;
    // the marker's state, draws and constants (parked in LDS by the front)
    float rhs[NT], a[NT], bb[NT], dd[NT], lc[NT], corr[NT];
    double thr[NT], z[NT];
    const float dj = lpf[c];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        rhs[t] = rhs_lds[t * B + c]; a[t] = acur[t * B + c]; bb[t] = bcur[t * B + c]; dd[t] = dcur[t * B + c];
        thr[t] = lpd[t * B + c]; z[t] = lpd[(NT + t) * B + c]; lc[t] = lpf[(1 + t) * B + c];
        corr[t] = 0.f;
    }
    const MtPre<NT> Q = mt_precompute<METHOD, NT>(K, dj, lc);
    float an[NT], bnw[NT], dn[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { an[t] = a[t]; bnw[t] = bb[t]; dn[t] = dd[t]; }
    int nredo = 0;
    float pq[64];
    // thread c reads column c of the Gram rows (waves after the section) / column c' of the cross-Gram rows of a section: one
    // dword per lane, coalesced; the row pointer is uniform
    auto load_g = [&](int s) {
        const char* base = reinterpret_cast<const char*>(A.gram + (int64_t)(64 * s) * b);
        unsigned off = 4u * (unsigned)c;
#pragma unroll
        for (int u = 0; u < 64; ++u) { pq[u] = *reinterpret_cast<const float*>(base + off); off += 4u * (unsigned)b; asm volatile("" : "+v"(off)); }
    };
    auto load_c = [&](int s) {
        const char* base = reinterpret_cast<const char*>(A.cross_next + (int64_t)(64 * s) * bn);
        unsigned off = 4u * (unsigned)cn;
#pragma unroll
        for (int u = 0; u < 64; ++u) { pq[u] = *reinterpret_cast<const float*>(base + off); off += 4u * (unsigned)bn; asm volatile("" : "+v"(off)); }
    };
    auto prefetch = [&](int s) {
        if (rowthr && wave > s) load_g(s);
        else if (colthr) load_c(s);
    };
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    prefetch(0);
#pragma unroll 1
    for (int s = 0; s < kSec; ++s) {
        if (wave == s) {
            // ---- walk section s (lane = marker 64 s + lane), diagonal tile s: stride 64
            const float* tile = tiles + s * 4096;
            float Al[NT][NT], cl[NT], da[NT], rs[NT], wev[NT];
            mt1_linear_coeffs<NT>(K, Q, dj, bb, z, Al, cl);
#pragma unroll
            for (int t = 0; t < NT; ++t) { da[t] = dj * a[t]; rs[t] = rhs[t]; }                       // MTBayesABC.jl:82
            auto eval_own = [&](const float (&w)[NT], float (&ao)[NT], float (&bo)[NT], float (&d_o)[NT], float (&Dl)[NT]) {
#pragma unroll
                for (int t = 0; t < NT; ++t) { ao[t] = a[t]; bo[t] = bb[t]; d_o[t] = dd[t]; Dl[t] = 0.f; }
                mt1_eval<NT>(K, Q, PriorMem{lpr, 1}, w, dj, thr, z, ao, bo, d_o, Dl, Al, cl);
            };
            auto bcast = [&](int l, const float (&Dl)[NT], float g) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float D = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Dl[t]), l));
                    rhs[t] = fmaf(D, g, rhs[t]);                                // (strictly upper tile: lanes <= l are not moved)
                }
            };
            auto step_fast = [&](int l, float g) {
                float w[NT], bo[NT], Dl[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) w[t] = rhs[t] + da[t];
                mt1_linear_beta<NT>(Al, cl, w, bo);
#pragma unroll
                for (int t = 0; t < NT; ++t) Dl[t] = a[t] - bo[t];
                bcast(l, Dl, g);
            };
            auto walk_fast = [&]() {
                float gn[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) gn[u] = tile[u * 64 + lane];
#pragma unroll 1
                for (int l = 0; l < 64; l += 8) {
                    float g[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) g[u] = gn[u];
                    if (l + 8 < 64) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) gn[u] = tile[(l + 8 + u) * 64 + lane];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) step_fast(l + u, g[u]);
                }
            };
            // (the rhs-independent part of the literal evaluation, once per marker: mt1_hoist)
            const Mt1Hoist<NT> Hh = mt1_hoist<NT>(K, Q, PriorMem{lpr, 1}, dj, bb, dd, z);
            // skip: markers that take NO step -- out of the model for every trait at entry and predicted to stay there (below): their
            // alpha_old - alpha_new is 0 for every trait, and applying a zero change is an exact no-op on every running rhs, so leaving the
            // step out changes no bit.  The prediction is verified like everything else by the literal evaluation after the walk.
            auto walk_mixed = [&](unsigned long long slow, unsigned long long skip) {
                unsigned long long todo = ~skip;
                if (todo == 0ull) return;
                int l = __builtin_ctzll(todo);
                float g = tile[l * 64 + lane];
#pragma unroll 1
                while (true) {
                    todo &= todo - 1ull;
                    const int ln = todo ? __builtin_ctzll(todo) : 63;
                    float w[NT], bo[NT], d_o[NT], Dl[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) w[t] = rhs[t] + da[t];
                    if ((slow >> l) & 1ull) mt1_eval_hoisted<NT>(K, Q, Hh, PriorMem{lpr, 1}, w, a, dd, thr, Al, cl, bo, d_o, Dl);      // (wave-uniform)
                    else {
                        mt1_linear_beta<NT>(Al, cl, w, bo);
#pragma unroll
                        for (int t = 0; t < NT; ++t) Dl[t] = a[t] - bo[t];
                    }
                    const float gl = g;
                    g = tile[ln * 64 + lane];
                    bcast(l, Dl, gl);
                    if (todo == 0ull) break;
                    l = ln;
                }
            };
            bool in_all = true, out_all = true;
#pragma unroll
            for (int t = 0; t < NT; ++t) { in_all = in_all && (dd[t] == 1.f); out_all = out_all && (dd[t] == 0.f) && (a[t] == 0.f); }
            unsigned long long slow = __ballot(!in_all), skip = 0ull;
            if (__any(out_all)) {
                // markers outside the model: one evaluation against the SECTION-ENTRY rhs predicts which of them stay out (the changes
                // inside a section move a marker's rhs by a few percent of what it takes to bring it in)
                float w0[NT], bo[NT], d_o[NT], Dp[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) w0[t] = rhs[t] + da[t];
                mt1_eval_hoisted<NT>(K, Q, Hh, PriorMem{lpr, 1}, w0, a, dd, thr, Al, cl, bo, d_o, Dp);
                bool stays = out_all;
#pragma unroll
                for (int t = 0; t < NT; ++t) stays = stays && (d_o[t] == 0.f);
                skip = __ballot(stays);
                slow &= ~skip;
            }
            if (__popcll(slow) * 4 > 64) slow = ~skip;
            if (slow == 0ull && skip == 0ull) walk_fast(); else walk_mixed(slow, skip);
            float Dl[NT];
            for (int pass = 0; pass < 64; ++pass) {
#pragma unroll
                for (int t = 0; t < NT; ++t) wev[t] = rhs[t] + da[t];                                 // what the lane's marker was evaluated with
                eval_own(wev, an, bnw, dn, Dl);
                bool ok = true, moved = false;
#pragma unroll
                for (int t = 0; t < NT; ++t) { ok = ok && (dn[t] == 1.f); moved = moved || (Dl[t] != 0.f); }
                // a marker walked with the linear form must have stayed in the model for every trait; one that took no step must not have moved
                const unsigned long long bad = (__ballot(!ok) & ~slow & ~skip) | (__ballot(moved) & skip);
                if (bad == 0ull) break;
                slow |= bad; skip &= ~bad;
                if (__popcll(slow) * 4 > 64) slow = ~skip;
#pragma unroll
                for (int t = 0; t < NT; ++t) rhs[t] = rs[t];
                walk_mixed(slow, skip);
                ++nredo;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acur[t * B + c] = an[t]; bcur[t * B + c] = bnw[t]; dcur[t * B + c] = dn[t];
                rhs_lds[t * B + c] = a[t] - an[t];                              // D of this marker, read by everybody after the barrier
            }
        }
        lds_barrier();
        // the section's changes: fmaf chains in marker order, traits interleaved; 8 broadcast reads per trait and batch
        const bool do_r = rowthr && wave > s;
        if (do_r || colthr) {
#pragma unroll
            for (int k0 = 0; k0 < 64; k0 += 8) {
                float dv[NT][8];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4 d0 = *reinterpret_cast<const float4*>(rhs_lds + t * B + 64 * s + k0);
                    const float4 d1 = *reinterpret_cast<const float4*>(rhs_lds + t * B + 64 * s + k0 + 4);
                    dv[t][0] = d0.x; dv[t][1] = d0.y; dv[t][2] = d0.z; dv[t][3] = d0.w; dv[t][4] = d1.x; dv[t][5] = d1.y; dv[t][6] = d1.z; dv[t][7] = d1.w;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if (do_r) rhs[t] = fmaf(dv[t][u], pq[k0 + u], rhs[t]);
                        else corr[t] = fmaf(dv[t][u], pq[k0 + u], corr[t]);
                    }
                }
            }
        }
        if (s + 1 < kSec) prefetch(s + 1);
    }
    __syncthreads();
    const long long tk4 = clock64();
    // the block's change list in marker order
    bool changed = false;
    if (rowthr) {
#pragma unroll
        for (int t = 0; t < NT; ++t) changed = changed || (astart[t * B + c] != acur[t * B + c]);
    }
    const unsigned long long cm = __ballot(changed);
    if (lane == 0) wcnt[wave] = __popcll(cm);
    __syncthreads();
    int base = 0, nfin = 0;
#pragma unroll
    for (int q = 0; q < kStepThreads / 64; ++q) { const int v = wcnt[q]; base += (q < wave) ? v : 0; nfin += v; }
    // ---- global stores last
    if (!rowthr && bn > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) A.corr_out[t * B + (tid - kB)] = colthr ? corr[t] : 0.f;
    }
    if (changed) {
        const int e = base + __popcll(cm & ((1ull << lane) - 1ull));
        A.ev_out->idx[e] = (int32_t)(j0 + c);
#pragma unroll
        for (int t = 0; t < NT; ++t) A.ev_out->delta[t][e] = astart[t * B + c] - acur[t * B + c];
    }
    if (rowthr) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float a_fin = acur[t * B + c];
            if (a_fin != astart[t * B + c]) A.alpha[(int64_t)t * p + j0 + c] = a_fin;
            A.beta[(int64_t)t * p + j0 + c] = bcur[t * B + c];
            delta[(int64_t)t * p + j0 + c] = dcur[t * B + c];
        }
    }
    if (tid == 0) {
        A.ev_out->count = (int32_t)nfin;
        atomicAdd(&A.counters[0], (unsigned long long)nfin);
        atomicAdd(&A.counters[2], (unsigned long long)(tk1 - tk0));      // front
        atomicAdd(&A.counters[5], (unsigned long long)(tk4 - tk1));      // sections (walks + off-diagonal applies)
    }
    if (lane == 0 && nredo) atomicAdd(&A.counters[7], (unsigned long long)nredo);      // sections walked again
}

// ---- dense_big_mt under RULE T (jwas_sweep_params.section_solve; the comment block above k_section_inverse_mt).  The same
// plan -- thread c (waves 0..3) owns marker c, one 64-marker section after the other; the next block's lookahead correction is
// formed by a HELPER workgroup (corr_helper, sampler_common.hpp) from the changes each section publishes -- but a section in which every marker
// is in the model for every trait is SOLVED, not walked: wave s forms y (one evaluation of the linear form per lane), waves 4..7 multiply it with the section's inverse (wave
// 4 + q: the q-th quarter of the columns for all NT rows of every marker, one float4 of T per lane, trait and column group,
// fetched a section ahead), wave s adds the four partial products, forms the new effects and verifies them with ONE literal
// evaluation per lane while everybody else already applies the section's changes (undone if the verification fails: the
// section is then walked as in dense_big_mt).  The two roles run SEPARATE loops with the same sequence of barriers, so that the
// registers of one role (the walker's state and Gram prefetch / the solver's 4 NT^2 float4 of T) are not live in the other.
// NT <= 3 (4 traits: 64 float4 of T per lane).
template <int METHOD, int NT, class KF>
__device__ __forceinline__ void dense_big_mt_solve(char* smem, const StepSmem& SM, const SamplerArgs& A, const KF& consts_of_marker,
                                                   const double* lpr, long long tk0, long long tk1)
{
    static_assert(is_sampler1(METHOD) && NT <= 3, "sampler I, at most three traits");
    constexpr int kB = 256, kSec = kB / 64;
    static_assert(kStepThreads == 2 * kB, "waves 0..3: markers, waves 4..7: the solve");
    const int B = SM.B, b = A.b, bn = A.b_next;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t j0 = A.j0, p = A.p;
    float* rhs_lds = reinterpret_cast<float*>(smem + SM.rhs_off);       // entry rhs; reused as D[t][c] = alpha_old - alpha_new
    float* acur = reinterpret_cast<float*>(smem + SM.acur_off);
    const float* astart = reinterpret_cast<const float*>(smem + SM.astart_off);
    float* bcur = reinterpret_cast<float*>(smem + SM.bcur_off);
    float* dcur = reinterpret_cast<float*>(smem + SM.dcur_off);
    const double* lpd = reinterpret_cast<const double*>(smem + SM.prepd_off);
    const float* lpf = reinterpret_cast<const float*>(smem + SM.prepf_off);
    const float* tiles = reinterpret_cast<const float*>(smem + SM.rows_off);      // [4][64][64], strictly upper (the walk's)
    // scratch behind the tiles: y of the section [NT][64], the four partial products [4][NT][64], flags
    float* ybuf = reinterpret_cast<float*>(smem + SM.rows_off) + kSec * 4096;
    float* part = ybuf + NT * 64;
    int* sflag = reinterpret_cast<int*>(part + 4 * NT * 64);           // [s]: section s is solved; [4 + s]: ... and its verification failed
    float* tcol = reinterpret_cast<float*>(sflag + 8);                 // [kSolveMaxOdd][NT][NT][64]: the columns of T_s of the section's markers outside the model at entry
    int* wcnt = reinterpret_cast<int*>(smem + SM.wcnt_off);
    float* delta = reinterpret_cast<float*>(A.delta);
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    typedef float v4f __attribute__((ext_vector_type(4)));
    // D of section s (rhs_lds) times 64 prefetched Gram values: fmaf chains in marker order, traits interleaved
    auto apply_changes = [&](int s, float (&acc)[NT], const float (&pq)[64]) { apply_section_changes<NT>(rhs_lds + 64 * s, B, acc, pq); };
    // ---- the mat-vec D~ = T y: waves 4..7, wave 4 + q the q-th quarter of the columns (4 NT column groups of four) for the NT rows of
    // every marker.  (All eight waves with an eighth each -- twice the issue rate -- was measured: the 147 KB of a section's T pass
    // the CU's vector memory pipe at 64 B/clk = 2.3 k cycles, and a wave that is stuck issuing loads is late at the next barrier;
    // the solver waves issue theirs while the verification / the apply run and nobody waits for them.)
    constexpr int QG = 4 * NT;                                           // column groups of four per quarter
    const int w8 = __builtin_amdgcn_readfirstlane(wave & 3);            // (wave-uniform, and the compiler knows)
    // (g0, g1: the wave's column groups [g0, g1) only -- the solver waves fetch a section's T in four pieces, one per barrier
    // interval, so that no interval carries the whole 2.3 k cycles the 147 KB take through the CU's vector memory pipe)
    auto load_t = [&](int s, v4f (&tq)[NT][QG], int g0 = 0, int g1 = 64) {
        // buffer loads: ONE vector register (the lane's 16-byte offset) addresses all of them, the (group, trait) offset is a
        // scalar operand
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(A.tsec + (int64_t)s * tsec_floats<NT>() + (int64_t)(w8 * QG) * (64 * NT) * 4), 0, 0x7fffffff, 0x00020000);
        const unsigned voff = 16u * (unsigned)lane;
#pragma unroll
        for (int g = 0; g < QG; ++g)
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                if (g < g0 || g >= g1) continue;                         // (compile-time after inlining: g0 / g1 are literals at every call)
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)((g * (64 * NT) + k * 64) * 16), 0);
                tq[k][g] = __builtin_bit_cast(v4f, raw);
            }
    };
    auto matvec = [&](const v4f (&tq)[NT][QG]) {
        float acc[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) acc[k] = 0.f;
        const v4f* y4 = reinterpret_cast<const v4f*>(ybuf) + w8 * QG;
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            const v4f yv = y4[g];                                        // (broadcast read: columns 4 (w QG + g) .. + 3)
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                acc[k] = fmaf(tq[k][g].x, yv.x, acc[k]); acc[k] = fmaf(tq[k][g].y, yv.y, acc[k]);
                acc[k] = fmaf(tq[k][g].z, yv.z, acc[k]); acc[k] = fmaf(tq[k][g].w, yv.w, acc[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < NT; ++k) part[(w8 * NT + k) * 64 + lane] = acc[k];
    };
    if (wave < 4) {
        // =================================== markers (thread c = marker c) ===================================
        const int c = tid;
        const MtConsts<NT> K = consts_of_marker(c);                     // (the shared constants, or the marker's own: multi-trait BayesA/B)
        float rhs[NT], a[NT], bb[NT], dd[NT];
        const float dj = lpf[c];
        bool in_all = true;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            rhs[t] = rhs_lds[t * B + c]; a[t] = acur[t * B + c]; bb[t] = bcur[t * B + c]; dd[t] = dcur[t * B + c];
            in_all = in_all && (dd[t] == 1.f);
        }
        // the marker's draws and x'x-only terms are needed by its own evaluations only (the coefficients below, the verification /
        // the walk of its section): fetched from LDS there, not held in registers across the other sections (the registers
        // carry the wave's eighth of T and its Gram prefetch)
        auto draws_of = [&](double (&thr)[NT], double (&z)[NT], MtPre<NT>& Q) {
            float lc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) { thr[t] = lpd[t * B + c]; z[t] = lpd[(NT + t) * B + c]; lc[t] = lpf[(1 + t) * B + c]; }
            Q = mt_precompute<METHOD, NT>(K, dj, lc);
        };
        float pq[64];
        auto load_g = [&](int s) {
            // column c of the Gram rows of section s: one dword per lane, coalesced; the row pointer is uniform.  (Buffer loads with
            // the row offset as a scalar operand -- one instruction per load instead of two -- were measured: 2.7 k cycles per
            // section instead of 0.5 k.)
            const char* base = reinterpret_cast<const char*>(A.gram + (int64_t)(64 * s) * b);
            unsigned off = 4u * (unsigned)c;
#pragma unroll
            for (int u = 0; u < 64; ++u) { pq[u] = *reinterpret_cast<const float*>(base + off); off += 4u * (unsigned)b; asm volatile("" : "+v"(off)); }
        };
        // The Gram prefetch is dead in the wave that OWNS the running section (it applies nothing any more), but the compiler sees
        // one loop for all waves and keeps the 64 registers alive through the owner's verification, which then spills.  An empty
        // asm that "defines" them at the end of the owner's regions ends their live range there: the registers are free inside.
        auto kill_pq = [&] {
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("" : "=v"(pq[u]));
        };
        if (wave > 0) load_g(0);
        // Rule L's coefficients of the thread's own marker: all four waves at once, before the chain starts
        float Al[NT][NT], cl[NT], da[NT];
        {
            double thr[NT], z[NT];
            MtPre<NT> Q;
            draws_of(thr, z, Q);
            mt1_linear_coeffs<NT>(K, Q, dj, bb, z, Al, cl);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) da[t] = dj * a[t];                                                  // MTBayesABC.jl:82
        {
            const unsigned long long slow0 = __ballot(!in_all);
            if (lane == 0) { sflag[wave] = (__popcll(slow0) <= kSolveMaxOdd) ? 1 : 0; sflag[4 + wave] = 0; }
        }
        lds_barrier();                                                                                  // B0
        int nredo = 0, nsolved = 0, nfailed = 0, nexc_all = 0;
        long long cy_y = 0, cy_mv = 0, cy_cmb = 0, cy_ver = 0, cy_tail = 0, cy_x1 = 0, cy_xn = 0;      // (diagnostics: the phases of a solved section, wave 0's clock)
#pragma unroll 1
        for (int s = 0; s < kSec; ++s) {
            const bool fast = sflag[s] != 0;
            bool redo = !fast;
            const long long ts0 = clock64();
            long long ts4 = ts0;
            if (fast) {
                float rsv[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) rsv[t] = rhs[t];
                if (wave == s) {                                         // y of the section
                    float w[NT], bo[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) w[t] = rhs[t] + da[t];
                    mt1_linear_beta<NT>(Al, cl, w, bo);
#pragma unroll
                    for (int t = 0; t < NT; ++t) ybuf[t * 64 + lane] = a[t] - bo[t];
                }
                lds_barrier();                                           // B1: y is there
                const long long ts1 = clock64();
                // the markers of the section that are outside the model at entry WILL be exceptions: their NT columns of T_s go to LDS
                // now (direct loads, no registers), while this wave has nothing to do -- the correction below then costs no memory latency
                unsigned long long odd0 = 0ull;
                if (wave == s) {
                    odd0 = __ballot(!in_all);
                    typedef __attribute__((address_space(3))) void lds_void;
                    const float* Ts = A.tsec + (int64_t)s * tsec_floats<NT>();
                    int o = 0;
#pragma unroll 1
                    for (unsigned long long m_ = odd0; m_ != 0ull; m_ &= m_ - 1ull, ++o) {
                        const int e = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m_));
#pragma unroll
                        for (int k = 0; k < NT; ++k)
#pragma unroll
                            for (int m = 0; m < NT; ++m) {
                                const int cidx = m * 64 + e;
                                __builtin_amdgcn_global_load_lds(Ts + ((int64_t)(cidx >> 2) * (64 * NT) + (k * 64 + lane)) * 4 + (cidx & 3),
                                                                 (lds_void*)(tcol + ((o * NT + k) * NT + m) * 64), 4, 0, 0);
                            }
                    }
                }
                lds_barrier();                                           // B2: the four partial products are there
                const long long ts2 = clock64();
                float bo[NT], Dt[NT], yv[NT];
                bool isexc = false;                                      // the lane's marker was taken by its literal evaluation (results already in LDS)
                int nexc = 0;
                if (wave == s) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        Dt[t] = (part[t * 64 + lane] + part[(NT + t) * 64 + lane]) + (part[(2 * NT + t) * 64 + lane] + part[(3 * NT + t) * 64 + lane]);
                        yv[t] = ybuf[t * 64 + lane];
                        bo[t] = a[t] - Dt[t];
                        rhs_lds[t * B + c] = a[t] - bo[t];               // D of this marker, read by everybody after the barrier
                    }
                }
                lds_barrier();                                           // B3: D is there
                const long long ts3 = clock64();
                if (wave == s) {
                    // the literal evaluation at the right-hand side these effects imply:  rhs + R Lc (y - D~); exceptions in marker
                    // order (the comment above kSolveMaxOdd)
                    double thr[NT], z[NT];
                    MtPre<NT> Q;
                    draws_of(thr, z, Q);
                    const float* Ts = A.tsec + (int64_t)s * tsec_floats<NT>();
                    unsigned long long fixed = 0ull;
                    bool failed = false;
                    const long long tx0 = clock64();
                    float ao[NT], bv[NT], dv2[NT];
                    long long tx1 = 0;
                    // (ONE copy of the evaluation, inside the loop: a straight-line first pass plus a loop for the exceptions was
                    // measured SLOWER -- 2.8 k instead of 2.3 k cycles for the first pass: the sampler workgroup runs cold code,
                    // its time is instruction fetch, and code size is what counts)
#pragma unroll 1
                    for (;;) {
                        float v[NT], qv[NT], wev[NT], Dl2[NT];
#pragma unroll
                        for (int t = 0; t < NT; ++t) v[t] = yv[t] - Dt[t];
#pragma unroll
                        for (int k = 0; k < NT; ++k) {
                            float acc = Q.C11[k] * v[k];
#pragma unroll
                            for (int j = 0; j < k; ++j) acc = fmaf(K.Ginv[k][j] + (dj * 1.f) * K.Rinv[k][j], v[j], acc);
                            qv[k] = acc;
                        }
#pragma unroll
                        for (int m = 0; m < NT; ++m) {
                            float acc = 0.f;
#pragma unroll
                            for (int k = 0; k < NT; ++k) acc = fmaf(K.Rm[m][k], qv[k], acc);
                            wev[m] = (rhs[m] + acc) + da[m];
                        }
#pragma unroll
                        for (int t = 0; t < NT; ++t) { ao[t] = a[t]; bv[t] = bb[t]; dv2[t] = dd[t]; Dl2[t] = 0.f; }
                        mt1_eval<NT, false>(K, Q, PriorMem{lpr, 1}, wev, dj, thr, z, ao, bv, dv2, Dl2);
                        bool ok = in_all;
#pragma unroll
                        for (int t = 0; t < NT; ++t) ok = ok && (dv2[t] == 1.f);
                        const unsigned long long bad = __ballot(!ok) & ~fixed;
                        if (tx1 == 0) tx1 = clock64();
                        if (bad == 0ull) break;
                        if (++nexc > kSolveMaxExc) { failed = true; break; }
                        const int e = __builtin_amdgcn_readfirstlane(__builtin_ctzll(bad));      // the first exception: final up to here
                        float del[NT];
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const float ve = a[t] - ao[t];
                            del[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ve - Dt[t]), e));
                            if (lane == e) {                     // its results go to LDS at once (a section that falls back is walked: rewritten)
                                acur[t * B + c] = ao[t]; bcur[t * B + c] = bv[t]; dcur[t * B + c] = dv2[t];
                                rhs_lds[t * B + c] = ve; Dt[t] = ve;
                            }
                        }
                        if (lane == e) isexc = true;
                        float tv[NT][NT];
                        if ((odd0 >> e) & 1ull) {                                   // (wave-uniform) prefetched above
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            const int o = __popcll(odd0 & ((1ull << e) - 1ull));
#pragma unroll
                            for (int k = 0; k < NT; ++k)
#pragma unroll
                                for (int m = 0; m < NT; ++m) tv[k][m] = tcol[((o * NT + k) * NT + m) * 64 + lane];
                        } else {
#pragma unroll
                            for (int k = 0; k < NT; ++k)
#pragma unroll
                                for (int m = 0; m < NT; ++m) {
                                    const int cidx = m * 64 + e;
                                    tv[k][m] = Ts[((int64_t)(cidx >> 2) * (64 * NT) + (k * 64 + lane)) * 4 + (cidx & 3)];
                                }
                        }
                        if (lane > e) {
#pragma unroll
                            for (int k = 0; k < NT; ++k)
#pragma unroll
                                for (int m = 0; m < NT; ++m) Dt[k] = fmaf(tv[k][m], del[m], Dt[k]);
                        }
                        fixed |= 1ull << e;
                    }
                    if (lane == 0) { cy_x1 += tx1 - tx0; cy_xn += clock64() - tx1; }
                    kill_pq();
                    if (failed) { if (lane == 0) sflag[4 + s] = 1; }
                    else if (nexc > 0) {
                        // the optimistic apply of the others used the first D~: they restore and apply these
                        if (!isexc) {
#pragma unroll
                            for (int t = 0; t < NT; ++t) { bo[t] = a[t] - Dt[t]; rhs_lds[t * B + c] = a[t] - bo[t]; }
                        }
                        if (lane == 0) sflag[4 + s] = 2;
                    }
                } else if (wave > s) {
                    apply_changes(s, rhs, pq);                           // (optimistic: undone below if the verdict is "no")
                }
                lds_barrier();                                           // B4: the verdict
                ts4 = clock64();
                cy_y += ts1 - ts0; cy_mv += ts2 - ts1; cy_cmb += ts3 - ts2; cy_ver += ts4 - ts3;
                const int verdict = sflag[4 + s];                        // 0: solved; 2: solved with exceptions; 1: to the walk
                redo = verdict == 1;
                if (redo) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) rhs[t] = rsv[t];
                    if (wave == s) ++nfailed;
                } else if (wave == s) {
                    ++nsolved; nexc_all += nexc;
                    if (!isexc) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) { acur[t * B + c] = bo[t]; bcur[t * B + c] = bo[t]; dcur[t * B + c] = 1.f; }
                    }
                } else if (verdict == 2 && wave > s) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) rhs[t] = rsv[t];
                    apply_changes(s, rhs, pq);
                }
            }
            if (redo) {
                if (wave == s) {
                    // ---- walk section s (lane = marker 64 s + lane) on its strictly-upper diagonal tile, as dense_big_mt does; the
                    // tile is fetched HERE (direct loads: 4 rows of 64 floats per instruction), by the wave that walks it -- the
                    // launch's front does not pay for tiles that nine sections in ten never read
                    {
                        typedef __attribute__((address_space(3))) void lds_void;
                        float* tw = reinterpret_cast<float*>(smem + SM.rows_off) + s * 4096;
                        const float* src = A.gram + (int64_t)(64 * s + (lane >> 4)) * kB + 64 * s + (lane & 15) * 4;
#pragma unroll 1
                        for (int r4 = 0; r4 < 16; ++r4)
                            __builtin_amdgcn_global_load_lds(src + (int64_t)(4 * r4) * kB, (lds_void*)(tw + r4 * 256), 16, 0, 0);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        mask_diagonal_tile(tw, 64, lane, 64);
                    }
                    const float* tile = tiles + s * 4096;
                    double thr[NT], z[NT];
                    MtPre<NT> Q;
                    draws_of(thr, z, Q);
                    float rs[NT], wev[NT], an[NT], bnw[NT], dn[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) rs[t] = rhs[t];
                    auto eval_own = [&](const float (&w)[NT], float (&ao)[NT], float (&bo)[NT], float (&d_o)[NT], float (&Dl)[NT]) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) { ao[t] = a[t]; bo[t] = bb[t]; d_o[t] = dd[t]; Dl[t] = 0.f; }
                        mt1_eval<NT>(K, Q, PriorMem{lpr, 1}, w, dj, thr, z, ao, bo, d_o, Dl, Al, cl);
                    };
                    const Mt1Hoist<NT> Hh = mt1_hoist<NT>(K, Q, PriorMem{lpr, 1}, dj, bb, dd, z);      // (the rhs-independent part, once per marker)
                    auto walk_mixed = [&](unsigned long long slow) {
                        float g = tile[lane];
#pragma unroll 1
                        for (int l = 0; l < 64; ++l) {
                            float w[NT], bo[NT], d_o[NT], Dl[NT];
#pragma unroll
                            for (int t = 0; t < NT; ++t) w[t] = rhs[t] + da[t];
                            if ((slow >> l) & 1ull) mt1_eval_hoisted<NT>(K, Q, Hh, PriorMem{lpr, 1}, w, a, dd, thr, Al, cl, bo, d_o, Dl);      // (wave-uniform)
                            else {
                                mt1_linear_beta<NT>(Al, cl, w, bo);
#pragma unroll
                                for (int t = 0; t < NT; ++t) Dl[t] = a[t] - bo[t];
                            }
                            const float gl = g;
                            g = tile[(l + 1 < 64 ? l + 1 : 63) * 64 + lane];
#pragma unroll
                            for (int t = 0; t < NT; ++t) {
                                const float D = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Dl[t]), l));
                                rhs[t] = fmaf(D, gl, rhs[t]);                                             // (strictly upper tile: lanes <= l are not moved)
                            }
                        }
                    };
                    // (one loop serves the all-fast case too: this path is the exception here, code size matters more than its speed)
                    unsigned long long slow = __ballot(!in_all);
                    if (__popcll(slow) * 4 > 64) slow = ~0ull;
                    walk_mixed(slow);
                    float Dl[NT];
                    for (int pass = 0; pass < 64; ++pass) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) wev[t] = rhs[t] + da[t];                             // what the lane's marker was evaluated with
                        eval_own(wev, an, bnw, dn, Dl);
                        bool ok = true;
#pragma unroll
                        for (int t = 0; t < NT; ++t) ok = ok && (dn[t] == 1.f);
                        const unsigned long long bad = __ballot(!ok) & ~slow;
                        if (bad == 0ull) break;
                        slow |= bad;
                        if (__popcll(slow) * 4 > 64) slow = ~0ull;
#pragma unroll
                        for (int t = 0; t < NT; ++t) rhs[t] = rs[t];
                        walk_mixed(slow);
                        ++nredo;
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        acur[t * B + c] = an[t]; bcur[t * B + c] = bnw[t]; dcur[t * B + c] = dn[t];
                        rhs_lds[t * B + c] = a[t] - an[t];
                    }
                    kill_pq();
                }
                lds_barrier();                                           // B5
                if (wave > s) apply_changes(s, rhs, pq);
            }
            // the section's changes are final: hand them to the helper workgroup that forms the NEXT block's lookahead correction
            // (corr_helper, sampler_common.hpp) -- value and tag in one 8-byte write-through store per trait, fire and forget: the reader polls the tag,
            // so no acknowledgement (s_waitcnt vmcnt) and no flag are needed, and nobody in THIS workgroup waits for anything.
            if (wave == s) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    __hip_atomic_store(A.xch + t * kB + c,
                                       ((unsigned long long)(unsigned)(A.xch_epoch + s + 1) << 32) | (unsigned long long)__float_as_uint(rhs_lds[t * B + c]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (s + 1 < kSec && wave > s + 1) load_g(s + 1);
            if (!redo) cy_tail += clock64() - ts4;
        }
        if (tid == 0) {
            atomicAdd(&A.counters[18], (unsigned long long)cy_y); atomicAdd(&A.counters[19], (unsigned long long)cy_mv);
            atomicAdd(&A.counters[20], (unsigned long long)cy_cmb); atomicAdd(&A.counters[21], (unsigned long long)cy_ver);
            atomicAdd(&A.counters[22], (unsigned long long)cy_tail);
        }
        if (lane == 0 && nredo) atomicAdd(&A.counters[7], (unsigned long long)nredo);      // sections walked again
        if (lane == 0 && (nsolved | nfailed)) {                                             // Rule T: sections solved / fallen back to the walk
            atomicAdd(&A.counters[16], (unsigned long long)nsolved);
            atomicAdd(&A.counters[17], (unsigned long long)nfailed);
            if (nexc_all) atomicAdd(&A.counters[23], (unsigned long long)nexc_all);         // ... exceptions taken inside the solved ones
            atomicAdd(&A.counters[13], (unsigned long long)cy_x1);                         // (the first verification pass | the exception passes, cycles)
            atomicAdd(&A.counters[14], (unsigned long long)cy_xn);
        }
    } else {
        // ====================== waves 4..7: their quarters of the mat-vec, nothing else (the same barriers as above) ======================
        v4f tq[NT][QG];
        load_t(0, tq);
        lds_barrier();                                                                                  // B0
#pragma unroll 1
        for (int s = 0; s < kSec; ++s) {
            const bool fast = sflag[s] != 0;
            bool redo = !fast;
            if (fast) {
                lds_barrier();                                           // B1: y is there
                matvec(tq);
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < kSec) load_t(s + 1, tq, 0, QG / 4);          // the next section's T: a quarter per barrier interval
                lds_barrier();                                           // B2: the partial products are there
                if (s + 1 < kSec) load_t(s + 1, tq, QG / 4, QG / 2);
                lds_barrier();                                           // B3: D is there
                if (s + 1 < kSec) load_t(s + 1, tq, QG / 2, 3 * QG / 4);
                lds_barrier();                                           // B4: the verdict
                if (s + 1 < kSec) load_t(s + 1, tq, 3 * QG / 4, QG);
                redo = sflag[4 + s] == 1;
            }
            if (redo) {
                if (!fast && s + 1 < kSec) load_t(s + 1, tq);
                lds_barrier();                                           // B5
            }
        }
    }
    __syncthreads();
    const long long tk4 = clock64();
    // the block's change list in marker order
    const bool rowthr = tid < kB;
    const int c = rowthr ? tid : 0;
    bool changed = false;
    if (rowthr) {
#pragma unroll
        for (int t = 0; t < NT; ++t) changed = changed || (astart[t * B + c] != acur[t * B + c]);
    }
    const unsigned long long cm = __ballot(changed);
    if (lane == 0) wcnt[wave] = __popcll(cm);
    __syncthreads();
    int base = 0, nfin = 0;
#pragma unroll
    for (int q = 0; q < kStepThreads / 64; ++q) { const int v = wcnt[q]; base += (q < wave) ? v : 0; nfin += v; }
    // ---- global stores last
    if (changed) {
        const int e = base + __popcll(cm & ((1ull << lane) - 1ull));
        A.ev_out->idx[e] = (int32_t)(j0 + c);
#pragma unroll
        for (int t = 0; t < NT; ++t) A.ev_out->delta[t][e] = astart[t * B + c] - acur[t * B + c];
    }
    if (rowthr) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float a_fin = acur[t * B + c];
            if (a_fin != astart[t * B + c]) A.alpha[(int64_t)t * p + j0 + c] = a_fin;
            A.beta[(int64_t)t * p + j0 + c] = bcur[t * B + c];
            delta[(int64_t)t * p + j0 + c] = dcur[t * B + c];
        }
    }
    if (tid == 0) {
        A.ev_out->count = (int32_t)nfin;
        atomicAdd(&A.counters[0], (unsigned long long)nfin);
        atomicAdd(&A.counters[2], (unsigned long long)(tk1 - tk0));      // front
        atomicAdd(&A.counters[5], (unsigned long long)(tk4 - tk1));      // sections (solves / walks + off-diagonal applies)
    }
}

template <int METHOD, int NT, bool DW = false>
__device__ __forceinline__ void sampler_role_mt(char* smem, const SamplerArgs& A)
{
    constexpr bool kDW = DW && !is_sampler2(METHOD);
    const bool pm = A.lpr_mat != nullptr;           // marker-specific joint priors (host: only with parked draws)
    constexpr bool kPG = has_marker_cov(METHOD);   // a t x t effect covariance per marker (host: only with parked draws)
    const StepSmem SM(A.bsz, NT, mt_park_nd(A.bsz, NT) + (pm ? (1 << NT) : 0), mt_park_nf(A.bsz, NT) + (kPG ? NT * NT : 0));
    const int B = SM.B;
    const bool parked = mt_park_nd(B, NT) != 0;
    constexpr bool kTab = (METHOD == kMTBayesC2) && (NT <= 3);       // sampler II from per-marker state tables
    constexpr int kTS = kTab ? (1 << NT) : 1, kTV = kTab ? NT * (NT + 1) / 2 + 1 : 1;
    const DevParams* P = A.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = A.b;
    const int64_t j0 = A.j0, p = A.p;
    float* rhs_lds = reinterpret_cast<float*>(smem + SM.rhs_off);
    float* acur = reinterpret_cast<float*>(smem + SM.acur_off);
    float* astart = reinterpret_cast<float*>(smem + SM.astart_off);
    float* bcur = reinterpret_cast<float*>(smem + SM.bcur_off);
    float* dcur = reinterpret_cast<float*>(smem + SM.dcur_off);
    double* lpd = reinterpret_cast<double*>(smem + SM.prepd_off);     // [2 NT][B] thresholds, normals (if parked)
    float* lpf = reinterpret_cast<float*>(smem + SM.prepf_off);       // [B] x'x (if parked)
    float* lpg = lpf + (1 + NT) * B;                                  // [NT*NT][B] the marker's own G^-1 (kPG)
    float* delta = reinterpret_cast<float*>(A.delta);
    const long long tk0 = clock64();

    // The sweep's constants come through P: loads (one memory latency -- microseconds under the update role's stream).  Here they are only
    // ISSUED; everything computed from them (fill_K, below) waits until the front has issued the loads of the markers' state and partial
    // sums, so that the role starts with ONE memory latency.  (Computing K first, and reading P->nreps for the branches in front of the
    // marker loads, was three dependent latencies before the first marker load: ~5 k of the front's 19 k cycles per block.)
    MtConsts<NT> K;
    float rawR[NT * NT], rawG[NT * NT], rawV[NT * NT], rawVe[NT];
    double rawPi[NT];
#pragma unroll
    for (int i = 0; i < NT * NT; ++i) { rawR[i] = P->Rinv[i]; rawG[i] = P->Ginv[i]; rawV[i] = P->vare[i]; }
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        rawVe[a] = 0.f; rawPi[a] = 0.0;
        if constexpr (is_mega(METHOD)) { rawVe[a] = P->var_effect[a * NT + a]; rawPi[a] = P->pi4[a]; }
    }
    auto fill_K = [&]() {
#pragma unroll
        for (int a = 0; a < NT; ++a) {
#pragma unroll
            for (int c = 0; c < NT; ++c) { K.Rinv[a][c] = rawR[a * NT + c]; K.Ginv[a][c] = rawG[a * NT + c]; K.Rm[a][c] = rawV[a * NT + c]; }
            K.invG[a] = 1.0f / K.Ginv[a][a];                            // MTBayesABC.jl:92
            K.lG[a] = logf_via_double(K.Ginv[a][a]);
            K.sG[a] = sqrtf(K.invG[a]);
            if constexpr (is_mega(METHOD)) {
                K.ie[a]  = 1.0f / rawV[a * NT + a];                     // invVarRes          BayesABC.jl:69
                K.var[a] = rawVe[a];
                K.iv[a]  = 1.0f / K.var[a];                             // invVarEffects[j]   :70
                K.lv[a]  = logf_via_double(K.var[a]);                   // logVarEffects[j]   :71
                K.sv[a]  = sqrtf(K.var[a]);
                K.lp0[a] = log(rawPi[a]);                               // logPi              :67
                K.lp1[a] = log(1.0 - rawPi[a]);                         // logPiComp          :68
            }
        }
    };
    // multi-trait BayesA/B: the constants that depend on G are the marker's own (its inverse was formed by k_prepare)
    auto with_ginv = [&](const float (&g)[NT * NT]) {
        MtConsts<NT> Kj = K;
        if constexpr (is_mega(METHOD)) {                                // (the parked matrix is the marker's variances)
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                Kj.var[a] = g[a * NT + a];
                Kj.iv[a]  = 1.0f / Kj.var[a];                           // invVarEffects[j]   BayesABC.jl:70
                Kj.sv[a]  = sqrtf(Kj.var[a]);
            }
            return Kj;
        }
#pragma unroll
        for (int a = 0; a < NT; ++a) {
#pragma unroll
            for (int c2 = 0; c2 < NT; ++c2) Kj.Ginv[a][c2] = g[a * NT + c2];
            Kj.invG[a] = 1.0f / Kj.Ginv[a][a];                      // MTBayesABC.jl:92
            Kj.lG[a] = logf_via_double(Kj.Ginv[a][a]);
            Kj.sG[a] = sqrtf(Kj.invG[a]);
        }
        return Kj;
    };
    auto consts_of = [&](int c) {                                     // marker c of the block (after the front's barrier)
        if constexpr (kPG) {
            float g[NT * NT];
#pragma unroll
            for (int i = 0; i < NT * NT; ++i) g[i] = lpg[i * B + c];
            return with_ginv(g);
        } else { (void)c; return K; }
    };
    // the 2^NT log prior state probabilities are indexed by the running state inside every evaluation: a global load
    // there would put a memory latency (microseconds under full-rate streaming) on each trait of each round -- LDS copy
    double* lpr = reinterpret_cast<double*>(smem + SM.lpr_off);
    const double lpr_mine = P->log_prior[tid < (1 << NT) ? tid : 0];

    // ---- front (all threads, ONE memory latency): every thread issues the loads of its marker's state, draws, x'x,
    // lookahead correction and row-group partials back to back, forms  rhs = fl32(sum of partials) + corr,  parks
    // everything the serial wave needs in LDS, and decides candidacy: a marker already in the model for some trait
    // (its effects always change) or one whose evaluation against the entry rhs changes an effect.  Candidates get
    // their Gram row staged in LDS; a change of a non-candidate reads its row from HBM inside the serial phase.
    // small blocks (the host's choice for dense priors): the whole Gram block with the very first loads, as in the
    // single-trait sampler (slot of marker c = c)
    const bool prestage = (B <= 128) && (B <= SM.max_cand);
    const bool gram_dma = prestage && b == B;              // full block: direct global -> LDS loads (see sampler_role_st)
    const bool cross_dma = gram_dma && SM.has_cross && A.b_next == B;
    if (gram_dma) dma_copy_to_lds(A.gram, reinterpret_cast<float*>(smem + SM.rows_off), B * B);
    // full 256-marker blocks of sampler I with one shared covariance, single pass: dense_big_mt (diagonal Gram tiles in LDS, the
    // rest in parallel) when (nearly) every marker changes.  The dense-walk-only instantiation knows that before any load and
    // fetches the tiles with the launch's first loads (the host selects it for full blocks only); the general instantiation
    // decides after the candidates have been counted.
    bool big_try = false;
    auto fetch_tiles = [&]() {
        if (wave < 4) {
            typedef __attribute__((address_space(3))) void lds_void;
            float* tile = reinterpret_cast<float*>(smem + SM.rows_off) + wave * 4096;
            const float* src = A.gram + (int64_t)(64 * wave + (lane >> 4)) * b + 64 * wave + (lane & 15) * 4;
#pragma unroll 1
            for (int r4 = 0; r4 < 16; ++r4)       // 4 rows of 64 floats per instruction: lane -> row lane / 16, float4 column lane % 16
                __builtin_amdgcn_global_load_lds(src + (int64_t)(4 * r4) * b, (lds_void*)(tile + r4 * 256), 16, 0, 0);
        }
    };
    auto finish_tiles = [&]() {                            // landed, strictly upper (each wave its own tile), visible to everybody
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wave < 4) mask_diagonal_tile(reinterpret_cast<float*>(smem + SM.rows_off) + wave * 4096, 64, lane, 64);
        __syncthreads();
    };
    if constexpr (is_sampler1(METHOD)) {
        big_try = (B == 256) && (b == B) && !pm && parked && (A.nreps == 1) && !A.dense_big_off;      // (any next block: its columns are threads 256 .. 256 + b_next - 1)
        if constexpr (kDW) { if (big_try && A.tsec == nullptr) fetch_tiles(); }      // (Rule T: a tile is fetched only by a section that has to be walked)
    }
    float4 gpre[8];
    if (prestage && !gram_dma) {
        const int per_row = B >> 2, total = b * per_row;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + u * kStepThreads;
            const int ec = e < total ? e : 0;
            const int row = ec / per_row, c4 = (ec - row * per_row) * 4;
            const float* src = A.gram + (int64_t)row * b;
            if (b == B) gpre[u] = *reinterpret_cast<const float4*>(src + c4);
            else {
                gpre[u].x = src[c4 < b ? c4 : 0]; gpre[u].y = src[c4 + 1 < b ? c4 + 1 : 0];
                gpre[u].z = src[c4 + 2 < b ? c4 + 2 : 0]; gpre[u].w = src[c4 + 3 < b ? c4 + 3 : 0];
            }
        }
    }
    bool cand[2] = {false, false};
    float djq_[2], a0[2][NT], b0[2][NT], d0[2][NT], w0[2][NT], lc0[2][NT], gq_[2][kPG ? NT * NT : 1];
    double thr0[2][NT], z0[2][NT];
    // Rule T launches (full 256-marker blocks): the front is bound by the instruction issue of its four marker waves (~50 loads
    // with their addresses, ~35 LDS stores per thread) -- waves 4..7, idle otherwise, take the right-hand sides (the row-group
    // partial sums + the lookahead correction: 3 nrg + 3 loads per marker) off them
    bool split_front = false;
    if constexpr (kDW && is_sampler1(METHOD) && NT <= 3) split_front = A.tsec != nullptr;
    if (split_front && tid >= 256) {
        const int c = tid - 256;
        float co[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) co[t] = A.corr_in[t * B + c];
        double psum[NT];
        sum_partials_traits<NT>(A.partials + c, (int64_t)A.nrg * A.bstride, A.nrg, A.bstride, psum);
#pragma unroll
        for (int t = 0; t < NT; ++t) rhs_lds[t * B + c] = (float)psum[t] + co[t];
    }
#pragma unroll
    for (int q = 0; q < (kDW ? 1 : 2); ++q) {
        const int c = tid + q * kStepThreads;
        if (c >= B) continue;
        const int cc = c < b ? c : 0;
        const int64_t j = j0 + cc;
        const float dj = A.xpx[j];
        float co[NT];
        djq_[q] = dj;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            a0[q][t] = A.alpha[(int64_t)t * p + j]; b0[q][t] = A.beta[(int64_t)t * p + j]; d0[q][t] = delta[(int64_t)t * p + j];
            co[t] = A.corr_in[t * B + c];
            thr0[q][t] = A.prep_d[(int64_t)t * p + j]; z0[q][t] = A.prep_d[(int64_t)(NT + t) * p + j];
            lc0[q][t] = A.prep_f[(int64_t)t * p + j];
        }
        if constexpr (kPG) {
#pragma unroll
            for (int i = 0; i < NT * NT; ++i) gq_[q][i] = A.ginv_mat[j * (NT * NT) + i];
        }
        double lpm[1 << NT];
        if (pm) {
#pragma unroll
            for (int st = 0; st < (1 << NT); ++st) lpm[st] = A.lpr_mat[(int64_t)(1 << NT) * j + st];
        }
        double psum[NT];
        if (!split_front) sum_partials_traits<NT>(A.partials + cc, (int64_t)A.nrg * A.bstride, A.nrg, A.bstride, psum);
        else {
#pragma unroll
            for (int t = 0; t < NT; ++t) psum[t] = 0.0;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const double sum = psum[t];
            const float rhs0 = (float)sum + co[t];
            const float a_in = (c < b) ? a0[q][t] : 0.f;
            if (!split_front) rhs_lds[t * B + c] = rhs0;
            acur[t * B + c] = a_in; astart[t * B + c] = a_in;
            bcur[t * B + c] = b0[q][t]; dcur[t * B + c] = d0[q][t];
            w0[q][t] = rhs0 + dj * a_in;                                                             // :82
            if (parked) { lpd[t * B + c] = thr0[q][t]; lpd[(NT + t) * B + c] = z0[q][t]; lpf[(1 + t) * B + c] = lc0[q][t]; }
            a0[q][t] = a_in;
        }
        if (parked) lpf[c] = dj;
        if constexpr (kPG) {
#pragma unroll
            for (int i = 0; i < NT * NT; ++i) lpg[i * B + c] = gq_[q][i];
        }
        if (pm) {
#pragma unroll
            for (int st = 0; st < (1 << NT); ++st) lpd[(2 * NT + st) * B + c] = lpm[st];
        }
    }
    fill_K();
    if (tid < (1 << NT)) lpr[tid] = lpr_mine;
    if (tid == 0) { int* wz = reinterpret_cast<int*>(smem + SM.wcnt_off); wz[14] = 0; wz[8] = 0; wz[9] = 0; wz[10] = 16; }      // [14]: set by the dense walk; [8..10]: skip and verify
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
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int c = tid; c < B; c += kStepThreads) { slot_p[c] = (short)(c < b ? c : -1); cand_p[c] = (short)c; }
    }
    const long long tkf1 = clock64();
    __syncthreads();
    const long long tkf2 = clock64();
    if (tid == 0) {       // (diagnostics: the front up to its LDS stores | the wait for its loads at the barrier)
        atomicAdd(&A.counters[10], (unsigned long long)(tkf1 - tk0)); atomicAdd(&A.counters[11], (unsigned long long)(tkf2 - tkf1));
    }
    // marker c's table of log prior state probabilities: the shared one (stride 1) or its own column of the parked
    // marker-specific priors (stride B)
    const int ls = pm ? B : 1;
    auto lpr_of = [&](int c) -> const double* { return pm ? lpd + 2 * NT * B + c : lpr; };
    bool stay[2] = {false, false};
    float pb[2][NT], pd[2][NT];
#pragma unroll
    for (int q = 0; q < (kDW ? 1 : 2); ++q) {
        const int c = tid + q * kStepThreads;
        if (c >= b) continue;
        bool in_model = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) in_model = in_model || (a0[q][t] != 0.f);
        bool moves = false;
        if constexpr (kDW) { in_model = true; }
        else if (!in_model) {
            const float dj = djq_[q];
            MtConsts<NT> Kc = K;
            if constexpr (kPG) Kc = with_ginv(gq_[q]);
            const MtPre<NT> Q0 = mt_precompute<METHOD, NT>(Kc, dj, lc0[q]);
            float an[NT], bn[NT], dn[NT], Dl[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) { an[t] = a0[q][t]; bn[t] = b0[q][t]; dn[t] = d0[q][t]; Dl[t] = 0.f; }
            if constexpr (is_sampler1(METHOD)) mt1_eval<NT, false>(Kc, Q0, PriorMem{lpr_of(c), ls}, w0[q], dj, thr0[q], z0[q], an, bn, dn, Dl);
            else if constexpr (kTab) {
                double T[kTS][kTV];
                mt2_load_tab<NT>(A.mt2_tab, p, j0 + c, T);
                mt2_eval_tab<NT>(K, lpr_of(c), ls, w0[q], T, thr0[q][0], z0[q], an, bn, dn, Dl);
            }
            else if constexpr (is_sampler2(METHOD)) mt2_eval<NT>(Kc, lpr_of(c), ls, w0[q], dj, thr0[q][0], z0[q], an, bn, dn, Dl);
            else mega_eval<NT>(Kc, Q0, w0[q], dj, thr0[q], z0[q], an, bn, dn, Dl);
#pragma unroll
            for (int t = 0; t < NT; ++t) moves = moves || (Dl[t] != 0.f);
            if (!moves) {
                stay[q] = true;
#pragma unroll
                for (int t = 0; t < NT; ++t) { pb[q][t] = bn[t]; pd[q][t] = dn[t]; }
            }
        }
        cand[q] = in_model || moves;
    }
    // PREFIX SKIP (as in the single-trait sampler): until the first candidate of the block commits the running rhs is the
    // entry rhs, so the evaluation above is final for every marker before it.  Their freshly drawn beta / delta are parked
    // (only theirs: a later marker is re-evaluated from its OLD state) and the serial wave starts at the first sub-block
    // that holds a candidate.  Single pass only.
    int first_sub = 16, ncand_all = 0;
    bool dense_walk = false;
    // SKIP AND VERIFY (single pass, 256- / 512- / 1024-marker blocks, the general path).  A 64-marker sub-block behind the first candidate that
    // holds no candidate itself -- every marker outside the model and staying there against the ENTRY right-hand side -- is not
    // walked by the serial wave: a helper wave (wave w: sub-blocks w, w + 7, w + 14) evaluates it, once, as soon as every candidate sub-block in front of it is done
    // (the right-hand side of its markers is then what it is at their own steps, unless a skipped sub-block in between moves), and
    // reports "nobody moves" or "somebody does".  The serial wave goes on with the next candidate sub-block when every skipped
    // sub-block in front of it has reported; after the first "somebody moves" it takes the chain over from that sub-block on, the
    // ordinary way (the helpers' results behind it are dropped: they only reach LDS after the walk).  Same chain, same bits: a
    // sparse steady state holds about one candidate per 512-marker block, and the serial wave used to evaluate every sub-block
    // behind it, one after the other.
    unsigned cand_mask = 0xffffu, skip_set = 0u;       // sub-blocks that hold a candidate | sub-blocks left to the helper waves
    if constexpr (kDW) {
        first_sub = 0; ncand_all = b; dense_walk = true;
        if constexpr (is_sampler1(METHOD)) {
            if (big_try) {
                // (Rule T launches: the host sets tsec exactly for the blocks that meet big_try, and the helper workgroup then counts on
                // the sampler to publish every section's changes)
                if constexpr (NT <= 3) {
                    if (A.tsec != nullptr) { dense_big_mt_solve<METHOD, NT>(smem, SM, A, consts_of, lpr, tk0, clock64()); return; }
                }
                finish_tiles();
                dense_big_mt<METHOD, NT>(smem, SM, A, consts_of(tid < 256 ? tid : 0), lpr, tk0, clock64());
                return;
            }
        }
        float* rows_m = reinterpret_cast<float*>(smem + SM.rows_off);
        for (int e = tid; e < (B >> 6) * 64 * 16; e += kStepThreads) {     // strictly upper diagonal tiles (see below)
            const int q = e >> 10, l = (e >> 4) & 63, c4 = (e & 15) * 4;
            float* dst = rows_m + (64 * q + l) * B + 64 * q + c4;
            if (c4 + 3 <= l) *reinterpret_cast<float4*>(dst) = float4{0.f, 0.f, 0.f, 0.f};
            else if (c4 <= l) { dst[0] = 0.f; if (c4 + 1 <= l) dst[1] = 0.f; if (c4 + 2 <= l) dst[2] = 0.f; }
        }
        __syncthreads();
    } else {
        int* wc = reinterpret_cast<int*>(smem + SM.wcnt_off);
        const int f0 = __any(cand[0]) ? 1 : 0, f1 = __any(cand[1]) ? 2 : 0;
        const int npop = __popcll(__ballot(cand[0])) + __popcll(__ballot(cand[1]));
        if (lane == 0) wc[wave] = f0 | f1 | (npop << 8);
        __syncthreads();
        unsigned mask = 0u;
#pragma unroll
        for (int q = 0; q < kStepThreads / 64; ++q) { const int v = wc[q]; ncand_all += v >> 8; mask |= (unsigned)(v & 1) << q | (unsigned)((v >> 1) & 1) << (8 + q); }
        if (mask) first_sub = __builtin_ctz(mask);
        const bool single_pass = (A.nreps > 0 ? A.nreps : b) == 1;
        if (!single_pass) first_sub = 0;
        cand_mask = mask;
        if (single_pass && !prestage && (B == 256 || B == 512 || B == 1024) && first_sub < 16 && !(A.compact_off & 8))
            skip_set = ~mask & ((1u << ((b + 63) >> 6)) - 1u) & ~((2u << first_sub) - 1u);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = tid + q * kStepThreads;
            if (stay[q] && c < b && (c >> 6) < first_sub) {
#pragma unroll
                for (int t = 0; t < NT; ++t) { bcur[t * B + c] = pb[q][t]; dcur[t * B + c] = pd[q][t]; }
            }
        }
        // The dense walk (below) runs on STRICTLY UPPER diagonal tiles: G[l][c] = 0 for c <= l inside a 64-marker section, so
        // that a lane's running rhs stops moving at its own step -- after the section it still holds the value the lane's
        // marker was evaluated with (no per-step copy of it), and the later steps' updates are exact no-ops on it.  The rows
        // are not read again after the walk (single pass).
        if constexpr (is_sampler1(METHOD)) {
            // (first_sub == 0: nothing was parked by the prefix skip -- the walk starts from every marker's OLD state)
            if (big_try && single_pass && first_sub == 0 && 5 * ncand_all >= 3 * b) {
                fetch_tiles();
                finish_tiles();
                dense_big_mt<METHOD, NT>(smem, SM, A, consts_of(tid < 256 ? tid : 0), lpr, tk0, clock64());
                return;
            }
        }
        dense_walk = !is_sampler2(METHOD) && single_pass && prestage && 5 * ncand_all >= 3 * b;
        if (dense_walk) {
            // one float4 column group per thread and pass: whole groups left of the diagonal with one store
            float* rows_m = reinterpret_cast<float*>(smem + SM.rows_off);
            for (int e = tid; e < (B >> 6) * 64 * 16; e += kStepThreads) {
                const int q = e >> 10, l = (e >> 4) & 63, c4 = (e & 15) * 4;
                float* dst = rows_m + (64 * q + l) * B + 64 * q + c4;
                if (c4 + 3 <= l) *reinterpret_cast<float4*>(dst) = float4{0.f, 0.f, 0.f, 0.f};
                else if (c4 <= l) { dst[0] = 0.f; if (c4 + 1 <= l) dst[1] = 0.f; if (c4 + 2 <= l) dst[2] = 0.f; }
            }
        }
        __syncthreads();                                   // (stage_rows reuses the slots)
    }
    const long long tk1 = clock64();
    const int nstaged_mt = (kDW || prestage) ? b : (first_sub >= 16 ? 0 : stage_rows(smem, SM, A, cand));
    if (cross_dma) {                                       // waves 1..7: the cross-Gram rows to LDS while wave 0 walks the block
        dma_copy_to_lds(A.cross_next, reinterpret_cast<float*>(smem + SM.cross_off), B * B, 1);
        if (wave != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else prefetch_cross_rows(smem, SM, A, nstaged_mt);
    prefetch_next_gram(A, prestage);
    int* wcnt_s = reinterpret_cast<int*>(smem + SM.wcnt_off);
    long long tk3 = 0, tk4 = 0, tk5 = 0;
    int nrounds = 0;
    if (wave == 0) {
    tk3 = clock64();


    const int nsub = (b + 63) / 64;
    const int nreps = A.nreps > 0 ? A.nreps : b;
    RngKey key{P->seed_lo, P->seed_hi, P->iter, 0u};

    // ---- DENSE blocks (every marker of a <= 128-marker block is in the model for some trait -- the default all-ones
    // multi-trait prior): sequential walk instead of speculative rounds, as in the single-trait sampler.  Every lane
    // evaluates ITS OWN marker against its own running rhs at every step -- no operand is broadcast; the step's marker
    // is lane l, whose per-trait alpha_old - alpha_new are broadcast with NT v_readlane and applied to the running rhs of
    // the whole block (NT x 2 registers per lane) with the marker's Gram row from LDS (read a step ahead).  A lane's
    // result is final at its own step: it keeps the w it was evaluated with and recomputes its update after the walk.
    bool dense_done = false;
    // Sampler I, every marker of the block in the model for every trait at entry (the reference's default prior keeps it
    // that way: the states with a trait missing have probability ~0): the walk SPECULATES that every delta stays 1 and
    // evaluates a marker with Rule L's linear form (mt1_linear_coeffs: A, c of all 64 markers of a section formed in
    // parallel, NT^2 fused multiply-adds per marker on the chain).  After a 64-marker section ONE full mt1_eval per lane
    // (all 64 markers at once, each with the w it was walked with) both verifies the speculation and yields the final
    // state -- by Rule L the linear form's own numbers whenever the speculation held; if any marker left the model for a
    // trait the section is walked again from its saved rhs with those markers evaluated the general way.
    if (dense_walk) {
        const float* rows = reinterpret_cast<const float*>(smem + SM.rows_off);
        float rhsq[NT][2], aq[NT][2], bq[NT][2], dq[NT][2], djq[2], wev[2][NT];
        double thrq[NT][2], zq[NT][2];
        MtPre<NT> Qq[2];
        MtConsts<NT> Kq[kPG ? 2 : 1];                               // (kPG: the two markers' own constants)
        auto KQ = [&](int q) -> const MtConsts<NT>& { if constexpr (kPG) return Kq[q]; else { (void)q; return K; } };
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = (64 * q + lane < b) ? 64 * q + lane : 0;
            djq[q] = lpf[c];                                        // (B <= 128: the draws are always parked in LDS)
            float lcq[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) lcq[t] = lpf[(1 + t) * B + c];
            if constexpr (kPG) Kq[q] = consts_of(c);
            Qq[q] = mt_precompute<METHOD, NT>(KQ(q), djq[q], lcq);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                rhsq[t][q] = rhs_lds[t * B + c]; aq[t][q] = acur[t * B + c]; bq[t][q] = bcur[t * B + c]; dq[t][q] = dcur[t * B + c];
                thrq[t][q] = lpd[t * B + c]; zq[t][q] = lpd[(NT + t) * B + c];
                wev[q][t] = 0.f;
            }
        }
        const bool speculate = is_sampler1(METHOD);
        // one marker evaluated in-lane from (w, its state at block entry, its draws)
        auto eval_own = [&](int q, const float (&w)[NT], float (&an)[NT], float (&bn)[NT], float (&dn)[NT], float (&Dl)[NT],
                            const float (*Apre)[NT] = nullptr, const float* cpre = nullptr) {
            const int c = (64 * q + lane < b) ? 64 * q + lane : 0;
            double thr[NT], z[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) { an[t] = aq[t][q]; bn[t] = bq[t][q]; dn[t] = dq[t][q]; Dl[t] = 0.f; thr[t] = thrq[t][q]; z[t] = zq[t][q]; }
            // (the shared prior table from registers -- v_cndmask trees instead of the LDS lookup -- was measured: 79 ms
            // per sweep instead of 55 at 3 traits x 20k x 100k; the LDS read overlaps the trait's arithmetic well enough)
            if constexpr (is_sampler1(METHOD)) mt1_eval<NT>(KQ(q), Qq[q], PriorMem{lpr_of(c), ls}, w, djq[q], thr, z, an, bn, dn, Dl, Apre, cpre);
            else { (void)Apre; (void)cpre; mega_eval<NT>(KQ(q), Qq[q], w, djq[q], thr, z, an, bn, dn, Dl); }
        };
        // the speculative conditionals = Rule L's linear form (mt1_linear_coeffs): A, c of the lane's own marker are formed once
        // per section; a step is NT^2 fused multiply-adds.  Dl = alpha_old - alpha_new
        auto eval_fast = [&](int q, const float (&w)[NT], const float (&Al)[NT][NT], const float (&cl)[NT], float (&bn)[NT], float (&Dl)[NT]) {
            mt1_linear_beta<NT>(Al, cl, w, bn);
#pragma unroll
            for (int k = 0; k < NT; ++k) Dl[k] = aq[k][q] - bn[k];
        };
        auto linear_of = [&](int q, float (&Al)[NT][NT], float (&cl)[NT]) {
            float b_old[NT];
            double zz[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) { b_old[t] = bq[t][q]; zz[t] = zq[t][q]; }
            mt1_linear_coeffs<NT>(KQ(q), Qq[q], djq[q], b_old, zz, Al, cl);
        };
        // 64-marker section q: eight steps per batch without a branch, the Gram rows read a batch ahead
        // (TWO: blocks of 128 markers -- the second half's running rhs follows too; a compile-time property of the loop body)
        auto section = [&](auto qc, auto fastc, auto twoc, const float* grow, int nsteps, const float (&Al)[NT][NT], const float (&cl)[NT]) {
            constexpr int Q = decltype(qc)::value;
            constexpr bool FAST = decltype(fastc)::value;
            constexpr bool TWO = decltype(twoc)::value;
            float da[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) da[t] = djq[Q] * aq[t][Q];                                 // :82
            auto step = [&](int l, float c0, float c1) {
                float w[NT], an[NT], bn[NT], dn[NT], Dl[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) w[t] = rhsq[t][Q] + da[t];
                if constexpr (FAST) eval_fast(Q, w, Al, cl, bn, Dl);
                else eval_own(Q, w, an, bn, dn, Dl);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    // (the diagonal tiles are strictly upper: lanes <= l are not moved -- a lane keeps the rhs of its own step)
                    const float D = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Dl[t]), l));
                    if (Q == 0) rhsq[t][0] = fmaf(D, c0, rhsq[t][0]);                                   // D = 0: exact no-op
                    if (TWO) rhsq[t][1] = fmaf(D, c1, rhsq[t][1]);
                }
            };
            constexpr int kBatch = FAST ? 8 : 2;
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
                if (l + 2 * kBatch <= nsteps) load(l + kBatch);
#pragma unroll
                for (int u = 0; u < kBatch; ++u) step(l + u, c0[u], c1[u]);
            }
#pragma unroll 1
            for (; l < nsteps; ++l) step(l, (Q == 0) ? grow[l * B + lane] : 0.f, TWO ? grow[l * B + 64 + lane] : 0.f);
#pragma unroll
            for (int t = 0; t < NT; ++t) wev[Q][t] = rhsq[t][Q] + da[t];     // every lane: what its marker was evaluated with
        };
        using std::integral_constant;
        // the same section with some markers (bit l of `slow`) evaluated the general way and the others speculatively: one
        // step per loop trip (used when the speculation missed, or when a marker is not in the model for every trait at entry)
        auto section_mixed = [&](auto qc, auto twoc, const float* grow, int nsteps, unsigned long long slow, const float (&Al)[NT][NT], const float (&cl)[NT]) {
            constexpr int Q = decltype(qc)::value;
            constexpr bool TWO = decltype(twoc)::value;
            float da[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) da[t] = djq[Q] * aq[t][Q];
            float g0 = (Q == 0) ? grow[lane] : 0.f;
            float g1 = TWO ? grow[64 + lane] : 0.f;
#pragma unroll 1
            for (int l = 0; l < nsteps; ++l) {
                float w[NT], an[NT], bn[NT], dn[NT], Dl[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) w[t] = rhsq[t][Q] + da[t];
                if ((slow >> l) & 1ull) eval_own(Q, w, an, bn, dn, Dl, Al, cl);                       // (wave-uniform)
                else eval_fast(Q, w, Al, cl, bn, Dl);
                const float c0 = g0, c1 = g1;
                grow += B;                                           // next marker's row (one past the block: the overflow row)
                if (Q == 0) g0 = grow[lane];
                if (TWO) g1 = grow[64 + lane];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float D = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Dl[t]), l));
                    if (Q == 0) rhsq[t][0] = fmaf(D, c0, rhsq[t][0]);
                    if (TWO) rhsq[t][1] = fmaf(D, c1, rhsq[t][1]);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) wev[Q][t] = rhsq[t][Q] + da[t];
        };
        auto run_section = [&](auto qc, auto twoc) {
            constexpr int Q = decltype(qc)::value;
            const int nsteps = (b < 64 * (Q + 1) ? b : 64 * (Q + 1)) - 64 * Q;
            if (nsteps <= 0) return;
            const float* grow = rows + 64 * Q * B;                   // (all rows staged in marker order: slot = marker)
            const int c = 64 * Q + lane;
            float an[NT], bn[NT], dn[NT], Dl[NT];
            float Alq[NT][NT], clq[NT];                              // Rule L's coefficients of the section's markers: formed ONCE
            if (speculate) {
                linear_of(Q, Alq, clq);
                float rs[NT][2];
#pragma unroll
                for (int t = 0; t < NT; ++t) { rs[t][0] = rhsq[t][0]; rs[t][1] = rhsq[t][1]; }
                // markers that are not in the model for every trait at entry cannot be speculated on
                bool in_all = true;
#pragma unroll
                for (int t = 0; t < NT; ++t) in_all = in_all && (dq[t][Q] == 1.f);
                unsigned long long slow = __ballot(!in_all && c < b);
                if (__popcll(slow) * 4 > nsteps) slow = ~0ull;       // not a block to speculate on: everything the general way
                if (slow == 0ull) section(qc, integral_constant<bool, true>{}, twoc, grow, nsteps, Alq, clq);
                else section_mixed(qc, twoc, grow, nsteps, slow, Alq, clq);
                for (int pass = 0; pass < 64; ++pass) {
                    eval_own(Q, wev[Q], an, bn, dn, Dl, Alq, clq);   // the exact evaluation of every marker of the section
                    bool ok = true;
#pragma unroll
                    for (int t = 0; t < NT; ++t) ok = ok && (dn[t] == 1.f);
                    // a speculated marker that leaves the model for a trait: its broadcast changes were wrong -- evaluate it
                    // (and whatever else looks wrong now) the general way and walk the section again from its saved rhs
                    const unsigned long long bad = __ballot(!ok && c < b) & ~slow;
                    if (bad == 0ull) break;
                    slow |= bad;
                    if (__popcll(slow) * 4 > nsteps) slow = ~0ull;   // (misses are not rare here: stop speculating)
#pragma unroll
                    for (int t = 0; t < NT; ++t) { rhsq[t][0] = rs[t][0]; rhsq[t][1] = rs[t][1]; }
                    section_mixed(qc, twoc, grow, nsteps, slow, Alq, clq);
                    ++nrounds;                                       // (diagnostics: sections walked again)
                }
            } else {
                section(qc, integral_constant<bool, false>{}, twoc, grow, nsteps, Alq, clq);
                eval_own(Q, wev[Q], an, bn, dn, Dl);
            }
            if (c < B)
#pragma unroll
                for (int t = 0; t < NT; ++t) { acur[t * B + c] = (c < b) ? an[t] : 0.f; bcur[t * B + c] = bn[t]; dcur[t * B + c] = dn[t]; }
        };
        if (B > 64) {
            run_section(integral_constant<int, 0>{}, integral_constant<bool, true>{});
            run_section(integral_constant<int, 1>{}, integral_constant<bool, true>{});
        } else run_section(integral_constant<int, 0>{}, integral_constant<bool, false>{});
        dense_done = true;
    }
    if (dense_done && lane == 0) wcnt_s[14] = 1;

    const int s_first = (nreps == 1 && !dense_done) ? (first_sub < nsub ? first_sub : nsub) : 0;       // prefix skip (single pass only)
    int taken_over_from = 16;                                  // skip and verify: the sub-block from which the serial wave walked the ordinary way again
    if constexpr (!kDW)
    for (int rep = 0; rep < (dense_done ? 0 : nreps); ++rep) {
        key.rep = (uint32_t)rep;
        // (skip and verify: `skipping` until a helper wave reports a move; `checked` = skipped sub-blocks whose report has been read)
        bool skipping = skip_set != 0u;
        unsigned checked = 0u;
        auto reports = [&](unsigned need) -> unsigned {         // bit s: sub-block s stays as it is | bit 16 + s: one of its markers moves
            unsigned v;
            while (true) {
                v = (unsigned)__hip_atomic_load(&wcnt_s[9], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (((v | (v >> 16)) & need) == need) break;
                __builtin_amdgcn_s_sleep(1);
            }
            return (v >> 16) & need;
        };
        int s = s_first;
#pragma unroll 1
        while (true) {
            if (skipping) {
                if (s < nsub && ((skip_set >> s) & 1u)) { ++s; continue; }
                const unsigned need = skip_set & ((1u << s) - 1u) & ~checked;      // (s = nsub: every skipped sub-block)
                if (need != 0u) {
                    const unsigned moved = reports(need);
                    checked |= need;
                    if (moved != 0u) {                           // the chain goes on the ordinary way from the first of them
                        s = __builtin_ctz(moved);
                        skipping = false;
                        taken_over_from = s;
                        if (lane == 0) wcnt_s[10] = s;
                        // (helper waves behind it are still waiting for their turn: let them run -- what they find is dropped)
                        __hip_atomic_store(&wcnt_s[8], 16, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            if (s >= nsub) break;
            const int c = 64 * s + lane;
            const bool valid = c < b;
            const int64_t j = j0 + (valid ? c : 0);
            const uint32_t marker = P->marker0 + (uint32_t)j;
            unsigned long long pending = __ballot(valid);
            const float dj = parked ? lpf[c] : A.xpx[j];
            double thr[NT], z[NT];
            float a_cur[NT], b_cur[NT], d_cur[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                a_cur[t] = acur[t * B + c]; b_cur[t] = bcur[t * B + c]; d_cur[t] = dcur[t * B + c];
                if (rep == 0) {
                    if (parked) { thr[t] = lpd[t * B + c]; z[t] = lpd[(NT + t) * B + c]; }
                    else { thr[t] = A.prep_d[(int64_t)t * p + j]; z[t] = A.prep_d[(int64_t)(NT + t) * p + j]; }
                }
                else {
                    const double u = draw_uniform(key, marker, (uint32_t)t);
                    thr[t] = is_sampler2(METHOD) ? u : log((1.0 - u) / u);
                    z[t] = draw_normal(key, marker, (uint32_t)t);
                }
            }
            float lcm[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) lcm[t] = parked ? lpf[(1 + t) * B + c] : A.prep_f[(int64_t)t * p + j];
            const MtConsts<NT> Km = consts_of(valid ? c : 0);          // (kPG: this marker's own G-dependent constants)
            const MtPre<NT> Qm = mt_precompute<METHOD, NT>(Km, dj, lcm); // x'x-only terms, once per marker (SIMD over the sub-block)
            double T[kTS][kTV];
            if constexpr (kTab) mt2_load_tab<NT>(A.mt2_tab, p, j, T);
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
                    if constexpr (is_sampler1(METHOD)) mt1_eval<NT>(Km, Qm, PriorMem{lpr_of(c), ls}, w, dj, thr, z, an, bn, dn, Dl);
                    else if constexpr (kTab) mt2_eval_tab<NT>(K, lpr_of(c), ls, w, T, thr[0], z, an, bn, dn, Dl);
                    else if constexpr (is_sampler2(METHOD)) mt2_eval<NT>(Km, lpr_of(c), ls, w, dj, thr[0], z, an, bn, dn, Dl);
                    else mega_eval<NT>(Km, Qm, w, dj, thr, z, an, bn, dn, Dl);
#pragma unroll
                    for (int t = 0; t < NT; ++t) is_event = is_event || (Dl[t] != 0.f);
                }
                ++nrounds;
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
                apply_gram_row<NT>(smem, SM, A, 64 * s + k, D, lane, nreps == 1 ? 64 * s : 0);      // :311,317
                if (pending == 0ull) break;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) { acur[t * B + c] = a_cur[t]; bcur[t * B + c] = b_cur[t]; dcur[t * B + c] = d_cur[t]; }
            ++s;
            // (the helper waves behind this sub-block may start: everything it moved is in the right-hand sides they read)
            if (skipping) __hip_atomic_store(&wcnt_s[8], s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }

    tk4 = clock64();
    int base = 0;
#pragma unroll 1
    for (int s = s_first; s < nsub; ++s) {                    // (no change before the first candidate's sub-block)
        if (((skip_set >> s) & 1u) && s < taken_over_from) continue;      // (... nor in a sub-block the helper waves found unmoved)
        const int c = 64 * s + lane;
        bool changed = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) changed = changed || ((c < b) && astart[t * B + c] != acur[t * B + c]);
        const unsigned long long cm = __ballot(changed);
        if (changed) reinterpret_cast<int*>(smem + SM.log_off)[base + __popcll(cm & ((1ull << lane) - 1ull))] = c;
        base += __popcll(cm);
    }
    if (lane == 0) wcnt_s[15] = base;
    tk5 = clock64();
    }   // wave 0
    // ---- skip and verify, the helper side: wave w evaluates the skipped ones among sub-blocks w, w + 7, w + 14 (1024-marker blocks have
    // 16 sub-blocks), in that order -- the serial loop's evaluation of a sub-block, once
    constexpr int kHS = 3;
    float hb[kHS][NT], hd[kHS][NT];                            // their markers' new beta / delta: to LDS after the walk, if they count
#pragma unroll
    for (int k = 0; k < kHS; ++k)
#pragma unroll
        for (int t = 0; t < NT; ++t) { hb[k][t] = 0.f; hd[k][t] = 0.f; }
    auto help = [&](int s, float (&ob)[NT], float (&od)[NT]) {
        const int cprev = 31 - __builtin_clz(cand_mask & ((1u << s) - 1u));      // the last candidate sub-block in front (there is one: s > first_sub)
        const int c = 64 * s + lane;
        const bool valid = c < b;
        const int64_t j = j0 + (valid ? c : 0);
        const float dj = parked ? lpf[c] : A.xpx[j];
        double thr[NT], z[NT];
        float an[NT], bn[NT], dn[NT], Dl[NT], w[NT], lcm[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            an[t] = acur[t * B + c]; bn[t] = bcur[t * B + c]; dn[t] = dcur[t * B + c]; Dl[t] = 0.f;
            if (parked) { thr[t] = lpd[t * B + c]; z[t] = lpd[(NT + t) * B + c]; }
            else { thr[t] = A.prep_d[(int64_t)t * p + j]; z[t] = A.prep_d[(int64_t)(NT + t) * p + j]; }
            lcm[t] = parked ? lpf[(1 + t) * B + c] : A.prep_f[(int64_t)t * p + j];
        }
        const MtConsts<NT> Km = consts_of(valid ? c : 0);
        const MtPre<NT> Qm = mt_precompute<METHOD, NT>(Km, dj, lcm);
        // (state, draws and the x'x-only terms are the block-entry ones whatever the serial wave does meanwhile: everything above runs
        // BEFORE the wait; only the right-hand sides have to be the ones of this sub-block's own steps)
        while (__hip_atomic_load(&wcnt_s[8], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= cprev) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int t = 0; t < NT; ++t) w[t] = rhs_lds[t * B + c] + dj * an[t];                            // :82
        bool moved = false;
        if (valid) {
            if constexpr (is_sampler1(METHOD)) mt1_eval<NT>(Km, Qm, PriorMem{lpr_of(c), ls}, w, dj, thr, z, an, bn, dn, Dl);
            else if constexpr (kTab) {
                double T[kTS][kTV];
                mt2_load_tab<NT>(A.mt2_tab, p, j, T);
                mt2_eval_tab<NT>(K, lpr_of(c), ls, w, T, thr[0], z, an, bn, dn, Dl);
            }
            else if constexpr (is_sampler2(METHOD)) mt2_eval<NT>(Km, lpr_of(c), ls, w, dj, thr[0], z, an, bn, dn, Dl);
            else mega_eval<NT>(Km, Qm, w, dj, thr, z, an, bn, dn, Dl);
#pragma unroll
            for (int t = 0; t < NT; ++t) { moved = moved || (Dl[t] != 0.f); ob[t] = bn[t]; od[t] = dn[t]; }
        }
        const bool any_moved = __any(moved);
        if (lane == 0) __hip_atomic_fetch_or(&wcnt_s[9], any_moved ? (0x10000 << s) : (1 << s), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == 0) atomicAdd(&A.counters[31], 1ull);       // (diagnostics: sub-blocks evaluated by a helper wave)
    };
    if constexpr (!kDW) {
        if (wave != 0 && skip_set != 0u) {
#pragma unroll
            for (int k = 0; k < kHS; ++k) {
                const int s = wave + 7 * k;
                if (s < 16 && ((skip_set >> s) & 1u)) help(s, hb[k], hd[k]);
            }
        }
    }
    __syncthreads();
    if constexpr (!kDW) {
        if (skip_set != 0u) {                                  // (workgroup-uniform)
            // [10]: the sub-block from which the serial wave walked the ordinary way again (16: none) -- the helpers' results in front of it are the chain's
            if (wave != 0) {
#pragma unroll
                for (int k = 0; k < kHS; ++k) {
                    const int s = wave + 7 * k;
                    if (s < 16 && ((skip_set >> s) & 1u) && s < wcnt_s[10] && 64 * s + lane < b) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) { bcur[t * B + 64 * s + lane] = hb[k][t]; dcur[t * B + 64 * s + lane] = hd[k][t]; }
                    }
                }
            }
            if (tid == 0 && wcnt_s[10] != 16) atomicAdd(&A.counters[30], 1ull);      // (diagnostics: blocks in which a skipped marker moved)
            __syncthreads();
        }
    }
    const int nfin = wcnt_s[15];
    if (A.b_next > 0 && cross_dma && wcnt_s[14] != 0) {
        // dense walk with the cross-Gram rows in LDS: every marker is an entry (alpha_old - alpha_new = 0: exact no-op); one
        // thread per (trait, column of the next block), the chain in marker order as in corr_phase
        const float* crossL = reinterpret_cast<const float*>(smem + SM.cross_off);
        for (int i = tid; i < NT * B; i += kStepThreads) rhs_lds[i] = astart[i] - acur[i];
        __syncthreads();
        for (int i = tid; i < NT * B; i += kStepThreads) {
            const int t = i / B, c = i - t * B;
            const float* dl = rhs_lds + t * B;
            float corr = 0.f;
            int e = 0;
#pragma unroll 1
            for (; e + 8 <= b; e += 8) {
                const float4 d0 = *reinterpret_cast<const float4*>(dl + e), d1 = *reinterpret_cast<const float4*>(dl + e + 4);
                float g[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) g[u] = crossL[(e + u) * B + c];
                corr = fmaf(d0.x, g[0], corr); corr = fmaf(d0.y, g[1], corr); corr = fmaf(d0.z, g[2], corr); corr = fmaf(d0.w, g[3], corr);
                corr = fmaf(d1.x, g[4], corr); corr = fmaf(d1.y, g[5], corr); corr = fmaf(d1.z, g[6], corr); corr = fmaf(d1.w, g[7], corr);
            }
            for (; e < b; ++e) corr = fmaf(dl[e], crossL[e * B + c], corr);
            A.corr_out[i] = corr;
        }
    } else if (A.b_next > 0) corr_phase<NT>(smem, SM, A, nfin);
    // ---- global stores LAST (a barrier after a global store waits for the store): the change list for the next update
    // role, then the block's state (beta / delta of every marker are new draws; alpha changes only where an event happened)
    {
        const int* fin = reinterpret_cast<const int*>(smem + SM.log_off);
        for (int e = tid; e < nfin; e += kStepThreads) {
            const int ce = fin[e];
            A.ev_out->idx[e] = (int32_t)(j0 + ce);
#pragma unroll
            for (int t = 0; t < NT; ++t) A.ev_out->delta[t][e] = astart[t * B + ce] - acur[t * B + ce];
        }
    }
    for (int c = tid; c < b; c += kStepThreads) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float a_fin = acur[t * B + c];
            if (a_fin != astart[t * B + c]) A.alpha[(int64_t)t * p + j0 + c] = a_fin;
            A.beta[(int64_t)t * p + j0 + c] = bcur[t * B + c];
            delta[(int64_t)t * p + j0 + c]  = dcur[t * B + c];
        }
    }
    if (tid == 0) {
        A.ev_out->count = (int32_t)nfin;
        atomicAdd(&A.counters[0], (unsigned long long)nfin);
        atomicAdd(&A.counters[2], (unsigned long long)(tk1 - tk0));      // phase cycle counts (diagnostics)
        atomicAdd(&A.counters[4], (unsigned long long)(tk3 - tk1));
        atomicAdd(&A.counters[5], (unsigned long long)(tk4 - tk3));
        atomicAdd(&A.counters[6], (unsigned long long)(tk5 - tk4));
        atomicAdd(&A.counters[7], (unsigned long long)nrounds);
    }
}


}  // namespace jw
