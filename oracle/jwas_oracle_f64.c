/*
 * jwas_oracle_f64.c -- CPU ORACLE, Float64 mode (test infrastructure only; see jwas_oracle.h for the contract).
 *
 * runMCMC(double_precision=true) (JWAS.jl:349-366; genotypes read as Float64, readgenotypes.jl:298,345) makes every array
 * of the marker path Float64, so the scalar kernels run with T = Float64 throughout.  This file restates them that way, in
 * the reference's LITERAL non-block order -- per marker: dot(x, ycorr), the scalar update, axpy -- which is what the
 * device's Float64 block form (csrc/f64_path.hpp) is compared with:
 *   orc64_bayesabc_sweep   BayesABC!            (markers/BayesianAlphabet/BayesABC.jl:60-80, kernel :24-58)
 *   orc64_bayesr_sweep     BayesR!              (markers/BayesianAlphabet/BayesR.jl:45-97)
 *   orc64_mt1_sweep        _MTBayesABC_samplerI! (markers/BayesianAlphabet/MTBayesABC.jl:57-127)
 * Draws: the same counter RNG as the Float32 oracle (orc_uniform / orc_normal: Philox4x32-10 keyed by the seed, counter =
 * (global marker, iteration, repetition, slot + 16 trait)), so a Float64 chain and a Float32 chain see the same draws.
 * Parity: unpinned against Julia output (no Julia here, no golden vectors for sampler output in the reference) -- the same
 * status as the Float32 oracle; pinned to it instead: on data where Float32 arithmetic is exact enough the two oracles
 * give the same indicator trajectories (tests/test_oracle_kat.py::test_float64_oracle_tracks_the_float32_oracle).
 */
#include "jwas_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC64_MAXT 8

static double dot64(const double* a, const double* b, int64_t n)
{
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

static void axpy64(double a, const double* x, double* y, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) y[i] = fma(a, x[i], y[i]);      /* BLAS.axpy! (FMA kernels) */
}

void orc64_xpx(const double* X, int64_t n, int64_t p, int64_t ld, double* out)
{
    for (int64_t j = 0; j < p; ++j) out[j] = dot64(X + j * ld, X + j * ld, n);     /* getXpRinvX, tools4genotypes.jl:28-31 */
}

/* BayesA/B/C.  var_effects, pi: p entries each.  delta: double 0/1. */
int orc64_bayesabc_sweep(const double* X, int64_t n, int64_t p, int64_t ld, const double* xpx,
                         double* r, double* alpha, double* beta, double* delta,
                         double vare, const double* var_effects, const double* pi,
                         uint64_t seed, uint32_t iter, uint32_t marker0)
{
    if (n <= 0 || p <= 0 || ld < n || !(vare > 0.0)) return -1;
    const double invVarRes = 1.0 / vare;                                                     /* :69 */
    for (int64_t j = 0; j < p; ++j) {
        const double* x = X + j * ld;
        const uint32_t m = marker0 + (uint32_t)j;
        const double logPi = log(pi[j]), logPiComp = log(1.0 - pi[j]);                       /* :67-68 */
        const double invVarEffect = 1.0 / var_effects[j], logVarEffect = log(var_effects[j]); /* :70-71 */
        const double xRinvy = dot64(x, r, n);                                                /* :76 */
        const double rhs = (xRinvy + xpx[j] * alpha[j]) * invVarRes;                         /* :36 */
        const double lhs = xpx[j] * invVarRes + invVarEffect;                                /* :37 */
        const double invLhs = 1.0 / lhs;                                                     /* :38 */
        const double gHat = rhs * invLhs;                                                    /* :39 */
        const double logDelta1 = -0.5 * (log(lhs) + logVarEffect - gHat * rhs) + logPiComp;  /* :40 */
        const double probDelta1 = 1.0 / (1.0 + exp(logPi - logDelta1));                      /* :41 */
        const double oldAlpha = alpha[j];
        const double u = orc_uniform(seed, m, iter, 0, 0), z = orc_normal(seed, m, iter, 0, 0);
        if (u < probDelta1) {                                                                /* :44-48 */
            delta[j] = 1.0;
            beta[j] = gHat + z * sqrt(invLhs);
            alpha[j] = beta[j];
            axpy64(oldAlpha - alpha[j], x, r, n);
        } else {                                                                             /* :50-56 */
            if (oldAlpha != 0.0) axpy64(oldAlpha, x, r, n);
            delta[j] = 0.0;
            beta[j] = z * sqrt(var_effects[j]);
            alpha[j] = 0.0;
        }
    }
    return 0;
}

/* BayesR.  pi: 4 entries, or p x 4 (pi_is_matrix).  delta: classes 1..4. */
int orc64_bayesr_sweep(const double* X, int64_t n, int64_t p, int64_t ld, const double* xpx,
                       double* r, double* alpha, int32_t* delta,
                       double vare, double sigma_sq, const double* pi, int pi_is_matrix, const double* gamma,
                       uint64_t seed, uint32_t iter, uint32_t marker0)
{
    if (n <= 0 || p <= 0 || ld < n || !(vare > 0.0)) return -1;
    if (!(sigma_sq > 0.0)) return -2;                                                        /* :50 */
    const double invVarRes = 1.0 / vare;
    for (int64_t j = 0; j < p; ++j) {
        const double* x = X + j * ld;
        const uint32_t m = marker0 + (uint32_t)j;
        const double* pj = pi_is_matrix ? pi + 4 * j : pi;
        const double rhs = (dot64(x, r, n) + xpx[j] * alpha[j]) * invVarRes;                 /* :60 */
        const double oldAlpha = alpha[j];
        double lp[4], probs[4];
        lp[0] = log(pj[0]);                                                                  /* :64 */
        for (int k = 1; k < 4; ++k) {                                                        /* :65-72 */
            const double varEffect = gamma[k] * sigma_sq;
            const double invVarEffect = 1.0 / varEffect;
            const double lhs = xpx[j] * invVarRes + invVarEffect;
            const double invLhs = 1.0 / lhs;
            const double betaHat = invLhs * rhs;
            lp[k] = 0.5 * (log(invLhs) - log(varEffect) + betaHat * rhs) + log(pj[k]);
        }
        double mx = lp[0];
        for (int k = 1; k < 4; ++k) if (lp[k] > mx) mx = lp[k];
        double se = 0.0;
        for (int k = 0; k < 4; ++k) se += exp(lp[k] - mx);
        const double log_norm = mx + log(se);                                                /* bayesr_logsumexp :1-4 */
        for (int k = 0; k < 4; ++k) probs[k] = exp(lp[k] - log_norm);                        /* :75-77 */
        const double u = orc_uniform(seed, m, iter, 0, 0), z = orc_normal(seed, m, iter, 0, 0);
        int cls = 0;                                                                         /* rand(Categorical(probs)) :79 */
        double cp = probs[0];
        while (cp <= u && cls < 3) { ++cls; cp += probs[cls]; }
        delta[j] = cls + 1;                                                                  /* :80 */
        if (cls == 0) {                                                                      /* :82-86 */
            if (oldAlpha != 0.0) axpy64(oldAlpha, x, r, n);
            alpha[j] = 0.0;
        } else {                                                                             /* :88-94 */
            const double varEffect = gamma[cls] * sigma_sq;
            const double lhs = xpx[j] * invVarRes + 1.0 / varEffect;
            const double invLhs = 1.0 / lhs;
            alpha[j] = invLhs * rhs + z * sqrt(invLhs);
            axpy64(oldAlpha - alpha[j], x, r, n);
        }
    }
    return 0;
}

static int inv64(const double* A, int t, double* Ainv)      /* inv(::Matrix{Float64}): Gauss-Jordan, partial pivoting */
{
    double M[ORC64_MAXT][2 * ORC64_MAXT];
    for (int i = 0; i < t; ++i) for (int j = 0; j < t; ++j) { M[i][j] = A[i * t + j]; M[i][t + j] = (i == j); }
    for (int c = 0; c < t; ++c) {
        int piv = c;
        for (int i = c + 1; i < t; ++i) if (fabs(M[i][c]) > fabs(M[piv][c])) piv = i;
        if (M[piv][c] == 0.0) return -1;
        if (piv != c) for (int j = 0; j < 2 * t; ++j) { double tmp = M[c][j]; M[c][j] = M[piv][j]; M[piv][j] = tmp; }
        const double d = M[c][c];
        for (int j = 0; j < 2 * t; ++j) M[c][j] /= d;
        for (int i = 0; i < t; ++i) if (i != c) { const double f = M[i][c]; if (f != 0.0) for (int j = 0; j < 2 * t; ++j) M[i][j] -= f * M[c][j]; }
    }
    for (int i = 0; i < t; ++i) for (int j = 0; j < t; ++j) Ainv[i * t + j] = M[i][t + j];
    return 0;
}

/* Multi-trait sampler I.  r: [t][ldr] residuals; alpha/beta/delta: [t][p]; vare, var_effect: t x t; log_prior: 2^t. */
int orc64_mt1_sweep(int t, const double* X, int64_t n, int64_t p, int64_t ld, const double* xpx,
                    double* r, int64_t ldr, double* alpha, double* beta, double* delta,
                    const double* vare, const double* var_effect, const double* log_prior,
                    uint64_t seed, uint32_t iter, uint32_t marker0)
{
    if (t < 2 || t > ORC64_MAXT || n <= 0 || p <= 0 || ld < n) return -1;
    double Rinv[ORC64_MAXT * ORC64_MAXT], Ginv[ORC64_MAXT * ORC64_MAXT];
    if (inv64(vare, t, Rinv) || inv64(var_effect, t, Ginv)) return -2;                       /* :66-67 */
    for (int64_t j = 0; j < p; ++j) {
        const double* x = X + j * ld;
        const uint32_t m = marker0 + (uint32_t)j;
        const double d = xpx[j];
        double b[ORC64_MAXT], newa[ORC64_MAXT], olda[ORC64_MAXT], dl[ORC64_MAXT], w[ORC64_MAXT];
        for (int k = 0; k < t; ++k) {                                                        /* :78-83 */
            b[k] = beta[k * p + j];
            olda[k] = newa[k] = alpha[k * p + j];
            dl[k] = delta[k * p + j];
            w[k] = dot64(x, r + k * ldr, n) + d * olda[k];
        }
        for (int k = 0; k < t; ++k) {                                                        /* :85-121 */
            const double Ginv11 = Ginv[k * t + k];
            const double C11 = Ginv11 + Rinv[k * t + k] * d;                                 /* :89 */
            double rhs0 = 0.0, c12b = 0.0, wR = 0.0;
            for (int q = 0; q < t; ++q) {
                wR = wR + w[q] * Rinv[q * t + k];                                            /* :96 */
                if (q == k) continue;
                const double C12q = Ginv[k * t + q] + (d * dl[q]) * Rinv[k * t + q];         /* :90 */
                rhs0 = rhs0 + Ginv[k * t + q] * b[q];                                        /* :93 */
                c12b = c12b + C12q * b[q];
            }
            rhs0 = -rhs0;
            const double invLhs0 = 1.0 / Ginv11, gHat0 = rhs0 * invLhs0;                     /* :92,:94 */
            const double invLhs1 = 1.0 / C11, rhs1 = wR - c12b, gHat1 = rhs1 * invLhs1;      /* :95-97 */
            unsigned s0 = 0u;
            for (int q = 0; q < t; ++q) if (q != k && dl[q] != 0.0) s0 |= 1u << q;
            const unsigned s1 = s0 | (1u << k);
            const double logDelta0 = -0.5 * (log(Ginv11) - gHat0 * gHat0 * Ginv11) + log_prior[s0];      /* :104 */
            const double logDelta1 = -0.5 * (log(C11) - gHat1 * gHat1 * C11) + log_prior[s1];            /* :105 */
            const double probDelta1 = 1.0 / (1.0 + exp(logDelta0 - logDelta1));              /* :107 */
            const double u = orc_uniform(seed, m, iter, 0, (uint32_t)k), z = orc_normal(seed, m, iter, 0, (uint32_t)k);
            if (u < probDelta1) {                                                            /* :108-111 */
                dl[k] = 1.0;
                b[k] = newa[k] = gHat1 + z * sqrt(invLhs1);
                axpy64(olda[k] - newa[k], x, r + k * ldr, n);
            } else {                                                                         /* :112-119 */
                b[k] = gHat0 + z * sqrt(invLhs0);
                dl[k] = 0.0;
                newa[k] = 0.0;
                if (olda[k] != 0.0) axpy64(olda[k], x, r + k * ldr, n);
            }
        }
        for (int k = 0; k < t; ++k) { beta[k * p + j] = b[k]; delta[k * p + j] = dl[k]; alpha[k * p + j] = newa[k]; }
    }
    return 0;
}

/* Block form of BayesABC! with within-block repetitions (BayesABC_block!, BayesABC.jl:118-188), T = Float64: uniform blocks
 * of `bs` markers, nreps <= 0 = every block its own size (:153).  The block's Gram is formed here (X_b'X_b, :263-266). */
int orc64_bayesabc_block_sweep(const double* X, int64_t n, int64_t p, int64_t ld, const double* xpx, int64_t bs, int nreps_arg,
                               double* r, double* alpha, double* beta, double* delta,
                               double vare, const double* var_effects, const double* pi,
                               uint64_t seed, uint32_t iter, uint32_t marker0)
{
    if (n <= 0 || p <= 0 || ld < n || !(vare > 0.0) || bs < 1) return -1;
    const double invVarRes = 1.0 / vare;
    double* G = (double*)malloc(sizeof(double) * (size_t)bs * bs);
    double* rhs_b = (double*)malloc(sizeof(double) * (size_t)bs);
    double* a_old = (double*)malloc(sizeof(double) * (size_t)bs);
    for (int64_t j0 = 0; j0 < p; j0 += bs) {
        const int64_t b = (j0 + bs <= p) ? bs : p - j0;
        for (int64_t a = 0; a < b; ++a)
            for (int64_t c = 0; c < b; ++c) G[a * b + c] = dot64(X + (j0 + a) * ld, X + (j0 + c) * ld, n);
        for (int64_t k = 0; k < b; ++k) { rhs_b[k] = dot64(X + (j0 + k) * ld, r, n); a_old[k] = alpha[j0 + k]; }      /* :152 */
        const int nreps = nreps_arg > 0 ? nreps_arg : (int)b;                                                   /* :153 */
        for (int rep = 0; rep < nreps; ++rep)
            for (int64_t k = 0; k < b; ++k) {                                                                   /* :155-178 */
                const int64_t j = j0 + k;
                const uint32_t m = marker0 + (uint32_t)j;
                const double rhs = (rhs_b[k] + xpx[j] * alpha[j]) * invVarRes;
                const double lhs = xpx[j] * invVarRes + 1.0 / var_effects[j];
                const double invLhs = 1.0 / lhs, gHat = rhs * invLhs;
                const double logDelta1 = -0.5 * (log(lhs) + log(var_effects[j]) - gHat * rhs) + log(1.0 - pi[j]);
                const double probDelta1 = 1.0 / (1.0 + exp(log(pi[j]) - logDelta1));
                const double oldAlpha = alpha[j];
                const double u = orc_uniform(seed, m, iter, (uint32_t)rep, 0), z = orc_normal(seed, m, iter, (uint32_t)rep, 0);
                double coef;
                if (u < probDelta1) { delta[j] = 1.0; beta[j] = gHat + z * sqrt(invLhs); alpha[j] = beta[j]; coef = oldAlpha - alpha[j]; }
                else { delta[j] = 0.0; beta[j] = z * sqrt(var_effects[j]); alpha[j] = 0.0; coef = oldAlpha; }
                if (coef != 0.0) axpy64(coef, G + k * b, rhs_b, b);                                             /* :169,172 */
            }
        for (int64_t k = 0; k < b; ++k) {                                                                       /* :181-185 */
            const double d = a_old[k] - alpha[j0 + k];
            if (d != 0.0) axpy64(d, X + (j0 + k) * ld, r, n);
        }
    }
    free(G); free(rhs_b); free(a_old);
    return 0;
}


/* The general Float64 block form of BayesABC! (BayesABC_block!, BayesABC.jl:118-188; BayesABC_block_independent!, :190-255):
 * any partition `starts` (nb + 1 entries, 0-based, starts[nb] = p), residual weights w (NULL = unit: X_b'R^-1 X_b, X_b'R^-1 r and
 * x'R^-1 x with R^-1 = diag(w), tools4genotypes.jl:59-78,237-275), nreps <= 0 = every block its own size (:153),
 * independent != 0: every block's right-hand side from the residual at the START of the sweep, the residual reconciled
 * afterwards in (block, marker) order (:251-253).  xpx: x'R^-1 x (orc64_xpx_w). */
void orc64_xpx_w(const double* X, int64_t n, int64_t p, int64_t ld, const double* w, double* out)
{
    for (int64_t j = 0; j < p; ++j) {
        const double* x = X + j * ld;
        double s = 0.0;
        for (int64_t i = 0; i < n; ++i) s += (x[i] * x[i]) * (w ? w[i] : 1.0);
        out[j] = s;
    }
}

int orc64_bayesabc_block_sweep_ex(const double* X, int64_t n, int64_t p, int64_t ld, const double* xpx, const double* w,
                                  const int64_t* starts, int64_t nb, int nreps_arg, int independent,
                                  double* r, double* alpha, double* beta, double* delta,
                                  double vare, const double* var_effects, const double* pi,
                                  uint64_t seed, uint32_t iter, uint32_t marker0)
{
    if (n <= 0 || p <= 0 || ld < n || !(vare > 0.0) || nb < 1 || !starts) return -1;
    const double invVarRes = 1.0 / vare;
    int64_t bmax = 0;
    for (int64_t k = 0; k < nb; ++k) if (starts[k + 1] - starts[k] > bmax) bmax = starts[k + 1] - starts[k];
    double* G = (double*)malloc(sizeof(double) * (size_t)bmax * bmax);
    double* rhs_b = (double*)malloc(sizeof(double) * (size_t)bmax);
    double* a_old = (double*)malloc(sizeof(double) * (size_t)p);
    double* rw = (double*)malloc(sizeof(double) * (size_t)n);
    for (int64_t j = 0; j < p; ++j) a_old[j] = alpha[j];
    if (independent) for (int64_t i = 0; i < n; ++i) rw[i] = r[i] * (w ? w[i] : 1.0);      /* the snapshot, weighted once */
    for (int64_t blk = 0; blk < nb; ++blk) {
        const int64_t j0 = starts[blk], b = starts[blk + 1] - j0;
        for (int64_t a = 0; a < b; ++a)
            for (int64_t c = 0; c < b; ++c) {
                const double *xa = X + (j0 + a) * ld, *xc = X + (j0 + c) * ld;
                double sum = 0.0;
                for (int64_t i = 0; i < n; ++i) sum += (xa[i] * xc[i]) * (w ? w[i] : 1.0);
                G[a * b + c] = sum;
            }
        if (!independent) for (int64_t i = 0; i < n; ++i) rw[i] = r[i] * (w ? w[i] : 1.0);
        for (int64_t k = 0; k < b; ++k) rhs_b[k] = dot64(X + (j0 + k) * ld, rw, n);                             /* block_rhs! */
        const int nreps = nreps_arg > 0 ? nreps_arg : (int)b;                                                   /* :153 */
        for (int rep = 0; rep < nreps; ++rep)
            for (int64_t k = 0; k < b; ++k) {                                                                   /* :155-178 */
                const int64_t j = j0 + k;
                const uint32_t m = marker0 + (uint32_t)j;
                const double rhs = (rhs_b[k] + xpx[j] * alpha[j]) * invVarRes;
                const double lhs = xpx[j] * invVarRes + 1.0 / var_effects[j];
                const double invLhs = 1.0 / lhs, gHat = rhs * invLhs;
                const double logDelta1 = -0.5 * (log(lhs) + log(var_effects[j]) - gHat * rhs) + log(1.0 - pi[j]);
                const double probDelta1 = 1.0 / (1.0 + exp(log(pi[j]) - logDelta1));
                const double oldAlpha = alpha[j];
                const double u = orc_uniform(seed, m, iter, (uint32_t)rep, 0), z = orc_normal(seed, m, iter, (uint32_t)rep, 0);
                double coef;
                if (u < probDelta1) { delta[j] = 1.0; beta[j] = gHat + z * sqrt(invLhs); alpha[j] = beta[j]; coef = oldAlpha - alpha[j]; }
                else { delta[j] = 0.0; beta[j] = z * sqrt(var_effects[j]); alpha[j] = 0.0; coef = oldAlpha; }
                if (coef != 0.0) axpy64(coef, G + k * b, rhs_b, b);                                             /* :169,172 */
            }
        if (!independent)
            for (int64_t k = 0; k < b; ++k) {                                                                   /* :181-185 */
                const double d = a_old[j0 + k] - alpha[j0 + k];
                if (d != 0.0) axpy64(d, X + (j0 + k) * ld, r, n);
            }
    }
    if (independent)
        for (int64_t j = 0; j < p; ++j) {                                                                       /* :251-253 */
            const double d = a_old[j] - alpha[j];
            if (d != 0.0) axpy64(d, X + j * ld, r, n);
        }
    free(G); free(rhs_b); free(a_old); free(rw);
    return 0;
}
