/*
 * jwas_oracle.c -- CPU ORACLE (test infrastructure only; see jwas_oracle.h for the contract).
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no implicit FMA contraction, every fused
 * operation below is an explicit fmaf/fma).
 */
#include "jwas_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3")   */
/* ------------------------------------------------------------------------------------------ */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline double u52(uint32_t lo, uint32_t hi)
{
    uint64_t k = (((uint64_t)hi << 32) | lo) >> 12;            /* 52 random bits           */
    return ((double)k + 0.5) * 0x1.0p-52;                      /* exact; in (0,1)          */
}

static inline void draw_block(uint64_t seed, uint32_t marker, uint32_t iter, uint32_t rep,
                              uint32_t trait, uint32_t slot, uint32_t out[4])
{
    uint32_t ctr[4] = { marker, iter, rep, slot + 16u * trait };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    orc_philox4x32_10(ctr, key, out);
}

double orc_uniform(uint64_t seed, uint32_t marker, uint32_t iter, uint32_t rep, uint32_t trait)
{
    uint32_t w[4];
    draw_block(seed, marker, iter, rep, trait, 0u, w);
    return u52(w[0], w[1]);
}

double orc_normal(uint64_t seed, uint32_t marker, uint32_t iter, uint32_t rep, uint32_t trait)
{
    uint32_t w[4];
    draw_block(seed, marker, iter, rep, trait, 1u, w);
    double u1 = u52(w[0], w[1]), u2 = u52(w[2], w[3]);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

/* ------------------------------------------------------------------------------------------ */
/* inner products                                                                             */
/* ------------------------------------------------------------------------------------------ */
/* Residual weights R^-1 (mme.invweights).  Every inner product of the path is a' R^-1 b with the weight carried by
 * the RIGHT operand, b_i -> fl32(w_i * b_i):  x'R^-1 r (BayesABC.jl:76 with xRinvArray, block_rhs! tools4genotypes.jl:
 * 59-78), x'R^-1 x (getXpRinvX, :28-31), X_b'R^-1 X_b (:263-266) and the lookahead cross-Grams.  NULL = unit weights.
 * The pointer is borrowed (test infrastructure: one oracle context per process). */
static const float* g_rinv = NULL;
void orc_set_weights(const float* rinv) { g_rinv = rinv; }

/* ORC_ACC_DEVICE: the block right-hand side x'r summed in the ORDER the device sums it (jwas.jl_amd/csrc/sweep.hpp
 * update_role): 256-row slices; in a slice lane l holds rows 4l..4l+3 (product, then three fused multiply-adds, fp64);
 * the 64 lane values are folded pairwise at distances 32, 16, 8, 4, 2, 1 (butterfly8); the slices of a row group (spg of
 * them) are added in order, then the row groups in order; rounded to fp32 once.  The values are the same exact fp32
 * products as ORC_ACC_F64 -- only the association of the fp64 additions differs -- so with this mode the oracle's chain
 * and the device's agree in EVERY bit.  orc_set_device_order(spg) (harness state): spg = slices per row group of the
 * device context (jwas_hip_update_geometry). */
static int g_dev_spg = 8;
void orc_set_device_order(int spg) { g_dev_spg = spg > 0 ? spg : 8; }
static float dot_device_order(const float* x, const float* r, int64_t n, const float* w)
{
    const int64_t nsl = (n + 255) / 256, nrg = (nsl + g_dev_spg - 1) / g_dev_spg;
    double total = 0.0;
    for (int64_t rg = 0; rg < nrg; ++rg) {
        double P = 0.0;
        for (int wv = 0; wv < 8; ++wv) {
            double T = 0.0;
            const int64_t sl = rg * g_dev_spg + wv;
            if (wv < g_dev_spg && sl < nsl) {
                double L[64];
                for (int l = 0; l < 64; ++l) {
                    const int64_t i0 = sl * 256 + 4 * l;
                    double acc = 0.0;
                    for (int q = 0; q < 4; ++q) {
                        const int64_t i = i0 + q;
                        const float xv = i < n ? x[i] : 0.0f;
                        const float rw = i < n ? (w ? r[i] * w[i] : r[i]) : 0.0f;
                        acc = (q == 0) ? (double)xv * (double)rw : fma((double)xv, (double)rw, acc);
                    }
                    L[l] = acc;
                }
                for (int d = 32; d >= 1; d >>= 1) for (int i = 0; i < d; ++i) L[i] = L[i] + L[i + d];
                T = L[0];
            }
            P += T;
        }
        total += P;
    }
    return (float)total;
}

static float dot_acc(const float* a, const float* b, int64_t n, int acc)
{
    if (acc == ORC_ACC_DEVICE) return dot_device_order(a, b, n, g_rinv);
    if (g_rinv) {
        double s = 0.0;
        for (int64_t i = 0; i < n; ++i) s += (double)a[i] * (double)(b[i] * g_rinv[i]);
        return (float)s;
    }
    if (acc == ORC_ACC_F64) {
        double s = 0.0;
        for (int64_t i = 0; i < n; ++i) s += (double)a[i] * (double)b[i];   /* exact products */
        return (float)s;
    }
    float s = 0.0f;
    for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* The block right-hand side x'R^-1 r on 2-BIT PACKED storage, in the packed update role's own order (csrc/update_role.hpp
 * update_role_wide): the centring is factored out of the sum,
 *     sum_i (v_i - mu) w_i r_i = sum_i c_i w_i r_i - mu (R - M),   c_i = code (0 for a missing one), R = sum_i w_i r_i, M = sum_missing w_i r_i
 * (decode_marker!, streaming_genotypes.jl:978-1002; uncentred: + mu M), every sum in fp64 over exact products, rounded to fp32
 * once.  orc_set_packed_source (harness state): the codes of the matrix X the sweeps are called with, one code 0..3 per byte,
 * column-major [p][n]; NULL = off (the decoded matrix's own products, as for dense storage). */
static const uint8_t* g_pk_codes = NULL;
static const float* g_pk_means = NULL;
static const float* g_pk_X = NULL;
static int64_t g_pk_ld = 0, g_pk_n = 0;
static int g_pk_centered = 1;
void orc_set_packed_source(const uint8_t* codes, const float* means, int centered, const float* X, int64_t n, int64_t ld)
{
    g_pk_codes = codes; g_pk_means = means; g_pk_centered = centered; g_pk_X = X; g_pk_n = n; g_pk_ld = ld;
}
static float dot_xr(const float* x, const float* r, int64_t n, int acc)
{
    if (!g_pk_codes) return dot_acc(x, r, n, acc);
    const int64_t j = (x - g_pk_X) / g_pk_ld;
    const uint8_t* c = g_pk_codes + j * g_pk_n;
    const double mu = (double)g_pk_means[j];
    double s1 = 0.0, R = 0.0, M = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const float wr = g_rinv ? r[i] * g_rinv[i] : r[i];
        if (c[i] == 3) M += (double)wr; else s1 = fma((double)c[i], (double)wr, s1);
        R += (double)wr;
    }
    return (float)(g_pk_centered ? fma(-mu, R - M, s1) : fma(mu, M, s1));
}

static void axpy_f32(float a, const float* x, float* y, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) y[i] = fmaf(a, x[i], y[i]);
}

void orc_xpx(const float* X, int64_t n, int64_t p, int64_t ld, float* xpx, int acc)
{
    for (int64_t j = 0; j < p; ++j) xpx[j] = dot_acc(X + j * ld, X + j * ld, n, acc);
}

/* Cross-Gram X_prev' X_this as the lookahead correction uses it (la_block_rhs): out[a * b + c] = x_{jp+a}' x_{j0+c}. */
void orc_cross_gram(const float* X, int64_t n, int64_t ld, int64_t jp, int64_t bp, int64_t j0, int64_t b, float* out, int acc)
{
    for (int64_t a = 0; a < bp; ++a)
        for (int64_t c = 0; c < b; ++c) out[a * b + c] = dot_acc(X + (jp + a) * ld, X + (j0 + c) * ld, n, acc);
}

void orc_gram(const float* X, int64_t n, int64_t ld, int64_t j0, int64_t b, float* G, int acc)
{
    for (int64_t a = 0; a < b; ++a)
        for (int64_t c = 0; c <= a; ++c) {
            float v = dot_acc(X + (j0 + a) * ld, X + (j0 + c) * ld, n, acc);
            G[a * b + c] = v;
            G[c * b + a] = v;
        }
}

void orc_residual_minus_xalpha(const float* X, int64_t n, int64_t p, int64_t ld,
                               const float* alpha, float* r)
{
    for (int64_t j = 0; j < p; ++j)
        if (alpha[j] != 0.0f) axpy_f32(-alpha[j], X + j * ld, r, n);
}

static inline float logf_via_double(float x) { return (float)log((double)x); }

/* ------------------------------------------------------------------------------------------ */
/* BayesA/B/C scalar kernel -- bayesabc_update_marker! (BayesABC.jl:24-58)                     */
/* s = x_j' r (fp32).  Returns the axpy coefficient (alpha_old - alpha_new); 0 => no axpy.     */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    float  ie;        /* invVarRes = 1/vare                  (BayesABC.jl:69)                  */
} abc_sweep_consts;

/* Rule D (the device's definition for sweeps under a UNIFORM prior pi = 0 -- RR-BLUP, BayesA, BayesL, the reference's own
 * benchmark setting; csrc/kernels.hpp AbcMarker::rule_d): every marker is included whatever its rhs, and its new effect is
 * ONE fused multiply-add of the block rhs,  alpha = fmaf(c1, s, c0),  c1 = fl32(fl64(ie) fl64(1/lhs)),
 * c0 = fl32(fl64(ie) fl64(1/lhs) (fl64(d) fl64(alpha_old)) + z sqrt(1/lhs))  -- the same conditional mean and draw as
 * :36,:39,:46 in one rounding instead of four.  The caller (oracle.py) switches it on exactly for such sweeps;
 * orc_set_abc_rule_d(0) is the literal operation order (compared in tests/test_oracle_kat.py). */
static int g_abc_rule_d = 0;
void orc_set_abc_rule_d(int on) { g_abc_rule_d = on; }

static inline float abc_update(float s, float d, float* alpha, float* beta, float* delta,
                               float ie, float var_j, double pi_j, double u, double z)
{
    if (g_abc_rule_d && pi_j == 0.0) {
        const float a_old = *alpha;
        const float lhs = d * ie + 1.0f / var_j;                       /* :37 */
        const float invLhs = 1.0f / lhs;                               /* :38 */
        const double k1 = (double)ie * (double)invLhs;
        const float c1 = (float)k1;
        const float c0 = (float)(k1 * ((double)d * (double)a_old) + z * (double)sqrtf(invLhs));
        *delta = 1.0f;
        *beta = fmaf(c1, s, c0);
        *alpha = *beta;
        return a_old - *alpha;
    }
    /* per-sweep vectors of BayesABC.jl:66-71, evaluated per marker (same values) */
    const double lp0 = log(pi_j);                 /* logPi[j]      (Float64)                    */
    const double lp1 = log(1.0 - pi_j);           /* logPiComp[j]  (Float64)                    */
    const float  iv  = 1.0f / var_j;              /* invVarEffects[j]                           */
    const float  lv  = logf_via_double(var_j);    /* logVarEffects[j]                           */

    const float a_old  = *alpha;
    const float rhs    = (s + d * a_old) * ie;                        /* :36 */
    const float lhs    = d * ie + iv;                                 /* :37 */
    const float invLhs = 1.0f / lhs;                                  /* :38 */
    const float gHat   = rhs * invLhs;                                /* :39 */
    const float inner  = (logf_via_double(lhs) + lv) - gHat * rhs;    /* Float32 part of :40 */
    const double logDelta1  = -0.5 * (double)inner + lp1;             /* :40 */
    const double probDelta1 = 1.0 / (1.0 + exp(lp0 - logDelta1));     /* :41 */

    if (u < probDelta1) {                                             /* :44 */
        *delta = 1.0f;
        *beta  = (float)((double)gHat + z * (double)sqrtf(invLhs));   /* :46 */
        *alpha = *beta;
        return a_old - *alpha;                                        /* :48 */
    }
    *delta = 0.0f;
    *beta  = (float)(z * (double)sqrtf(var_j));                       /* :54 */
    *alpha = 0.0f;
    return a_old;                                                     /* :50-52 (0 => skipped) */
}

static int abc_args_ok(int64_t n, int64_t p, int64_t ld, float vare)
{
    return n > 0 && p >= 0 && ld >= n && vare > 0.0f;
}

int orc_bayesabc_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                       float* r, float* alpha, float* beta, float* delta,
                       float vare, const float* var_effects, const double* pi,
                       uint64_t seed, uint32_t iter, uint32_t marker0, int acc)
{
    if (!abc_args_ok(n, p, ld, vare)) return -1;
    const float ie = 1.0f / vare;
    for (int64_t j = 0; j < p; ++j) {                                  /* BayesABC.jl:73-79 */
        const float* x = X + j * ld;
        const uint32_t m = marker0 + (uint32_t)j;
        const float s = dot_xr(x, r, n, acc);                         /* :76 */
        const double u = orc_uniform(seed, m, iter, 0, 0);
        const double z = orc_normal(seed, m, iter, 0, 0);
        const float a = abc_update(s, xpx[j], &alpha[j], &beta[j], &delta[j], ie,
                                   var_effects[j], pi[j], u, z);
        if (a != 0.0f) axpy_f32(a, x, r, n);
    }
    return 0;
}

static int blocks_ok(const int64_t* bs, int64_t nblocks, int64_t p)
{
    if (nblocks < 1 || bs[0] != 0) return 0;
    for (int64_t i = 1; i < nblocks; ++i) if (bs[i] <= bs[i - 1] || bs[i] >= p) return 0;
    return 1;
}

static int bayesabc_block_sweep_impl(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                             const int64_t* block_starts, int64_t nblocks, const float* grams,
                             float* r, float* alpha, float* beta, float* delta,
                             float vare, const float* var_effects, const double* pi,
                             int nreps_arg, uint64_t seed, uint32_t iter, uint32_t marker0, int acc, int independent)
{
    if (!abc_args_ok(n, p, ld, vare) || !blocks_ok(block_starts, nblocks, p)) return -1;
    const float ie = 1.0f / vare;
    float* dall = independent ? (float*)calloc((size_t)(p > 0 ? p : 1), sizeof(float)) : NULL;
    const float* G = grams;
    for (int64_t bi = 0; bi < nblocks; ++bi) {                          /* BayesABC.jl:145 */
        const int64_t j0 = block_starts[bi];
        const int64_t b  = (bi + 1 < nblocks ? block_starts[bi + 1] : p) - j0;
        float* a_old_blk = (float*)malloc(sizeof(float) * (size_t)b);
        float* rhs_b     = (float*)malloc(sizeof(float) * (size_t)b);
        memcpy(a_old_blk, alpha + j0, sizeof(float) * (size_t)b);       /* :150 */
        for (int64_t k = 0; k < b; ++k) rhs_b[k] = dot_xr(X + (j0 + k) * ld, r, n, acc); /* :152 */
        const int nreps = nreps_arg > 0 ? nreps_arg : (int)b;           /* :153 */
        for (int rep = 0; rep < nreps; ++rep)
            for (int64_t k = 0; k < b; ++k) {                           /* :155-178 */
                const int64_t j = j0 + k;
                const uint32_t m = marker0 + (uint32_t)j;
                const double u = orc_uniform(seed, m, iter, (uint32_t)rep, 0);
                const double z = orc_normal(seed, m, iter, (uint32_t)rep, 0);
                const float a = abc_update(rhs_b[k], xpx[j], &alpha[j], &beta[j], &delta[j], ie,
                                           var_effects[j], pi[j], u, z);
                if (a != 0.0f) axpy_f32(a, G + k * b, rhs_b, b);        /* :169,172 (G symmetric) */
            }
        for (int64_t k = 0; k < b; ++k) {                               /* :181-185 */
            const float d = a_old_blk[k] - alpha[j0 + k];
            if (independent) dall[j0 + k] = d;                          /* block_deltas[i]  :247-248 */
            else if (d != 0.0f) axpy_f32(d, X + (j0 + k) * ld, r, n);
        }
        free(a_old_blk); free(rhs_b);
        G += b * b;
    }
    if (independent) {                                                  /* reconcile  :251-253 */
        for (int64_t j = 0; j < p; ++j) if (dall[j] != 0.0f) axpy_f32(dall[j], X + j * ld, r, n);
        free(dall);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* BayesR scalar kernel (BayesR.jl:56-96).  Returns axpy coefficient.                          */
/* ------------------------------------------------------------------------------------------ */
static inline float bayesr_update(float s, float d, float* alpha, int32_t* delta, float ie,
                                  float sigma_sq, const double* pi_j, const double* gamma,
                                  double u, double z)
{
    const float a_old = *alpha;
    const float rhs   = (s + d * a_old) * ie;                                    /* :60 */
    const float die   = d * ie;                                                  /* Float32 product in :70 */
    double lp[4], probs[4];
    lp[0] = log(pi_j[0]);                                                        /* :64 */
    for (int k = 1; k < 4; ++k) {                                                /* :65-72 */
        const double varEffect = gamma[k] * (double)sigma_sq;
        const double invVarEffect = 1.0 / varEffect;
        const double lhs = (double)die + invVarEffect;
        const double invLhs = 1.0 / lhs;
        const double betaHat = invLhs * (double)rhs;
        lp[k] = 0.5 * (log(invLhs) - log(varEffect) + betaHat * (double)rhs) + log(pi_j[k]);
    }
    double mx = lp[0];                                                           /* :1-4 */
    for (int k = 1; k < 4; ++k) if (lp[k] > mx) mx = lp[k];
    double se = 0.0;
    for (int k = 0; k < 4; ++k) se += exp(lp[k] - mx);
    const double log_norm = mx + log(se);
    for (int k = 0; k < 4; ++k) probs[k] = exp(lp[k] - log_norm);                /* :75-77 */

    /* rand(Categorical(probs)) (:79): Distributions.jl 0.25 (not vendored) walks the CDF while
     * cp <= draw; the reference's own replay harness restates it the same way up to ties
     * (benchmarks/bayesr_parity_replay_jwas.jl:37-41). */
    int cls = 0;
    double cp = probs[0];
    while (cp <= u && cls < 3) { ++cls; cp += probs[cls]; }
    *delta = cls + 1;                                                            /* :80 */

    if (cls == 0) {                                                              /* :82-86 */
        *alpha = 0.0f;
        return a_old;
    }
    const double varEffect = gamma[cls] * (double)sigma_sq;                      /* :88-94 */
    const double lhs = (double)die + 1.0 / varEffect;
    const double invLhs = 1.0 / lhs;
    const double betaHat = invLhs * (double)rhs;
    *alpha = (float)(betaHat + z * sqrt(invLhs));
    return a_old - *alpha;
}

static int bayesr_priors_ok(const double* pi, int pi_is_matrix, int64_t p)
{
    if (pi_is_matrix) return 1;                      /* BayesR.jl:16-20: shape checks only */
    (void)p;
    double s = 0.0;
    for (int k = 0; k < 4; ++k) { if (pi[k] < 0.0) return 0; s += pi[k]; }
    return fabs(s - 1.0) <= 1e-8;                    /* BayesR.jl:9-14 */
}

int orc_bayesr_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                     float* r, float* alpha, int32_t* delta,
                     float vare, float sigma_sq, const double* pi, int pi_is_matrix,
                     const double* gamma, uint64_t seed, uint32_t iter, uint32_t marker0, int acc)
{
    if (!abc_args_ok(n, p, ld, vare) || !bayesr_priors_ok(pi, pi_is_matrix, p)) return -1;
    if (!(sigma_sq > 0.0f)) return -2;
    const float ie = 1.0f / vare;
    for (int64_t j = 0; j < p; ++j) {
        const float* x = X + j * ld;
        const uint32_t m = marker0 + (uint32_t)j;
        const float s = dot_xr(x, r, n, acc);
        const double u = orc_uniform(seed, m, iter, 0, 0);
        const double z = orc_normal(seed, m, iter, 0, 0);
        const float a = bayesr_update(s, xpx[j], &alpha[j], &delta[j], ie, sigma_sq,
                                      pi_is_matrix ? pi + 4 * j : pi, gamma, u, z);
        if (a != 0.0f) axpy_f32(a, x, r, n);
    }
    return 0;
}

static int bayesr_block_sweep_impl(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                           const int64_t* block_starts, int64_t nblocks, const float* grams,
                           float* r, float* alpha, int32_t* delta,
                           float vare, float sigma_sq, const double* pi, int pi_is_matrix,
                           const double* gamma, int nreps_arg,
                           uint64_t seed, uint32_t iter, uint32_t marker0, int acc, int independent)
{
    if (!abc_args_ok(n, p, ld, vare) || !bayesr_priors_ok(pi, pi_is_matrix, p) ||
        !blocks_ok(block_starts, nblocks, p)) return -1;
    if (!(sigma_sq > 0.0f)) return -2;
    const float ie = 1.0f / vare;
    float* dall = independent ? (float*)calloc((size_t)(p > 0 ? p : 1), sizeof(float)) : NULL;
    const float* G = grams;
    for (int64_t bi = 0; bi < nblocks; ++bi) {                          /* BayesR.jl:139-192 */
        const int64_t j0 = block_starts[bi];
        const int64_t b  = (bi + 1 < nblocks ? block_starts[bi + 1] : p) - j0;
        float* a_old_blk = (float*)malloc(sizeof(float) * (size_t)b);
        float* rhs_b     = (float*)malloc(sizeof(float) * (size_t)b);
        memcpy(a_old_blk, alpha + j0, sizeof(float) * (size_t)b);
        for (int64_t k = 0; k < b; ++k) rhs_b[k] = dot_xr(X + (j0 + k) * ld, r, n, acc);
        const int nreps = nreps_arg > 0 ? nreps_arg : (int)b;
        for (int rep = 0; rep < nreps; ++rep)
            for (int64_t k = 0; k < b; ++k) {
                const int64_t j = j0 + k;
                const uint32_t m = marker0 + (uint32_t)j;
                const double u = orc_uniform(seed, m, iter, (uint32_t)rep, 0);
                const double z = orc_normal(seed, m, iter, (uint32_t)rep, 0);
                const float a = bayesr_update(rhs_b[k], xpx[j], &alpha[j], &delta[j], ie, sigma_sq,
                                              pi_is_matrix ? pi + 4 * j : pi, gamma, u, z);
                if (a != 0.0f) axpy_f32(a, G + k * b, rhs_b, b);        /* :182 */
            }
        for (int64_t k = 0; k < b; ++k) {                               /* :186-190 */
            const float d = a_old_blk[k] - alpha[j0 + k];
            if (independent) dall[j0 + k] = d;                          /* BayesR.jl:264-265 */
            else if (d != 0.0f) axpy_f32(d, X + (j0 + k) * ld, r, n);
        }
        free(a_old_blk); free(rhs_b);
        G += b * b;
    }
    if (independent) {                                                  /* BayesR.jl:269-271 */
        for (int64_t j = 0; j < p; ++j) if (dall[j] != 0.0f) axpy_f32(dall[j], X + j * ld, r, n);
        free(dall);
    }
    return 0;
}

int orc_bayesr_block_nreps(int64_t iter, int64_t burnin, int64_t block_size)
{
    if (block_size < 1) return -1;                                       /* BayesR.jl:23 */
    return iter <= burnin ? 1 : (int)block_size;                         /* :24 */
}

void orc_bayesr_sigma_suffstats(const float* alpha, const int32_t* delta, int64_t p,
                                const double* gamma, double* ssq, int64_t* nnz)
{
    /* variance_components.jl:68-79.  The reference accumulates in eltype(alpha) with
     * alpha[j]^2/gamma[dj] promoted by the Float64 gamma: a Float64 running sum. */
    double s = 0.0; int64_t c = 0;
    for (int64_t j = 0; j < p; ++j)
        if (delta[j] > 1) { s += ((double)alpha[j] * (double)alpha[j]) / gamma[delta[j] - 1]; ++c; }
    *ssq = s; *nnz = c;
}

/* ------------------------------------------------------------------------------------------ */
/* multi-trait BayesC, sampler I (MTBayesABC.jl:57-127)                                        */
/* ------------------------------------------------------------------------------------------ */
#define ORC_MAXT 8

/* t x t inverse in double (Gauss-Jordan with partial pivoting), rounded to float -- stands in
 * for Julia's inv(::Matrix{Float32}) (LAPACK sgetri; MTBayesABC.jl:66-67). */
static int inv_small(const float* A, int t, float* Ainv)
{
    double M[ORC_MAXT][2 * ORC_MAXT];
    for (int i = 0; i < t; ++i) {
        for (int j = 0; j < t; ++j) { M[i][j] = A[i * t + j]; M[i][t + j] = (i == j); }
    }
    for (int c = 0; c < t; ++c) {
        int piv = c;
        for (int i = c + 1; i < t; ++i) if (fabs(M[i][c]) > fabs(M[piv][c])) piv = i;
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

/* Multi-trait BayesA/B (MTBayesABC.jl:66  Ginv = inv.(varEffects),  :86-90  Ginv[marker]): one t x t effect covariance per
 * marker.  Test-harness state: when set (p x t x t, row-major per marker), every multi-trait sweep inverts marker j's own
 * matrix instead of using the sweep-wide var_effect. */
static const float* g_var_effect_mat = NULL;
void orc_set_var_effect_matrix(const float* mat) { g_var_effect_mat = mat; }

/* ------------------------------------------------------------------------------------------ */
/* Multi-trait BayesA/B: one InverseWishart(df, scale + b_j b_j') draw per marker                 */
/* (sample_variance(data, 1, df, scale) per marker, variance_components.jl:181-186; the caller's  */
/* df is the reference's df + 1).  Bartlett's decomposition on the counter RNG, operation for     */
/* operation the device's k_sample_marker_covariances (csrc/sweep.hpp):                            */
/*   S = scale + b b' = C C';  A lower-triangular, A_ii = sqrt(chi2(df - i)), A_ik ~ N(0,1);       */
/*   K' = A^-1 C' (forward substitution);  G = K K', symmetrised, rounded to float.                */
/* Counter of a draw: (global marker, iteration, 0x80000000 | attempt, slot).                      */
/* ------------------------------------------------------------------------------------------ */
static void chol_lower(int t, const double* A, double* L);
static double iw_chi2(uint64_t seed, uint32_t marker, uint32_t iter, uint32_t slot, double nu)
{
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) }, w[4], w2[4];
    double a = 0.5 * nu, boost = 1.0;
    if (a < 1.0) {
        uint32_t ctr[4] = { marker, iter, 0x80000000u | 0xFFFFu, slot };
        orc_philox4x32_10(ctr, key, w);
        boost = exp(log(u52(w[0], w[1])) / a);
        a = a + 1.0;
    }
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    double g = d;
    for (uint32_t attempt = 0; attempt < 64u; ++attempt) {
        uint32_t ctr[4] = { marker, iter, 0x80000000u | attempt, slot }, ctr2[4] = { marker, iter, 0x80000000u | attempt, slot + 1u };
        orc_philox4x32_10(ctr, key, w);
        orc_philox4x32_10(ctr2, key, w2);
        const double x = sqrt(-2.0 * log(u52(w[0], w[1]))) * cos(6.283185307179586476925286766559 * u52(w[2], w[3]));
        const double u = u52(w2[0], w2[1]);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        g = d * v;
        if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) break;
    }
    return 2.0 * g * boost;
}

void orc_sample_marker_covariances(int t, int64_t p, const float* beta /* [t][p] */, double df, const double* scale /* t x t */,
                                   uint64_t seed, uint32_t iter, uint32_t marker0, float* var_mat /* [p][t][t] */)
{
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) }, w[4];
    for (int64_t j = 0; j < p; ++j) {
        const uint32_t marker = marker0 + (uint32_t)j;
        double b[ORC_MAXT], S[ORC_MAXT * ORC_MAXT], C[ORC_MAXT * ORC_MAXT], A[ORC_MAXT * ORC_MAXT], Kt[ORC_MAXT * ORC_MAXT], G[ORC_MAXT * ORC_MAXT];
        for (int a = 0; a < t; ++a) b[a] = (double)beta[(int64_t)a * p + j];
        for (int a = 0; a < t; ++a)
            for (int c = 0; c < t; ++c) { S[a * t + c] = scale[a * t + c] + b[a] * b[c]; A[a * t + c] = 0.0; }
        chol_lower(t, S, C);
        for (int i = 0; i < t; ++i) {
            A[i * t + i] = sqrt(iw_chi2(seed, marker, iter, 32u + 2u * (uint32_t)i, df - (double)i));
            for (int k = 0; k < i; ++k) {
                uint32_t ctr[4] = { marker, iter, 0x80000000u, 64u + 4u * (uint32_t)i + (uint32_t)k };
                orc_philox4x32_10(ctr, key, w);
                A[i * t + k] = sqrt(-2.0 * log(u52(w[0], w[1]))) * cos(6.283185307179586476925286766559 * u52(w[2], w[3]));
            }
        }
        for (int i = 0; i < t; ++i)
            for (int c = 0; c < t; ++c) {
                double acc = C[c * t + i];                                        /* C'[i][c] */
                for (int k = 0; k < i; ++k) acc = acc - A[i * t + k] * Kt[k * t + c];
                Kt[i * t + c] = acc / A[i * t + i];
            }
        for (int a = 0; a < t; ++a)
            for (int c = 0; c < t; ++c) {
                double s = 0.0;
                for (int i = 0; i < t; ++i) s = s + Kt[i * t + a] * Kt[i * t + c];
                G[a * t + c] = s;
            }
        for (int a = 0; a < t; ++a)
            for (int c = 0; c < t; ++c) var_mat[(j * t + a) * t + c] = (float)(0.5 * (G[a * t + c] + G[c * t + a]));
    }
}

/* constraint = true (variance_components.jl:112-117 through :184: sample_variance(data, 1, df, scale, false, true)): only the
 * diagonal is drawn, G_kk = (scale_kk + b_jk^2) / chi2(df), scale = the reference's df * scale and df = its df + 1; the
 * chi-square of trait k on the same counters as row k of the full draw (slot 32 + 2k); off-diagonals are zero. */
void orc_sample_marker_variances_diag(int t, int64_t p, const float* beta /* [t][p] */, double df, const double* scale /* t x t */,
                                      uint64_t seed, uint32_t iter, uint32_t marker0, float* var_mat /* [p][t][t] */)
{
    for (int64_t j = 0; j < p; ++j) {
        const uint32_t marker = marker0 + (uint32_t)j;
        for (int a = 0; a < t; ++a) {
            const double b = (double)beta[(int64_t)a * p + j];
            const double g = (scale[a * t + a] + b * b) / iw_chi2(seed, marker, iter, 32u + 2u * (uint32_t)a, df);
            for (int c = 0; c < t; ++c) var_mat[(j * t + a) * t + c] = (c == a) ? (float)g : 0.0f;
        }
    }
}

/* megaBayesABC! with BayesA/B (BayesABC.jl:1-8: [vari[i,i] for vari in locus_effect_variances]): the marker's own diagonal */
static inline const float* marker_var(int64_t j, int t, const float* var_all)
{
    return g_var_effect_mat ? g_var_effect_mat + j * t * t : var_all;
}

static inline const float* marker_ginv(int64_t j, int t, const float* Ginv_all, float* tmp)
{
    if (!g_var_effect_mat) return Ginv_all;
    if (inv_small(g_var_effect_mat + j * t * t, t, tmp)) for (int i = 0; i < t * t; ++i) tmp[i] = NAN;
    return tmp;
}

/* Rule L (the device's sampler-I definition, csrc/sweep.hpp): a marker that enters with every delta = 1 and -- by the exact
 * evaluation below, the reference's operation order -- leaves with every delta = 1 takes its new effects from the LINEAR
 * FORM of its t conditionals,  beta = A w + c  (A, c from the marker's constants, its old beta and its draws by the double
 * recurrence below, rounded to float; beta_k = fmaf(A[k][t-1], w[t-1], ... fmaf(A[k][0], w[0], c[k]))): the same
 * conditional means and draws in another association, a few ulp from the literal order.  orc_set_mt_linear_form(0)
 * switches the rule off (the literal restatement; tests/test_oracle_kat.py compares the two). */
static int g_mt_linear = 1;
void orc_set_mt_linear_form(int on) { g_mt_linear = on; }

static inline void mt1_linear_coeffs(int t, float d, const float* Rinv, const float* Ginv, const float* b_old, const double* z,
                                     float* Af /* t x t */, float* cf /* t */)
{
    double Ad[ORC_MAXT][ORC_MAXT], cd[ORC_MAXT];
    for (int k = 0; k < t; ++k) {
        const float C11 = Ginv[k * t + k] + Rinv[k * t + k] * d;                 /* :89 */
        const float invLhs1 = 1.0f / C11;                                        /* :95 */
        const double il = (double)invLhs1;
        double C12[ORC_MAXT];
        for (int m = 0; m < t; ++m) C12[m] = (double)(Ginv[k * t + m] + (d * 1.0f) * Rinv[k * t + m]);      /* :90, delta_m = 1 */
        for (int m = 0; m < t; ++m) {
            double acc = (double)Rinv[m * t + k];
            for (int j = 0; j < k; ++j) acc = acc - C12[j] * Ad[j][m];
            Ad[k][m] = il * acc;
        }
        double acc = 0.0;
        for (int j = 0; j < k; ++j) acc = acc - C12[j] * cd[j];
        for (int j = k + 1; j < t; ++j) acc = acc - C12[j] * (double)b_old[j];
        cd[k] = il * acc + z[k] * (double)sqrtf(invLhs1);
    }
    for (int k = 0; k < t; ++k) {
        cf[k] = (float)cd[k];
        for (int m = 0; m < t; ++m) Af[k * t + m] = (float)Ad[k][m];
    }
}
static inline void mt1_linear_beta(int t, const float* Af, const float* cf, const float* w, float* b_new)
{
    for (int k = 0; k < t; ++k) {
        float v = cf[k];
        for (int m = 0; m < t; ++m) v = fmaf(Af[k * t + m], w[m], v);
        b_new[k] = v;
    }
}
static inline void mt1_linear(int t, const float* w, float d, const float* Rinv, const float* Ginv, const float* b_old,
                              const double* z, float* b_new)
{
    float Af[ORC_MAXT * ORC_MAXT], cf[ORC_MAXT];
    mt1_linear_coeffs(t, d, Rinv, Ginv, b_old, z, Af, cf);
    mt1_linear_beta(t, Af, cf, w, b_new);
}

/* One marker.  w[k] = x'r_k + d*alpha_old_k already formed.  Writes axpy coefficients a[k]. */
static inline void mt1_update(int t, const float* w, float d, float* alpha, float* beta, float* delta,
                              int64_t stride, const float* Rinv, const float* Ginv,
                              const double* log_prior, uint64_t seed, uint32_t marker, uint32_t iter,
                              uint32_t rep, float* a_out)
{
    float b[ORC_MAXT], dl[ORC_MAXT], b_in[ORC_MAXT] = {0}, a_in[ORC_MAXT] = {0};
    double zz[ORC_MAXT] = {0};
    int all1 = g_mt_linear;
    for (int k = 0; k < t; ++k) { b[k] = beta[k * stride]; dl[k] = delta[k * stride]; b_in[k] = b[k]; a_in[k] = alpha[k * stride]; all1 = all1 && (dl[k] == 1.0f); }
    for (int k = 0; k < t; ++k) {                                                /* :85 */
        const float a_old = alpha[k * stride];
        const float Ginv11 = Ginv[k * t + k];                                    /* :86 */
        const float C11 = Ginv11 + Rinv[k * t + k] * d;                          /* :89 */
        float rhs0 = 0.0f, c12b = 0.0f, wR = 0.0f;
        for (int m = 0; m < t; ++m) {
            wR = wR + w[m] * Rinv[m * t + k];                                    /* w'*Rinv[:,k]  :96 */
            if (m == k) continue;
            const float C12m = Ginv[k * t + m] + (d * dl[m]) * Rinv[k * t + m];  /* :90 */
            rhs0 = rhs0 + Ginv[k * t + m] * b[m];                                /* :93 */
            c12b = c12b + C12m * b[m];                                           /* :96 */
        }
        rhs0 = -rhs0;
        const float invLhs0 = 1.0f / Ginv11;                                     /* :92 */
        const float gHat0 = rhs0 * invLhs0;                                      /* :94 */
        const float invLhs1 = 1.0f / C11;                                        /* :95 */
        const float rhs1 = wR - c12b;                                            /* :96 */
        const float gHat1 = rhs1 * invLhs1;                                      /* :97 */
        unsigned s0 = 0, s1 = 0;
        for (int m = 0; m < t; ++m) {
            const unsigned bit = (m == k) ? 0u : (dl[m] != 0.0f ? 1u : 0u);
            s0 |= bit << m; s1 |= bit << m;
        }
        s1 |= 1u << k;
        const float in0 = logf_via_double(Ginv11) - (gHat0 * gHat0) * Ginv11;    /* Float32 part :104 */
        const float in1 = logf_via_double(C11) - (gHat1 * gHat1) * C11;          /* Float32 part :105 */
        const double logDelta0 = -0.5 * (double)in0 + log_prior[s0];
        const double logDelta1 = -0.5 * (double)in1 + log_prior[s1];
        const double probDelta1 = 1.0 / (1.0 + exp(logDelta0 - logDelta1));      /* :107 */
        const double u = orc_uniform(seed, marker, iter, rep, (uint32_t)k);
        const double z = orc_normal(seed, marker, iter, rep, (uint32_t)k);
        zz[k] = z;
        if (u < probDelta1) {                                                    /* :108-111 */
            dl[k] = 1.0f;
            b[k] = (float)((double)gHat1 + z * (double)sqrtf(invLhs1));
            alpha[k * stride] = b[k];
            a_out[k] = a_old - b[k];
        } else {                                                                 /* :112-119 */
            b[k] = (float)((double)gHat0 + z * (double)sqrtf(invLhs0));
            dl[k] = 0.0f;
            alpha[k * stride] = 0.0f;
            a_out[k] = a_old;
        }
    }
    for (int k = 0; k < t; ++k) all1 = all1 && (dl[k] == 1.0f);
    if (all1) {                                                                  /* Rule L */
        mt1_linear(t, w, d, Rinv, Ginv, b_in, zz, b);
        for (int k = 0; k < t; ++k) { alpha[k * stride] = b[k]; a_out[k] = a_in[k] - b[k]; }
    }
    for (int k = 0; k < t; ++k) { beta[k * stride] = b[k]; delta[k * stride] = dl[k]; }
}

/* ------------------------------------------------------------------------------------------ */
/* multi-trait BayesC, sampler II: joint state (MTBayesABC.jl:129-210).                          */
/* States are indexed by bitmask s (bit k = trait k in the model); the reference iterates        */
/* collect(keys(pi)) (Dict order, unspecified) -- the label order only decides which uniform      */
/* interval maps to which state.  Temporaries are Float64 as in the reference (:153-157).         */
/* The t x t algebra (inv, cholesky, det: LAPACK in the reference) is restated with one fixed     */
/* operation order shared with the device: Cholesky lhs = L L', M = L^-1, inv = M'M,              */
/* det = prod L_ii^2, chol(inv) lower.                                                           */
/* ------------------------------------------------------------------------------------------ */
static void chol_lower(int t, const double* A, double* L)
{
    for (int i = 0; i < t * t; ++i) L[i] = 0.0;
    for (int j = 0; j < t; ++j) {
        double s = A[j * t + j];
        for (int k = 0; k < j; ++k) s = s - L[j * t + k] * L[j * t + k];
        L[j * t + j] = sqrt(s);
        for (int i = j + 1; i < t; ++i) {
            double v = A[i * t + j];
            for (int k = 0; k < j; ++k) v = v - L[i * t + k] * L[j * t + k];
            L[i * t + j] = v / L[j * t + j];
        }
    }
}

static void mt2_state(int t, unsigned st, const float* w, float d, const float* Rinv, const float* Ginv,
                      const double* z, double* logdet_quad /* out: -0.5*(log det lhs - rhs'gHat) */, double* cand)
{
    double lhs[ORC_MAXT * ORC_MAXT], L[ORC_MAXT * ORC_MAXT], M[ORC_MAXT * ORC_MAXT], inv[ORC_MAXT * ORC_MAXT];
    double C[ORC_MAXT * ORC_MAXT], rhs[ORC_MAXT], gHat[ORC_MAXT];
    for (int a = 0; a < t; ++a)
        for (int c = 0; c < t; ++c) {
            const double Da = (st >> a) & 1u ? 1.0 : 0.0, Dc = (st >> c) & 1u ? 1.0 : 0.0;
            const double rl = (Da * (double)Rinv[a * t + c]) * Dc;                  /* D*Rinv*D       :159 */
            lhs[a * t + c] = rl * (double)d + (double)Ginv[a * t + c];             /* :179 */
        }
    for (int a = 0; a < t; ++a) {                                                  /* (Rinv*D)'w     :180 */
        const double Da = (st >> a) & 1u ? 1.0 : 0.0;
        double s = 0.0;
        for (int m = 0; m < t; ++m) s = s + ((double)Rinv[m * t + a] * Da) * (double)w[m];
        rhs[a] = s;
    }
    chol_lower(t, lhs, L);
    for (int i = 0; i < t * t; ++i) M[i] = 0.0;
    for (int j = 0; j < t; ++j) {                                                  /* M = L^-1 */
        M[j * t + j] = 1.0 / L[j * t + j];
        for (int i = j + 1; i < t; ++i) {
            double s = 0.0;
            for (int k = j; k < i; ++k) s = s + L[i * t + k] * M[k * t + j];
            M[i * t + j] = -s / L[i * t + i];
        }
    }
    for (int a = 0; a < t; ++a)                                                    /* inv(lhs) = M'M :181 */
        for (int c = 0; c < t; ++c) {
            double s = 0.0;
            for (int k = (a > c ? a : c); k < t; ++k) s = s + M[k * t + a] * M[k * t + c];
            inv[a * t + c] = s;
        }
    double det = 1.0;
    for (int j = 0; j < t; ++j) det = det * (L[j * t + j] * L[j * t + j]);
    double quad = 0.0;
    for (int a = 0; a < t; ++a) {                                                  /* gHat = invLhs*rhs :183 */
        double s = 0.0;
        for (int c = 0; c < t; ++c) s = s + inv[a * t + c] * rhs[c];
        gHat[a] = s;
        quad = quad + rhs[a] * s;
    }
    *logdet_quad = -0.5 * (log(det) - quad);                                       /* :184 */
    chol_lower(t, inv, C);                                                         /* cholesky(Hermitian(invLhs)).L :182 */
    for (int a = 0; a < t; ++a) {                                                  /* gHat + L*z     :185 */
        double s = gHat[a];
        for (int c = 0; c <= a; ++c) s = s + C[a * t + c] * z[c];
        cand[a] = s;
    }
}

static inline void mt2_update(int t, const float* w, float d, float* alpha, float* beta, float* delta,
                              int64_t stride, const float* Rinv, const float* Ginv,
                              const double* log_prior, uint64_t seed, uint32_t marker, uint32_t iter,
                              uint32_t rep, float* a_out)
{
    const int ns = 1 << t;
    double z[ORC_MAXT], logDelta[1 << ORC_MAXT], cand[(1 << ORC_MAXT) * ORC_MAXT];
    for (int k = 0; k < t; ++k) z[k] = orc_normal(seed, marker, iter, rep, (uint32_t)k);   /* randn(ntraits) :176 */
    const double u = orc_uniform(seed, marker, iter, rep, 0);
    double mx = -INFINITY;
    for (int s = 0; s < ns; ++s) {
        double q;
        mt2_state(t, (unsigned)s, w, d, Rinv, Ginv, z, &q, cand + s * t);
        logDelta[s] = q + log_prior[s];
        if (logDelta[s] > mx) mx = logDelta[s];
    }
    double den = 0.0;                                                              /* :188-196 */
    for (int s = 0; s < ns; ++s) { logDelta[s] = exp(logDelta[s] - mx); den += logDelta[s]; }
    int which = ns - 1;                                                            /* rand(Categorical(probDelta)) :198 */
    double cp = 0.0;
    for (int s = 0; s < ns; ++s) { cp += logDelta[s] / den; if (u < cp) { which = s; break; } }
    for (int k = 0; k < t; ++k) {
        const double dk = (which >> k) & 1 ? 1.0 : 0.0;
        const double b_new = cand[which * t + k];
        const double a_new = dk * b_new;                                           /* diagm(delta)*beta :201 */
        const float a_old = alpha[k * stride];
        a_out[k] = (float)((double)a_old - a_new);                                 /* oldα-newα (Float64) -> axpy :204 */
        beta[k * stride]  = (float)b_new;
        delta[k * stride] = (float)dk;
        alpha[k * stride] = (float)a_new;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* megaBayesABC! (BayesABC.jl:1-8): G.constraint == true -> t independent single-trait BayesC   */
/* sweeps with vare[k,k], var_effect[k,k] and a per-trait pi (passed in prior[k]); draws of     */
/* trait k use slot k.                                                                          */
/* ------------------------------------------------------------------------------------------ */
static inline void mega_update(int t, const float* w_minus /* rhs_b (without d*alpha) */, float d, float* alpha,
                               float* beta, float* delta, int64_t stride, const float* vare, const float* var_effect,
                               const double* pi, uint64_t seed, uint32_t marker, uint32_t iter, uint32_t rep,
                               float* a_out)
{
    for (int k = 0; k < t; ++k) {
        const double u = orc_uniform(seed, marker, iter, rep, (uint32_t)k);
        const double z = orc_normal(seed, marker, iter, rep, (uint32_t)k);
        const float ie = 1.0f / vare[k * t + k];
        a_out[k] = abc_update(w_minus[k], d, &alpha[k * stride], &beta[k * stride], &delta[k * stride], ie,
                              var_effect[k * t + k], pi[k], u, z);
    }
}

enum { MT_SAMPLER_I = 1, MT_SAMPLER_II = 2, MT_MEGA = 3 };

/* s[k] = x'r_k (or the block rhs entry); dispatch on the multi-trait sampler kind */
static inline void mt_update(int kind, int t, const float* s, float d, float* alpha, float* beta, float* delta,
                             int64_t stride, const float* Rinv, const float* Ginv, const float* vare,
                             const float* var_effect, const double* prior, uint64_t seed, uint32_t marker,
                             uint32_t iter, uint32_t rep, float* a_out)
{
    if (kind == MT_MEGA) {
        mega_update(t, s, d, alpha, beta, delta, stride, vare, var_effect, prior, seed, marker, iter, rep, a_out);
        return;
    }
    float w[ORC_MAXT];
    for (int k = 0; k < t; ++k) w[k] = s[k] + d * alpha[k * stride];              /* MTBayesABC.jl:82,172 */
    if (kind == MT_SAMPLER_I) mt1_update(t, w, d, alpha, beta, delta, stride, Rinv, Ginv, prior, seed, marker, iter, rep, a_out);
    else                      mt2_update(t, w, d, alpha, beta, delta, stride, Rinv, Ginv, prior, seed, marker, iter, rep, a_out);
}

int orc_mt_sweep(int kind, const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                         int t, float* r, int64_t ld_r, float* alpha, float* beta, float* delta,
                         const float* vare, const float* var_effect,
                         const double* log_prior, int prior_is_matrix,
                         uint64_t seed, uint32_t iter, uint32_t marker0, int acc)
{
    if (t < 1 || t > ORC_MAXT || n <= 0 || ld < n || ld_r < n) return -1;
    float Rinv[ORC_MAXT * ORC_MAXT], Ginv[ORC_MAXT * ORC_MAXT], Gtmp_[ORC_MAXT * ORC_MAXT];
    (void)Gtmp_;
    if (kind < MT_SAMPLER_I || kind > MT_MEGA) return -1;
    if (inv_small(vare, t, Rinv) || inv_small(var_effect, t, Ginv)) return -2;
    const int nstates = 1 << t;
    for (int64_t j = 0; j < p; ++j) {
        const float* x = X + j * ld;
        float w[ORC_MAXT], a[ORC_MAXT];
        for (int k = 0; k < t; ++k) w[k] = dot_xr(x, r + k * ld_r, n, acc);     /* :82 */
        mt_update(kind, t, w, xpx[j], alpha + j, beta + j, delta + j, p, Rinv, marker_ginv(j, t, Ginv, Gtmp_), vare, marker_var(j, t, var_effect),
                  prior_is_matrix ? log_prior + (int64_t)nstates * j : log_prior,
                  seed, marker0 + (uint32_t)j, iter, 0, a);
        for (int k = 0; k < t; ++k) if (a[k] != 0.0f) axpy_f32(a[k], x, r + k * ld_r, n);
    }
    return 0;
}

static int mt_block_sweep_impl(int kind, const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                               const int64_t* block_starts, int64_t nblocks, const float* grams,
                               int t, float* r, int64_t ld_r, float* alpha, float* beta, float* delta,
                               const float* vare, const float* var_effect,
                               const double* log_prior, int prior_is_matrix, int nreps_arg,
                               uint64_t seed, uint32_t iter, uint32_t marker0, int acc, int independent)
{
    if (t < 1 || t > ORC_MAXT || n <= 0 || ld < n || ld_r < n || !blocks_ok(block_starts, nblocks, p)) return -1;
    float Rinv[ORC_MAXT * ORC_MAXT], Ginv[ORC_MAXT * ORC_MAXT], Gtmp_[ORC_MAXT * ORC_MAXT];
    (void)Gtmp_;
    if (kind < MT_SAMPLER_I || kind > MT_MEGA) return -1;
    if (inv_small(vare, t, Rinv) || inv_small(var_effect, t, Ginv)) return -2;
    const int nstates = 1 << t;
    float* dall = independent ? (float*)calloc((size_t)((p > 0 ? p : 1) * t), sizeof(float)) : NULL;
    const float* G = grams;
    for (int64_t bi = 0; bi < nblocks; ++bi) {                          /* MTBayesABC.jl:243-333 */
        const int64_t j0 = block_starts[bi];
        const int64_t b  = (bi + 1 < nblocks ? block_starts[bi + 1] : p) - j0;
        float* a_old_blk = (float*)malloc(sizeof(float) * (size_t)(b * t));
        float* rhs_b     = (float*)malloc(sizeof(float) * (size_t)(b * t));
        for (int k = 0; k < t; ++k) {
            memcpy(a_old_blk + k * b, alpha + k * p + j0, sizeof(float) * (size_t)b);
            for (int64_t c = 0; c < b; ++c)
                rhs_b[k * b + c] = dot_xr(X + (j0 + c) * ld, r + k * ld_r, n, acc);
        }
        const int nreps = nreps_arg > 0 ? nreps_arg : (int)b;
        for (int rep = 0; rep < nreps; ++rep)
            for (int64_t c = 0; c < b; ++c) {
                const int64_t j = j0 + c;
                float w[ORC_MAXT], a[ORC_MAXT];
                for (int k = 0; k < t; ++k) w[k] = rhs_b[k * b + c];
                mt_update(kind, t, w, xpx[j], alpha + j, beta + j, delta + j, p, Rinv, marker_ginv(j, t, Ginv, Gtmp_), vare, marker_var(j, t, var_effect),
                          prior_is_matrix ? log_prior + (int64_t)nstates * j : log_prior,
                          seed, marker0 + (uint32_t)j, iter, (uint32_t)rep, a);
                for (int k = 0; k < t; ++k)
                    if (a[k] != 0.0f) axpy_f32(a[k], G + c * b, rhs_b + k * b, b);   /* :311,317 */
            }
        for (int k = 0; k < t; ++k)
            for (int64_t c = 0; c < b; ++c) {                                        /* :329 */
                const float d = a_old_blk[k * b + c] - alpha[k * p + j0 + c];
                if (independent) dall[k * p + j0 + c] = d;                           /* :425-428 */
                else if (d != 0.0f) axpy_f32(d, X + (j0 + c) * ld, r + k * ld_r, n);
            }
        free(a_old_blk); free(rhs_b);
        G += b * b;
    }
    if (independent) {                                                               /* :431-435 */
        for (int64_t j = 0; j < p; ++j)
            for (int k = 0; k < t; ++k)
                if (dall[k * p + j] != 0.0f) axpy_f32(dall[k * p + j], X + j * ld, r + k * ld_r, n);
        free(dall);
    }
    return 0;
}


/* ------------------------------------------------------------------------------------------ */
/* One-block LOOKAHEAD form of the exact block chain (the schedule the HIP path runs).          */
/*                                                                                              */
/* Algebra (exact arithmetic): with r(b-2) the residual that holds the exit updates of blocks   */
/* <= b-2,   X_b' r(b-1) = X_b' r(b-2) + (X_b' X_{b-1}) d_{b-1},   d = alpha_old - alpha_new.    */
/* So the block RHS can be formed from the STALE residual r(b-2) (which lets the device stream   */
/* block b while block b-1 is still being sampled) and corrected with the cross-Gram columns of  */
/* the markers of block b-1 that changed.  Everything else is BayesABC_block! (BayesABC.jl:     */
/* 145-187).  Rounding differs from the plain block form only in how rhs_b is assembled:         */
/*   s[c]    = fl32( sum_i x_ic * r(b-2)_i )          (fp64 accumulation, rounded once)          */
/*   corr[c] = fmaf(d_j, fl32(x_j'x_c), corr[c])      from 0, over the changed markers j of      */
/*                                                    block b-1 in marker order                  */
/*   rhs[c]  = s[c] + corr[c]                         (one fp32 add)                             */
/* then r(b-1) = r(b-2) + X_{b-1} d_{b-1} by per-marker fmaf, as before.                         */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int64_t n, ld; const float* X; int acc; } la_ctx;

/* rhs (t x b) for block [j0, j0+b) from the stale residuals, corrected by the previous block's
 * net changes dprev (t x bprev, 0 where unchanged), then the previous exit update is applied. */
static void la_block_rhs(const la_ctx* L, int t, float* r, int64_t ld_r, int64_t j0, int64_t b,
                         int64_t jprev, int64_t bprev, const float* dprev, float* rhs)
{
    for (int k = 0; k < t; ++k)
        for (int64_t c = 0; c < b; ++c)
            rhs[k * b + c] = dot_xr(L->X + (j0 + c) * L->ld, r + k * ld_r, L->n, L->acc);
    /* corr[c] = sum over the changed markers of the previous block, accumulated from 0 in marker order with
     * fmaf; then ONE fp32 add onto the rounded dot product (the device forms corr at the end of the previous
     * block's sampler, before this block's partial sums exist). */
    float* corr = (float*)calloc((size_t)(b * t), sizeof(float));
    for (int64_t e = 0; e < bprev; ++e) {
        int any = 0;
        for (int k = 0; k < t; ++k) any |= (dprev[k * bprev + e] != 0.0f);
        if (!any) continue;
        const float* xe = L->X + (jprev + e) * L->ld;
        for (int64_t c = 0; c < b; ++c) {
            const float g = dot_acc(xe, L->X + (j0 + c) * L->ld, L->n, L->acc);      /* cross-Gram entry */
            for (int k = 0; k < t; ++k) corr[k * b + c] = fmaf(dprev[k * bprev + e], g, corr[k * b + c]);
        }
    }
    for (int64_t i = 0; i < b * t; ++i) rhs[i] = rhs[i] + corr[i];
    free(corr);
    for (int64_t e = 0; e < bprev; ++e)
        for (int k = 0; k < t; ++k)
            if (dprev[k * bprev + e] != 0.0f) axpy_f32(dprev[k * bprev + e], L->X + (jprev + e) * L->ld, r + k * ld_r, L->n);
}

/* ------------------------------------------------------------------------------------------ */
/* GROUPED LOOKAHEAD (the device's grouped launches: jwas_sweep_params.group_launch, csrc/sweep.hpp  */
/* k_group_step).  m = 2 or 4 consecutive blocks form a group; the partial sums of ALL blocks of     */
/* group g come from the residual that holds the exit updates of the groups <= g-2, and block s of    */
/* the group is corrected for                                                                        */
/*   cG  the changes of the whole group g-1,                                                          */
/*   cW  (s odd) the changes of block s-1, the first block of its pair,                                */
/*   cP  (m = 4, s >= 2) the changes of blocks 0 and 1 of its own group,                               */
/* each  c[col] = fmaf(d_j, fl32(x_j'x_col), c[col])  from 0 over the changed markers j in marker       */
/* order;  rhs = fl32(x'r) + ((cW + cG) + cP)  (absent terms +0).  Exact arithmetic: the block chain    */
/* of BayesABC.jl:145-187 / BayesR.jl:111-193.                                                          */
/* ------------------------------------------------------------------------------------------ */
static int g_la_group = 1;
void orc_set_lookahead_group(int m) { g_la_group = (m == 2 || m == 4) ? m : 1; }

typedef struct { int64_t* j; float* d; int64_t n, cap; } la_list;
static void la_list_push(la_list* l, int64_t j, float d)
{
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 64; l->j = (int64_t*)realloc(l->j, sizeof(int64_t) * (size_t)l->cap); l->d = (float*)realloc(l->d, sizeof(float) * (size_t)l->cap); }
    l->j[l->n] = j; l->d[l->n] = d; ++l->n;
}
/* corr (zeroed here) = the chain over list entries [e0, e1) for the columns of block [j0, j0+b) */
static void la_chain(const la_ctx* L, const la_list* ev, int64_t e0, int64_t e1, int64_t j0, int64_t b, float* corr)
{
    for (int64_t c = 0; c < b; ++c) corr[c] = 0.0f;
    for (int64_t e = e0; e < e1; ++e) {
        const float* xe = L->X + ev->j[e] * L->ld;
        for (int64_t c = 0; c < b; ++c) {
            const float g = dot_acc(xe, L->X + (j0 + c) * L->ld, L->n, L->acc);
            corr[c] = fmaf(ev->d[e], g, corr[c]);
        }
    }
}
static void la_apply(const la_ctx* L, const la_list* ev, float* r)
{
    for (int64_t e = 0; e < ev->n; ++e) axpy_f32(ev->d[e], L->X + ev->j[e] * L->ld, r, L->n);
}
typedef void (*la_block_fn)(void* u, int64_t j0, int64_t b, float* rhs_b, const float* G);
static int la_group_sweep(const la_ctx* L, int64_t p, const int64_t* block_starts, int64_t nblocks, const float* grams,
                          float* r, const float* alpha, la_block_fn fn, void* u)
{
    const int m = g_la_group;
    la_list prev = {0}, prevprev = {0};
    const float* G = grams;
    int64_t bmax = 0;
    for (int64_t bi = 0; bi < nblocks; ++bi) { const int64_t b = (bi + 1 < nblocks ? block_starts[bi + 1] : p) - block_starts[bi]; if (b > bmax) bmax = b; }
    float* srhs = (float*)malloc(sizeof(float) * (size_t)(bmax * m));
    float* cG = (float*)malloc(sizeof(float) * (size_t)bmax), *cW = (float*)malloc(sizeof(float) * (size_t)bmax), *cP = (float*)malloc(sizeof(float) * (size_t)bmax);
    float* a0 = (float*)malloc(sizeof(float) * (size_t)bmax);
    for (int64_t bi0 = 0; bi0 < nblocks; bi0 += m) {
        const int ns = (int)((nblocks - bi0) < m ? (nblocks - bi0) : m);
        la_apply(L, &prevprev, r);                                      /* the residual now holds the groups <= g-2 */
        prevprev.n = 0;
        for (int s = 0; s < ns; ++s) {
            const int64_t j0 = block_starts[bi0 + s], b = (bi0 + s + 1 < nblocks ? block_starts[bi0 + s + 1] : p) - j0;
            for (int64_t c = 0; c < b; ++c) srhs[s * bmax + c] = dot_xr(L->X + (j0 + c) * L->ld, r, L->n, L->acc);
        }
        la_list cur = {0};
        int64_t bound[5] = {0, 0, 0, 0, 0};                             /* cur entries of block s: [bound[s], bound[s+1]) */
        for (int s = 0; s < ns; ++s) {
            const int64_t j0 = block_starts[bi0 + s], b = (bi0 + s + 1 < nblocks ? block_starts[bi0 + s + 1] : p) - j0;
            float* rhs_b = srhs + s * bmax;
            la_chain(L, &prev, 0, prev.n, j0, b, cG);
            if (s & 1) la_chain(L, &cur, bound[s - 1], bound[s], j0, b, cW); else for (int64_t c = 0; c < b; ++c) cW[c] = 0.0f;
            if (m == 4 && s >= 2) la_chain(L, &cur, 0, bound[2], j0, b, cP); else for (int64_t c = 0; c < b; ++c) cP[c] = 0.0f;
            for (int64_t c = 0; c < b; ++c) rhs_b[c] = rhs_b[c] + ((cW[c] + cG[c]) + cP[c]);
            memcpy(a0, alpha + j0, sizeof(float) * (size_t)b);
            fn(u, j0, b, rhs_b, G);
            for (int64_t c = 0; c < b; ++c) { const float d = a0[c] - alpha[j0 + c]; if (d != 0.0f) la_list_push(&cur, j0 + c, d); }
            bound[s + 1] = cur.n;
            G += b * b;
        }
        free(prevprev.j); free(prevprev.d);
        prevprev = prev; prev = cur;
    }
    la_apply(L, &prevprev, r);
    la_apply(L, &prev, r);
    free(prevprev.j); free(prevprev.d); free(prev.j); free(prev.d);
    free(srhs); free(cG); free(cW); free(cP); free(a0);
    return 0;
}

typedef struct { const float* xpx; float *alpha, *beta, *delta; float ie; const float* var_effects; const double* pi; int nreps_arg;
                 uint64_t seed; uint32_t iter, marker0; } la_abc_user;
static void la_abc_block(void* uu, int64_t j0, int64_t b, float* rhs_b, const float* G)
{
    la_abc_user* U = (la_abc_user*)uu;
    const int nreps = U->nreps_arg > 0 ? U->nreps_arg : (int)b;
    for (int rep = 0; rep < nreps; ++rep)
        for (int64_t k = 0; k < b; ++k) {
            const int64_t j = j0 + k;
            const uint32_t mk = U->marker0 + (uint32_t)j;
            const double u = orc_uniform(U->seed, mk, U->iter, (uint32_t)rep, 0);
            const double z = orc_normal(U->seed, mk, U->iter, (uint32_t)rep, 0);
            const float a = abc_update(rhs_b[k], U->xpx[j], &U->alpha[j], &U->beta[j], &U->delta[j], U->ie, U->var_effects[j], U->pi[j], u, z);
            if (a != 0.0f) axpy_f32(a, G + k * b, rhs_b, b);
        }
}
typedef struct { const float* xpx; float* alpha; int32_t* delta; float ie, sigma_sq; const double* pi; int pi_is_matrix; const double* gamma;
                 int nreps_arg; uint64_t seed; uint32_t iter, marker0; } la_r_user;
static void la_r_block(void* uu, int64_t j0, int64_t b, float* rhs_b, const float* G)
{
    la_r_user* U = (la_r_user*)uu;
    const int nreps = U->nreps_arg > 0 ? U->nreps_arg : (int)b;
    for (int rep = 0; rep < nreps; ++rep)
        for (int64_t k = 0; k < b; ++k) {
            const int64_t j = j0 + k;
            const uint32_t mk = U->marker0 + (uint32_t)j;
            const double u = orc_uniform(U->seed, mk, U->iter, (uint32_t)rep, 0);
            const double z = orc_normal(U->seed, mk, U->iter, (uint32_t)rep, 0);
            const float a = bayesr_update(rhs_b[k], U->xpx[j], &U->alpha[j], &U->delta[j], U->ie, U->sigma_sq,
                                          U->pi_is_matrix ? U->pi + 4 * j : U->pi, U->gamma, u, z);
            if (a != 0.0f) axpy_f32(a, G + k * b, rhs_b, b);
        }
}

int orc_bayesabc_lookahead_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                                 const int64_t* block_starts, int64_t nblocks, const float* grams,
                                 float* r, float* alpha, float* beta, float* delta,
                                 float vare, const float* var_effects, const double* pi,
                                 int nreps_arg, uint64_t seed, uint32_t iter, uint32_t marker0, int acc)
{
    if (!abc_args_ok(n, p, ld, vare) || !blocks_ok(block_starts, nblocks, p)) return -1;
    const float ie = 1.0f / vare;
    const la_ctx L = { n, ld, X, acc };
    if (g_la_group > 1) {
        la_abc_user U = { xpx, alpha, beta, delta, ie, var_effects, pi, nreps_arg, seed, iter, marker0 };
        return la_group_sweep(&L, p, block_starts, nblocks, grams, r, alpha, la_abc_block, &U);
    }
    const float* G = grams;
    float* dprev = NULL; int64_t jprev = 0, bprev = 0;
    for (int64_t bi = 0; bi < nblocks; ++bi) {
        const int64_t j0 = block_starts[bi];
        const int64_t b  = (bi + 1 < nblocks ? block_starts[bi + 1] : p) - j0;
        float* rhs_b = (float*)malloc(sizeof(float) * (size_t)b);
        float* a0    = (float*)malloc(sizeof(float) * (size_t)b);
        la_block_rhs(&L, 1, r, ld, j0, b, jprev, bprev, dprev, rhs_b);
        memcpy(a0, alpha + j0, sizeof(float) * (size_t)b);
        const int nreps = nreps_arg > 0 ? nreps_arg : (int)b;
        for (int rep = 0; rep < nreps; ++rep)
            for (int64_t k = 0; k < b; ++k) {
                const int64_t j = j0 + k;
                const uint32_t m = marker0 + (uint32_t)j;
                const double u = orc_uniform(seed, m, iter, (uint32_t)rep, 0);
                const double z = orc_normal(seed, m, iter, (uint32_t)rep, 0);
                const float a = abc_update(rhs_b[k], xpx[j], &alpha[j], &beta[j], &delta[j], ie, var_effects[j], pi[j], u, z);
                if (a != 0.0f) axpy_f32(a, G + k * b, rhs_b, b);
            }
        for (int64_t k = 0; k < b; ++k) a0[k] = a0[k] - alpha[j0 + k];          /* net change of the block */
        free(dprev); free(rhs_b);
        dprev = a0; jprev = j0; bprev = b;
        G += b * b;
    }
    for (int64_t e = 0; e < bprev; ++e) if (dprev[e] != 0.0f) axpy_f32(dprev[e], X + (jprev + e) * ld, r, n);
    free(dprev);
    return 0;
}

int orc_bayesr_lookahead_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                               const int64_t* block_starts, int64_t nblocks, const float* grams,
                               float* r, float* alpha, int32_t* delta,
                               float vare, float sigma_sq, const double* pi, int pi_is_matrix,
                               const double* gamma, int nreps_arg,
                               uint64_t seed, uint32_t iter, uint32_t marker0, int acc)
{
    if (!abc_args_ok(n, p, ld, vare) || !bayesr_priors_ok(pi, pi_is_matrix, p) ||
        !blocks_ok(block_starts, nblocks, p)) return -1;
    if (!(sigma_sq > 0.0f)) return -2;
    const float ie = 1.0f / vare;
    const la_ctx L = { n, ld, X, acc };
    if (g_la_group > 1) {
        la_r_user U = { xpx, alpha, delta, ie, sigma_sq, pi, pi_is_matrix, gamma, nreps_arg, seed, iter, marker0 };
        return la_group_sweep(&L, p, block_starts, nblocks, grams, r, alpha, la_r_block, &U);
    }
    const float* G = grams;
    float* dprev = NULL; int64_t jprev = 0, bprev = 0;
    for (int64_t bi = 0; bi < nblocks; ++bi) {
        const int64_t j0 = block_starts[bi];
        const int64_t b  = (bi + 1 < nblocks ? block_starts[bi + 1] : p) - j0;
        float* rhs_b = (float*)malloc(sizeof(float) * (size_t)b);
        float* a0    = (float*)malloc(sizeof(float) * (size_t)b);
        la_block_rhs(&L, 1, r, ld, j0, b, jprev, bprev, dprev, rhs_b);
        memcpy(a0, alpha + j0, sizeof(float) * (size_t)b);
        const int nreps = nreps_arg > 0 ? nreps_arg : (int)b;
        for (int rep = 0; rep < nreps; ++rep)
            for (int64_t k = 0; k < b; ++k) {
                const int64_t j = j0 + k;
                const uint32_t m = marker0 + (uint32_t)j;
                const double u = orc_uniform(seed, m, iter, (uint32_t)rep, 0);
                const double z = orc_normal(seed, m, iter, (uint32_t)rep, 0);
                const float a = bayesr_update(rhs_b[k], xpx[j], &alpha[j], &delta[j], ie, sigma_sq,
                                              pi_is_matrix ? pi + 4 * j : pi, gamma, u, z);
                if (a != 0.0f) axpy_f32(a, G + k * b, rhs_b, b);
            }
        for (int64_t k = 0; k < b; ++k) a0[k] = a0[k] - alpha[j0 + k];
        free(dprev); free(rhs_b);
        dprev = a0; jprev = j0; bprev = b;
        G += b * b;
    }
    for (int64_t e = 0; e < bprev; ++e) if (dprev[e] != 0.0f) axpy_f32(dprev[e], X + (jprev + e) * ld, r, n);
    free(dprev);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* RULE T (the device's jwas_sweep_params.section_solve; csrc/sampler_mt.hpp): the dense chain of a */
/* 64-marker section as the triangular solve it is.  Sampler I, full 256-marker blocks, single     */
/* pass.  A section in which every marker is in the model for every trait at entry: Rule L makes   */
/* its chain (I + L) D = y,  L[(l,k),(j,m)] = A_l[k][m] G_lj (j < l),  y_l = alpha_l - (A_l (rhs_l */
/* + d_l alpha_l) + c_l),  D = alpha_old - alpha_new.  The device forms T = (I + L)^-1 once per     */
/* sweep (k_section_inverse_mt) and a section's effects with one mat-vec D~ = T y; the literal     */
/* evaluation (MTBayesABC.jl:85-120) at the right-hand side those effects imply verifies that every */
/* indicator stays 1.  EXCEPTIONS (a marker that is not in the model for every trait at entry, or    */
/* that the verification finds leaving it) are taken in marker order: T is lower block triangular,  */
/* so everything before the first exception e is final, the literal evaluation of e at the          */
/* right-hand side the solve implies for it, rhs + R Lc_e (y_e - D~_e), IS the chain's evaluation   */
/* of e (whatever y_e was: the row only recovers sum_{j<e} G_ej D_j from D~), its result v =         */
/* alpha_old - alpha_new replaces D~_e, and the rows behind it take the rank-t correction             */
/* D~_r += sum_m T[r,(e,m)] (v_m - D~_(e,m)) -- the solve of the same system with row e replaced by  */
/* "D_e = v".  More than ORC_SOLVE_MAX_ODD markers outside the model at entry: the section is not    */
/* tried; more than ORC_SOLVE_MAX_EXC exceptions in all: it falls back to the sequential chain.      */
/* Restated operation for operation (same accumulation types and orders), so that the comparison    */
/* stays bit for bit:                                                                                 */
/*   T column (jc, mc): rows of markers l < jc are 0, of jc the identity; for l > jc                 */
/*       u_m = (p0 + p1) + (p2 + p3),  p_q = sum over j = jc + q, jc + q + 4, ... < l of                */
/*             fma(G_lj, T[(j,m)], p_q)   (double, ascending j: the device's four lanes per column)   */
/*       T[(l,k)] = fl32( -( sum_m fma(A_l[k][m], u_m, .) ) )  (double, ascending m)                 */
/*   D~[(l,k)] = (p0 + p1) + (p2 + p3),  p_q = fmaf chain (float, from 0) over the columns           */
/*       c = m*64 + j in [16 t q, 16 t (q+1)), ascending                                              */
/*   alpha_new = alpha - D~ ;  D = alpha - alpha_new ;  rhs~ = rhs + R Lc (y - D~)                    */
/*   exception e:  v_k = alpha_k - alpha_new_k (literal) ; del_k = v_k - D~[(e,k)] ;  for l > e:          */
/*       D~[(l,k)] = fmaf(T[(l,k),(e,m)], del_m, D~[(l,k)]) for m ascending ;  D~[(e,k)] = v_k            */
/* Reference chain it stands for: MTBayesABC.jl:243-333 (block form of sampler I).                  */
/* ------------------------------------------------------------------------------------------ */
static int g_section_solve = 0;
void orc_set_section_solve(int on) { g_section_solve = on; }
enum { ORC_SOLVE_MAX_ODD = 16, ORC_SOLVE_MAX_EXC = 24 };            /* (csrc/sampler_mt.hpp: kSolveMaxOdd, kSolveMaxExc) */
static int64_t g_solve_sections = 0, g_solve_fallbacks = 0, g_solve_exceptions = 0;      /* diagnostics: sections solved / fallen back, exceptions of the solved ones, since the last reset */
void orc_section_solve_counts(int64_t* solved, int64_t* fallbacks, int reset)
{
    if (solved) *solved = g_solve_sections;
    if (fallbacks) *fallbacks = g_solve_fallbacks;
    if (reset) { g_solve_sections = 0; g_solve_fallbacks = 0; g_solve_exceptions = 0; }
}
int64_t orc_section_solve_exceptions(void) { return g_solve_exceptions; }

/* One 64-marker section [c0, c0 + 64) of a 256-marker block.  Returns 1 if it was solved (state, rhs_b updated), 0 if the
 * caller has to run it through the sequential chain (nothing touched). */
static int mt1_section_solve(int t, int64_t p, int64_t j0, int64_t b, int64_t c0, const float* G, const float* xpx,
                             float* rhs_b, float* alpha, float* beta, float* delta,
                             const float* vare, const float* Rinv, const float* Ginv_shared,
                             const double* log_prior, uint64_t seed, uint32_t iter, uint32_t marker0)
{
    enum { S = 64 };
    const int nr = S * t;
    int in_all[64], nodd = 0;
    for (int l = 0; l < S; ++l) {
        in_all[l] = 1;
        for (int k = 0; k < t; ++k) if (delta[k * p + j0 + c0 + l] != 1.0f) in_all[l] = 0;
        nodd += !in_all[l];
    }
    if (nodd > ORC_SOLVE_MAX_ODD) return 0;
    float* A   = (float*)malloc(sizeof(float) * (size_t)(S * t * t));
    float* cc  = (float*)malloc(sizeof(float) * (size_t)(S * t));
    float* y   = (float*)malloc(sizeof(float) * (size_t)nr);           /* [m*64 + j] */
    float* T   = (float*)calloc((size_t)nr * nr, sizeof(float));       /* [row (l,k) = k*64 + l][col m*64 + j] */
    float* Dt  = (float*)malloc(sizeof(float) * (size_t)nr);           /* [k*64 + l] */
    float* Gi  = (float*)malloc(sizeof(float) * (size_t)(S * t * t));  /* each marker's Ginv */
    float Gtmp[ORC_MAXT * ORC_MAXT];
    for (int l = 0; l < S; ++l) {
        const int64_t j = j0 + c0 + l;
        const float* Gl = marker_ginv(j, t, Ginv_shared, Gtmp);
        memcpy(Gi + l * t * t, Gl, sizeof(float) * (size_t)(t * t));
        float b_old[ORC_MAXT], w[ORC_MAXT], bo[ORC_MAXT];
        double z[ORC_MAXT];
        for (int k = 0; k < t; ++k) { b_old[k] = beta[k * p + j]; z[k] = orc_normal(seed, marker0 + (uint32_t)j, iter, 0, (uint32_t)k); }
        mt1_linear_coeffs(t, xpx[j], Rinv, Gi + l * t * t, b_old, z, A + l * t * t, cc + l * t);
        for (int k = 0; k < t; ++k) w[k] = rhs_b[k * b + c0 + l] + xpx[j] * alpha[k * p + j];
        mt1_linear_beta(t, A + l * t * t, cc + l * t, w, bo);
        for (int k = 0; k < t; ++k) y[k * S + l] = alpha[k * p + j] - bo[k];
    }
    for (int mc = 0; mc < t; ++mc)
        for (int jc = 0; jc < S; ++jc) {
            const int col = mc * S + jc;
            for (int k = 0; k < t; ++k) T[(size_t)(k * S + jc) * nr + col] = (k == mc) ? 1.0f : 0.0f;
            for (int l = jc + 1; l < S; ++l) {
                double u[ORC_MAXT], pu[4][ORC_MAXT];
                for (int q = 0; q < 4; ++q) {                                   /* the device's four lanes per column: j = jc + q, + 4, ... */
                    for (int m = 0; m < t; ++m) pu[q][m] = 0.0;
                    for (int j = jc + q; j < l; j += 4) {
                        const double g = (double)G[(c0 + l) * b + c0 + j];
                        for (int m = 0; m < t; ++m) pu[q][m] = fma(g, (double)T[(size_t)(m * S + j) * nr + col], pu[q][m]);
                    }
                }
                for (int m = 0; m < t; ++m) u[m] = (pu[0][m] + pu[1][m]) + (pu[2][m] + pu[3][m]);
                for (int k = 0; k < t; ++k) {
                    double v = 0.0;
                    for (int m = 0; m < t; ++m) v = fma((double)A[l * t * t + k * t + m], u[m], v);
                    T[(size_t)(k * S + l) * nr + col] = (float)(-v);
                }
            }
        }
    const int qc = 16 * t;                                                  /* columns per quarter (one wave's share on the device) */
    for (int r = 0; r < nr; ++r) {
        float pq[4];
        for (int q = 0; q < 4; ++q) {
            float acc = 0.0f;
            for (int cidx = qc * q; cidx < qc * (q + 1); ++cidx) acc = fmaf(T[(size_t)r * nr + cidx], y[cidx], acc);
            pq[q] = acc;
        }
        Dt[r] = (pq[0] + pq[1]) + (pq[2] + pq[3]);
    }
    /* the new effects: every marker verified in order, exceptions taken as they come (nothing is written before the end) */
    float* bo_all = (float*)malloc(sizeof(float) * (size_t)nr);          /* new alpha */
    float* bb_all = (float*)malloc(sizeof(float) * (size_t)nr);          /* new beta  */
    float* dd_all = (float*)malloc(sizeof(float) * (size_t)nr);          /* new delta */
    int ok = 1, nexc = 0;
    const int lin_save = g_mt_linear;
    for (int l = 0; l < S && ok; ++l) {
        const int64_t j = j0 + c0 + l;
        const float d = xpx[j];
        const float* Gl = Gi + l * t * t;
        float v[ORC_MAXT], q[ORC_MAXT], wev[ORC_MAXT], ta[ORC_MAXT], tb[ORC_MAXT], td[ORC_MAXT], aout[ORC_MAXT];
        for (int k = 0; k < t; ++k) v[k] = y[k * S + l] - Dt[k * S + l];
        for (int k = 0; k < t; ++k) {
            const float C11 = Gl[k * t + k] + Rinv[k * t + k] * d;
            float acc = C11 * v[k];
            for (int jj = 0; jj < k; ++jj) acc = fmaf(Gl[k * t + jj] + (d * 1.0f) * Rinv[k * t + jj], v[jj], acc);
            q[k] = acc;
        }
        for (int m = 0; m < t; ++m) {
            float acc = 0.0f;
            for (int k = 0; k < t; ++k) acc = fmaf(vare[m * t + k], q[k], acc);
            wev[m] = (rhs_b[m * b + c0 + l] + acc) + d * alpha[m * p + j];
        }
        for (int k = 0; k < t; ++k) { ta[k] = alpha[k * p + j]; tb[k] = beta[k * p + j]; td[k] = delta[k * p + j]; }
        g_mt_linear = 0;                                                       /* the literal order */
        mt1_update(t, wev, d, ta, tb, td, 1, Rinv, Gl, log_prior, seed, marker0 + (uint32_t)j, iter, 0, aout);
        g_mt_linear = lin_save;
        int stays = in_all[l];
        for (int k = 0; k < t; ++k) if (td[k] != 1.0f) stays = 0;
        if (stays) {                                                           /* the solve's own values */
            for (int k = 0; k < t; ++k) {
                bo_all[k * S + l] = alpha[k * p + j] - Dt[k * S + l];
                bb_all[k * S + l] = bo_all[k * S + l]; dd_all[k * S + l] = 1.0f;
            }
            continue;
        }
        if (++nexc > ORC_SOLVE_MAX_EXC) { ok = 0; break; }
        float del[ORC_MAXT];
        for (int k = 0; k < t; ++k) {                                          /* the literal evaluation IS the chain's: take it */
            const float ve = alpha[k * p + j] - ta[k];
            del[k] = ve - Dt[k * S + l];
            bo_all[k * S + l] = ta[k]; bb_all[k * S + l] = tb[k]; dd_all[k * S + l] = td[k];
        }
        for (int l2 = l + 1; l2 < S; ++l2)
            for (int k = 0; k < t; ++k) {
                float acc = Dt[k * S + l2];
                for (int m = 0; m < t; ++m) acc = fmaf(T[(size_t)(k * S + l2) * nr + m * S + l], del[m], acc);
                Dt[k * S + l2] = acc;
            }
        for (int k = 0; k < t; ++k) Dt[k * S + l] = alpha[k * p + j] - ta[k];
    }
    if (ok) {
        for (int l = 0; l < S; ++l) {
            const int64_t j = j0 + c0 + l;
            for (int k = 0; k < t; ++k) {
                const float a_old = alpha[k * p + j], an = bo_all[k * S + l];
                const float D = a_old - an;
                alpha[k * p + j] = an; beta[k * p + j] = bb_all[k * S + l]; delta[k * p + j] = dd_all[k * S + l];
                if (D != 0.0f) axpy_f32(D, G + (c0 + l) * b, rhs_b + k * b, b);       /* later sections see the changes in marker order */
            }
        }
        ++g_solve_sections; g_solve_exceptions += nexc;
    } else ++g_solve_fallbacks;
    free(bb_all); free(dd_all);
    free(A); free(cc); free(y); free(T); free(Dt); free(Gi); free(bo_all);
    return ok;
}

int orc_mt_lookahead_sweep(int kind, const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                                   const int64_t* block_starts, int64_t nblocks, const float* grams,
                                   int t, float* r, int64_t ld_r, float* alpha, float* beta, float* delta,
                                   const float* vare, const float* var_effect,
                                   const double* log_prior, int prior_is_matrix, int nreps_arg,
                                   uint64_t seed, uint32_t iter, uint32_t marker0, int acc)
{
    if (t < 1 || t > ORC_MAXT || n <= 0 || ld < n || ld_r < n || !blocks_ok(block_starts, nblocks, p)) return -1;
    float Rinv[ORC_MAXT * ORC_MAXT], Ginv[ORC_MAXT * ORC_MAXT], Gtmp_[ORC_MAXT * ORC_MAXT];
    (void)Gtmp_;
    if (kind < MT_SAMPLER_I || kind > MT_MEGA) return -1;
    if (inv_small(vare, t, Rinv) || inv_small(var_effect, t, Ginv)) return -2;
    const int nstates = 1 << t;
    const la_ctx L = { n, ld, X, acc };
    const float* G = grams;
    float* dprev = NULL; int64_t jprev = 0, bprev = 0;
    for (int64_t bi = 0; bi < nblocks; ++bi) {
        const int64_t j0 = block_starts[bi];
        const int64_t b  = (bi + 1 < nblocks ? block_starts[bi + 1] : p) - j0;
        float* rhs_b = (float*)malloc(sizeof(float) * (size_t)(b * t));
        float* a0    = (float*)malloc(sizeof(float) * (size_t)(b * t));
        la_block_rhs(&L, t, r, ld_r, j0, b, jprev, bprev, dprev, rhs_b);
        for (int k = 0; k < t; ++k) memcpy(a0 + k * b, alpha + k * p + j0, sizeof(float) * (size_t)b);
        const int nreps = nreps_arg > 0 ? nreps_arg : (int)b;
        /* Rule T (above): full 256-marker blocks of sampler I in a single pass, section by section */
        const int solve = g_section_solve && kind == MT_SAMPLER_I && b == 256 && nreps == 1 && !prior_is_matrix && t <= 3;
        for (int rep = 0; rep < nreps; ++rep)
            for (int64_t c = 0; c < b; ++c) {
                if (solve && (c & 63) == 0 &&
                    mt1_section_solve(t, p, j0, b, c, G, xpx, rhs_b, alpha, beta, delta, vare, Rinv, Ginv, log_prior, seed, iter, marker0)) {
                    c += 63;
                    continue;
                }
                const int64_t j = j0 + c;
                float w[ORC_MAXT], a[ORC_MAXT];
                for (int k = 0; k < t; ++k) w[k] = rhs_b[k * b + c];
                mt_update(kind, t, w, xpx[j], alpha + j, beta + j, delta + j, p, Rinv, marker_ginv(j, t, Ginv, Gtmp_), vare, marker_var(j, t, var_effect),
                          prior_is_matrix ? log_prior + (int64_t)nstates * j : log_prior,
                          seed, marker0 + (uint32_t)j, iter, (uint32_t)rep, a);
                for (int k = 0; k < t; ++k) if (a[k] != 0.0f) axpy_f32(a[k], G + c * b, rhs_b + k * b, b);
            }
        for (int k = 0; k < t; ++k)
            for (int64_t c = 0; c < b; ++c) a0[k * b + c] = a0[k * b + c] - alpha[k * p + j0 + c];
        free(dprev); free(rhs_b);
        dprev = a0; jprev = j0; bprev = b;
        G += b * b;
    }
    for (int64_t e = 0; e < bprev; ++e)
        for (int k = 0; k < t; ++k)
            if (dprev[k * bprev + e] != 0.0f) axpy_f32(dprev[k * bprev + e], X + (jprev + e) * ld, r + k * ld_r, n);
    free(dprev);
    return 0;
}

/* exported block forms: exact chain (independent = 0) and independent blocks (BayesABC.jl:190-255, BayesR.jl:195-273,
 * MTBayesABC.jl:335-440: every block RHS from the same residual snapshot; r += sum_b X_b*(alpha_old_b - alpha_b) afterwards) */
int orc_bayesabc_block_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                             const int64_t* block_starts, int64_t nblocks, const float* grams,
                             float* r, float* alpha, float* beta, float* delta,
                             float vare, const float* var_effects, const double* pi,
                             int nreps_arg, uint64_t seed, uint32_t iter, uint32_t marker0, int acc)
{
    return bayesabc_block_sweep_impl(X, n, p, ld, xpx, block_starts, nblocks, grams, r, alpha, beta, delta, vare, var_effects, pi, nreps_arg, seed, iter, marker0, acc, 0);
}

int orc_bayesabc_indep_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                             const int64_t* block_starts, int64_t nblocks, const float* grams,
                             float* r, float* alpha, float* beta, float* delta,
                             float vare, const float* var_effects, const double* pi,
                             int nreps_arg, uint64_t seed, uint32_t iter, uint32_t marker0, int acc)
{
    return bayesabc_block_sweep_impl(X, n, p, ld, xpx, block_starts, nblocks, grams, r, alpha, beta, delta, vare, var_effects, pi, nreps_arg, seed, iter, marker0, acc, 1);
}

int orc_bayesr_block_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                           const int64_t* block_starts, int64_t nblocks, const float* grams,
                           float* r, float* alpha, int32_t* delta,
                           float vare, float sigma_sq, const double* pi, int pi_is_matrix,
                           const double* gamma, int nreps_arg,
                           uint64_t seed, uint32_t iter, uint32_t marker0, int acc)
{
    return bayesr_block_sweep_impl(X, n, p, ld, xpx, block_starts, nblocks, grams, r, alpha, delta, vare, sigma_sq, pi, pi_is_matrix, gamma, nreps_arg, seed, iter, marker0, acc, 0);
}

int orc_bayesr_indep_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                           const int64_t* block_starts, int64_t nblocks, const float* grams,
                           float* r, float* alpha, int32_t* delta,
                           float vare, float sigma_sq, const double* pi, int pi_is_matrix,
                           const double* gamma, int nreps_arg,
                           uint64_t seed, uint32_t iter, uint32_t marker0, int acc)
{
    return bayesr_block_sweep_impl(X, n, p, ld, xpx, block_starts, nblocks, grams, r, alpha, delta, vare, sigma_sq, pi, pi_is_matrix, gamma, nreps_arg, seed, iter, marker0, acc, 1);
}

int orc_mt_block_sweep(int kind, const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                               const int64_t* block_starts, int64_t nblocks, const float* grams,
                               int t, float* r, int64_t ld_r, float* alpha, float* beta, float* delta,
                               const float* vare, const float* var_effect,
                               const double* log_prior, int prior_is_matrix, int nreps_arg,
                               uint64_t seed, uint32_t iter, uint32_t marker0, int acc)
{
    return mt_block_sweep_impl(kind, X, n, p, ld, xpx, block_starts, nblocks, grams, t, r, ld_r, alpha, beta, delta, vare, var_effect, log_prior, prior_is_matrix, nreps_arg, seed, iter, marker0, acc, 0);
}

int orc_mt_indep_sweep(int kind, const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                               const int64_t* block_starts, int64_t nblocks, const float* grams,
                               int t, float* r, int64_t ld_r, float* alpha, float* beta, float* delta,
                               const float* vare, const float* var_effect,
                               const double* log_prior, int prior_is_matrix, int nreps_arg,
                               uint64_t seed, uint32_t iter, uint32_t marker0, int acc)
{
    return mt_block_sweep_impl(kind, X, n, p, ld, xpx, block_starts, nblocks, grams, t, r, ld_r, alpha, beta, delta, vare, var_effect, log_prior, prior_is_matrix, nreps_arg, seed, iter, marker0, acc, 1);
}

/* sampler I entry points (kept for the existing callers) */
int orc_mtbayesc_I_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                         int t, float* r, int64_t ld_r, float* alpha, float* beta, float* delta,
                         const float* vare, const float* var_effect,
                         const double* log_prior, int prior_is_matrix,
                         uint64_t seed, uint32_t iter, uint32_t marker0, int acc)
{
    return orc_mt_sweep(MT_SAMPLER_I, X, n, p, ld, xpx, t, r, ld_r, alpha, beta, delta, vare, var_effect,
                        log_prior, prior_is_matrix, seed, iter, marker0, acc);
}

int orc_mtbayesc_I_block_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                               const int64_t* block_starts, int64_t nblocks, const float* grams,
                               int t, float* r, int64_t ld_r, float* alpha, float* beta, float* delta,
                               const float* vare, const float* var_effect,
                               const double* log_prior, int prior_is_matrix, int nreps_arg,
                               uint64_t seed, uint32_t iter, uint32_t marker0, int acc)
{
    return orc_mt_block_sweep(MT_SAMPLER_I, X, n, p, ld, xpx, block_starts, nblocks, grams, t, r, ld_r, alpha, beta,
                              delta, vare, var_effect, log_prior, prior_is_matrix, nreps_arg, seed, iter, marker0, acc);
}

int orc_mtbayesc_I_lookahead_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                                   const int64_t* block_starts, int64_t nblocks, const float* grams,
                                   int t, float* r, int64_t ld_r, float* alpha, float* beta, float* delta,
                                   const float* vare, const float* var_effect,
                                   const double* log_prior, int prior_is_matrix, int nreps_arg,
                                   uint64_t seed, uint32_t iter, uint32_t marker0, int acc)
{
    return orc_mt_lookahead_sweep(MT_SAMPLER_I, X, n, p, ld, xpx, block_starts, nblocks, grams, t, r, ld_r, alpha,
                                  beta, delta, vare, var_effect, log_prior, prior_is_matrix, nreps_arg, seed, iter,
                                  marker0, acc);
}


/* ------------------------------------------------------------------------------------------ */
/* running posterior means (output.jl:568-577)                                                 */
/* ------------------------------------------------------------------------------------------ */
void orc_accumulate(const float* alpha, const void* delta, int delta_is_class, int64_t p, double k,
                    float* mean_alpha, float* mean_alpha2, float* mean_delta)
{
    const float*   df = (const float*)delta;
    const int32_t* di = (const int32_t*)delta;
    for (int64_t j = 0; j < p; ++j) {
        /* Float32 arrays divided by the Float64 sample count, stored back into Float32 arrays */
        mean_alpha[j]  = (float)((double)mean_alpha[j]  + ((double)(alpha[j] - mean_alpha[j])) / k);
        const float a2 = alpha[j] * alpha[j];
        mean_alpha2[j] = (float)((double)mean_alpha2[j] + ((double)(a2 - mean_alpha2[j])) / k);
        const float ind = delta_is_class ? (di[j] > 1 ? 1.0f : 0.0f) : df[j];
        mean_delta[j]  = (float)((double)mean_delta[j]  + ((double)(ind - mean_delta[j])) / k);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* 2-bit codec (streaming_genotypes.jl:978-1002)                                               */
/* ------------------------------------------------------------------------------------------ */
void orc_decode_marker_2bit(const uint8_t* payload, int64_t n, int64_t j, float mean, int centered,
                            float* out)
{
    const int64_t stride = (n + 3) / 4;                                   /* cld(n,4)  :364-367 */
    const uint8_t* col = payload + j * stride;
    for (int64_t i = 0; i < n; ++i) {
        const unsigned code = (col[i >> 2] >> ((i & 3) << 1)) & 3u;      /* :993-995 */
        float v = code == 3u ? mean : (float)code;                       /* :996-997 */
        if (centered) v -= mean;                                         /* :998-999 */
        out[i] = v;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* CPU baseline timing: the reference's per-marker operation order (sdot, scalar update, saxpy) */
/* ------------------------------------------------------------------------------------------ */
static float dot_f32_mt(const float* a, const float* b, int64_t n, int nthreads)
{
    float s = 0.0f;
    if (nthreads <= 1) {
        /* 8 independent partial sums: what a SIMD sdot kernel does */
        float p[8] = {0};
        int64_t i = 0;
        for (; i + 8 <= n; i += 8)
            for (int q = 0; q < 8; ++q) p[q] += a[i + q] * b[i + q];
        for (; i < n; ++i) p[0] += a[i] * b[i];
        for (int q = 0; q < 8; ++q) s += p[q];
        return s;
    }
#ifdef _OPENMP
#pragma omp parallel for reduction(+:s) num_threads(nthreads) schedule(static)
#endif
    for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

static void axpy_f32_mt(float a, const float* x, float* y, int64_t n, int nthreads)
{
    if (nthreads <= 1) { axpy_f32(a, x, y, n); return; }
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for (int64_t i = 0; i < n; ++i) y[i] = fmaf(a, x[i], y[i]);
}

double orc_time_bayesc_sweeps(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                              float* r, float* alpha, float* beta, float* delta,
                              float vare, float var_effect, double pi,
                              uint64_t seed, int sweeps, int nthreads)
{
    struct timespec t0, t1;
    const float ie = 1.0f / vare;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int it = 0; it < sweeps; ++it)
        for (int64_t j = 0; j < p; ++j) {
            const float* x = X + j * ld;
            const float s = dot_f32_mt(x, r, n, nthreads);
            const double u = orc_uniform(seed, (uint32_t)j, (uint32_t)it + 1u, 0, 0);
            const double z = orc_normal(seed, (uint32_t)j, (uint32_t)it + 1u, 0, 0);
            const float a = abc_update(s, xpx[j], &alpha[j], &beta[j], &delta[j], ie, var_effect, pi, u, z);
            if (a != 0.0f) axpy_f32_mt(a, x, r, n, nthreads);
        }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ------------------------------------------------------------------------------------------ */
/* CPU baseline timing with a PERSISTENT thread team (bench.py cpu_baseline leg).               */
/* The reference's non-block per-marker order (fp32 dot, scalar update, conditional fp32 axpy:  */
/* BayesABC.jl:73-79, BayesR.jl:56-96, MTBayesABC.jl:80-121) with the rows of the dot / axpy    */
/* split over `nthreads` threads that live for the whole run and meet at ONE spin barrier per   */
/* marker -- what a well-threaded level-1 BLAS can do at best (a fork/join per dot, as in       */
/* orc_time_bayesc_sweeps, costs more than the 50k-element dot itself beyond ~8 threads).        */
/* kind: 0 = BayesC, 1 = BayesR, 2 = multi-trait sampler I.                                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct { volatile int count; volatile int sense; char pad[56]; } team_barrier;

static inline void team_wait(team_barrier* b, int nthreads, int* local_sense)
{
    const int s = !*local_sense;
    *local_sense = s;
    if (__atomic_add_fetch(&b->count, 1, __ATOMIC_ACQ_REL) == nthreads) {
        b->count = 0;
        __atomic_store_n(&b->sense, s, __ATOMIC_RELEASE);
    } else {
        while (__atomic_load_n(&b->sense, __ATOMIC_ACQUIRE) != s) __builtin_ia32_pause();
    }
}

static inline float dot8_f32(const float* a, const float* b, int64_t n)
{
    float q[8] = {0};
    int64_t i = 0;
    for (; i + 8 <= n; i += 8)
        for (int u = 0; u < 8; ++u) q[u] += a[i + u] * b[i + u];
    for (; i < n; ++i) q[0] += a[i] * b[i];
    float s = 0.0f;
    for (int u = 0; u < 8; ++u) s += q[u];
    return s;
}

double orc_time_sweeps_team(int kind, const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                            int t, float* r, int64_t ld_r, float* alpha, float* beta, void* delta,
                            const float* vare, const float* var_effect, const double* prior, const double* gamma,
                            uint64_t seed, int sweeps, int nthreads, double max_seconds, int64_t* markers_done)
{
    const int forkjoin = nthreads < 0;          /* nthreads < 0: |nthreads| threads, one fork/join per dot / axpy (what a */
    if (forkjoin) nthreads = -nthreads;         /* threaded BLAS does per call) instead of a persistent team              */
    if (nthreads < 1) nthreads = 1;
    volatile int stop = 0;                      /* set by thread 0 when max_seconds (> 0) have passed; read after the barrier */
    volatile int64_t done = 0;
    if (kind < 0 || kind > 2 || t < 1 || t > ORC_MAXT || (kind != 2 && t != 1)) return -1.0;
    float Rinv[ORC_MAXT * ORC_MAXT] = {0}, Ginv[ORC_MAXT * ORC_MAXT] = {0};
    if (kind == 2 && (inv_small(vare, t, Rinv) || inv_small(var_effect, t, Ginv))) return -1.0;
    const float ie = 1.0f / vare[0];
    if (forkjoin) {
        struct timespec f0, f1;
        clock_gettime(CLOCK_MONOTONIC, &f0);
        int64_t step = 0;
        int out_of_time = 0;
        for (int it = 0; it < sweeps && !out_of_time; ++it)
            for (int64_t j = 0; j < p; ++j, ++step) {
                if (max_seconds > 0.0 && (step & 15) == 0) {
                    clock_gettime(CLOCK_MONOTONIC, &f1);
                    if ((double)(f1.tv_sec - f0.tv_sec) + 1e-9 * (double)(f1.tv_nsec - f0.tv_nsec) > max_seconds) { out_of_time = 1; break; }
                }
                const float* x = X + j * ld;
                float s[ORC_MAXT], a[ORC_MAXT];
                for (int k = 0; k < t; ++k) s[k] = dot_f32_mt(x, r + k * ld_r, n, nthreads);
                const uint32_t m = (uint32_t)j, iter = (uint32_t)it + 1u;
                if (kind == 0) {
                    const double u = orc_uniform(seed, m, iter, 0, 0), z = orc_normal(seed, m, iter, 0, 0);
                    a[0] = abc_update(s[0], xpx[j], &alpha[j], &beta[j], &((float*)delta)[j], ie, var_effect[0], prior[0], u, z);
                } else if (kind == 1) {
                    const double u = orc_uniform(seed, m, iter, 0, 0), z = orc_normal(seed, m, iter, 0, 0);
                    a[0] = bayesr_update(s[0], xpx[j], &alpha[j], &((int32_t*)delta)[j], ie, var_effect[0], prior, gamma, u, z);
                } else {
                    float w[ORC_MAXT];
                    for (int k = 0; k < t; ++k) w[k] = s[k] + xpx[j] * alpha[k * p + j];
                    mt1_update(t, w, xpx[j], alpha + j, beta + j, (float*)delta + j, p, Rinv, Ginv, prior, seed, m, iter, 0, a);
                }
                for (int k = 0; k < t; ++k) if (a[k] != 0.0f) axpy_f32_mt(a[k], x, r + k * ld_r, n, nthreads);
            }
        clock_gettime(CLOCK_MONOTONIC, &f1);
        if (markers_done) *markers_done = step;
        return (double)(f1.tv_sec - f0.tv_sec) + 1e-9 * (double)(f1.tv_nsec - f0.tv_nsec);
    }
    float* partial = (float*)aligned_alloc(64, (size_t)2 * nthreads * 16 * sizeof(float));   /* [2][T][16]: one line per thread */
    team_barrier bar; bar.count = 0; bar.sense = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num(), T = omp_get_num_threads();
#else
        const int tid = 0, T = 1;
#endif
        int sense = 0;
        const int64_t chunk = ((n + T - 1) / T + 15) / 16 * 16;
        const int64_t lo = chunk * tid < n ? chunk * tid : n, hi = lo + chunk < n ? lo + chunk : n;
        int64_t pend_j = -1;                                   /* thread 0: state of the previous marker, written one barrier late */
        float pa[ORC_MAXT], pb[ORC_MAXT], pd[ORC_MAXT]; int32_t pcls = 0;
        int64_t step = 0;
        for (int it = 0; it < sweeps; ++it)
            for (int64_t j = 0; j < p; ++j, ++step) {
                const float* x = X + j * ld;
                float* mine = partial + ((size_t)(step & 1) * T + tid) * 16;
                for (int k = 0; k < t; ++k) mine[k] = dot8_f32(x + lo, r + k * ld_r + lo, hi - lo);
                if (tid == 0 && max_seconds > 0.0 && (step & 63) == 0) {
                    struct timespec tn;
                    clock_gettime(CLOCK_MONOTONIC, &tn);
                    if ((double)(tn.tv_sec - t0.tv_sec) + 1e-9 * (double)(tn.tv_nsec - t0.tv_nsec) > max_seconds) stop = 1;
                }
                team_wait(&bar, T, &sense);                    /* (the flag written before the barrier is visible after it) */
                if (stop) goto finished;
                if (tid == 0) done = step + 1;
                if (tid == 0 && pend_j >= 0) {                 /* every thread has finished reading the state of marker pend_j */
                    for (int k = 0; k < t; ++k) { alpha[k * p + pend_j] = pa[k]; if (kind != 1) { beta[k * p + pend_j] = pb[k]; ((float*)delta)[k * p + pend_j] = pd[k]; } }
                    if (kind == 1) ((int32_t*)delta)[pend_j] = pcls;
                }
                float s[ORC_MAXT], a[ORC_MAXT];
                const float* all = partial + (size_t)(step & 1) * T * 16;
                for (int k = 0; k < t; ++k) { float v = 0.0f; for (int q = 0; q < T; ++q) v += all[q * 16 + k]; s[k] = v; }
                float la[ORC_MAXT], lb[ORC_MAXT], ldl[ORC_MAXT]; int32_t lcls = 0;
                for (int k = 0; k < t; ++k) { la[k] = alpha[k * p + j]; if (kind != 1) { lb[k] = beta[k * p + j]; ldl[k] = ((float*)delta)[k * p + j]; } }
                const uint32_t m = (uint32_t)j, iter = (uint32_t)it + 1u;
                if (kind == 0) {
                    const double u = orc_uniform(seed, m, iter, 0, 0), z = orc_normal(seed, m, iter, 0, 0);
                    a[0] = abc_update(s[0], xpx[j], &la[0], &lb[0], &ldl[0], ie, var_effect[0], prior[0], u, z);
                } else if (kind == 1) {
                    const double u = orc_uniform(seed, m, iter, 0, 0), z = orc_normal(seed, m, iter, 0, 0);
                    a[0] = bayesr_update(s[0], xpx[j], &la[0], &lcls, ie, var_effect[0], prior, gamma, u, z);
                } else {
                    float w[ORC_MAXT];
                    for (int k = 0; k < t; ++k) w[k] = s[k] + xpx[j] * la[k];
                    mt1_update(t, w, xpx[j], la, lb, ldl, 1, Rinv, Ginv, prior, seed, m, iter, 0, a);
                }
                if (tid == 0) { pend_j = j; pcls = lcls; for (int k = 0; k < t; ++k) { pa[k] = la[k]; pb[k] = lb[k]; pd[k] = ldl[k]; } }
                for (int k = 0; k < t; ++k) if (a[k] != 0.0f) axpy_f32(a[k], x + lo, r + k * ld_r + lo, hi - lo);
            }
        team_wait(&bar, T, &sense);
finished:
        if (tid == 0 && pend_j >= 0) {
            for (int k = 0; k < t; ++k) { alpha[k * p + pend_j] = pa[k]; if (kind != 1) { beta[k * p + pend_j] = pb[k]; ((float*)delta)[k * p + pend_j] = pd[k]; } }
            if (kind == 1) ((int32_t*)delta)[pend_j] = pcls;
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (markers_done) *markers_done = done;
    free(partial);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
