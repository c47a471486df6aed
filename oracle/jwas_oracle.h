/*
 * jwas_oracle.h -- CPU ORACLE for the marker-effect Gibbs sweep.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's (reworkhow/JWAS.jl v2.3.6) single-site
 * marker samplers.  It exists so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg can check / time the HIP path against an independent implementation.
 * Nothing in the shipped package (jwas.jl_amd/) may include, link or call this file.
 *
 * PARITY STATUS: "parity unpinned" for sampler OUTPUT VALUES -- the reference ships no golden
 * vectors for sampler output and Julia is not installed here, so the chain on a given seed
 * cannot be compared with a Julia run.  What IS pinned against the reference's own tests are
 * the deterministic / closed-form items (tests/test_oracle_kat.py): BayesR sufficient
 * statistics, bayesr_block_nreps schedule, degenerate-prior class KAT, genetic2marker,
 * 2-bit codec tables, the one-marker multi-trait posterior, memory formulas.  Statistically the
 * whole stack is pinned against the reference's OWN published output: its 5-fold cross-validation
 * benchmark on its packaged simulated_annotations data (held-out cor(y, EBV) of eight method
 * families reproduced within 0.01), and against the exact state posterior of
 * small models by enumeration (tests/test_gpu_statistical.py).
 *
 * Reference files restated (all under /root/reference/src/1.JWAS/src/):
 *   markers/BayesianAlphabet/BayesABC.jl:24-80    bayesabc_update_marker!, BayesABC!
 *   markers/BayesianAlphabet/BayesABC.jl:118-188  BayesABC_block!
 *   markers/BayesianAlphabet/BayesR.jl:1-97       BayesR!
 *   markers/BayesianAlphabet/BayesR.jl:111-193    BayesR_block!
 *   markers/BayesianAlphabet/MTBayesABC.jl:57-127 _MTBayesABC_samplerI!
 *   markers/BayesianAlphabet/MTBayesABC.jl:129-210 _MTBayesABC_samplerII!  (+ block / independent forms :243-646)
 *   markers/BayesianAlphabet/BayesABC.jl:1-8,190-255  megaBayesABC!, BayesABC_block_independent!
 *   markers/tools4genotypes.jl:28-36,59-78,259-267  x'x, block_rhs!, block Grams
 *   variance_components.jl:68-79                  bayesr_sigma_sufficient_statistics
 *   output.jl:568-577                             running posterior means
 *
 * Numeric conventions (documented in DESIGN.md "Arithmetic contract"):
 *   - storage types follow the reference's default Float32 path; the scalar chain promotes to
 *     double exactly where Julia's promotion rules do (pi is Float64, randn()/rand() are Float64);
 *   - every transcendental is evaluated in double and rounded to the type Julia would hold;
 *   - x'r / x'x / Gram inner products: ORC_ACC_F64 accumulates exact fp32 products in double
 *     and rounds once (order-independent to ~1e-16, the mode the HIP path implements);
 *     ORC_ACC_F32 is a plain fp32 accumulation (what an sdot does, order unspecified);
 *   - r += a*x is fmaf(a, x, r) per element (OpenBLAS saxpy kernels are FMA kernels);
 *   - random numbers: counter-based Philox4x32-10 (Salmon et al., SC'11) keyed by the seed and
 *     indexed by (marker, iteration, repetition, slot/trait), so a draw does not depend on
 *     block size, shard or device count.  (The reference uses Julia's task-local Xoshiro256++,
 *     which cannot be reproduced without Julia)
 */
#ifndef JWAS_ORACLE_H
#define JWAS_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_ACC_F64 = 0, ORC_ACC_F32 = 1, ORC_ACC_DEVICE = 2 };
/* ORC_ACC_DEVICE: x'r summed in the device's association order (see jwas_oracle.c dot_device_order); spg = the device
 * context's slices per row group. */
void orc_set_device_order(int spg);
void orc_cross_gram(const float* X, int64_t n, int64_t ld, int64_t jp, int64_t bp, int64_t j0, int64_t b, float* out, int acc);

/* ---- random numbers -------------------------------------------------------------------- */
void   orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* u in (0,1): 52 random bits, (k+0.5)*2^-52.  slot 0 of (marker,iter,rep,trait). */
double orc_uniform(uint64_t seed, uint32_t marker, uint32_t iter, uint32_t rep, uint32_t trait);
/* z ~ N(0,1): Box-Muller on two 52-bit uniforms from slot 1 of (marker,iter,rep,trait). */
double orc_normal(uint64_t seed, uint32_t marker, uint32_t iter, uint32_t rep, uint32_t trait);

/* ---- storage-layer precompute (tools4genotypes.jl:28-36, 259-267) ------------------------ */
/* xpx[j] = x_j'x_j, X column-major n x p with leading dimension ld. */
void orc_xpx(const float* X, int64_t n, int64_t p, int64_t ld, float* xpx, int acc);
/* Gram of columns [j0, j0+b): G (b x b, row-major, full symmetric) = X_b' X_b. */
void orc_gram(const float* X, int64_t n, int64_t ld, int64_t j0, int64_t b, float* G, int acc);
/* out = r - X*alpha  (MCMC_BayesianAlphabet.jl:131-147), sequential fmaf in marker order. */
void orc_residual_minus_xalpha(const float* X, int64_t n, int64_t p, int64_t ld,
                               const float* alpha, float* r);

/* ---- single-trait BayesA/B/C, non-block (BayesABC.jl:60-80) ------------------------------ */
/* var_effects: p values (BayesC: all equal).  pi: p values (probability of a ZERO effect).
 * marker0: global index of column 0 (enters the RNG counter; 0 unless the caller holds a shard).
 * Returns 0, or -1 on invalid arguments. */
int orc_bayesabc_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                       float* r, float* alpha, float* beta, float* delta,
                       float vare, const float* var_effects, const double* pi,
                       uint64_t seed, uint32_t iter, uint32_t marker0, int acc);

/* ---- single-trait BayesA/B/C, exact block form (BayesABC.jl:118-188) --------------------- */
/* block_starts: nblocks 0-based start columns (ascending, first = 0); grams: concatenated
 * b_i x b_i row-major Gram blocks.  nreps <= 0 means nreps = block size (the reference's
 * schedule, BayesABC.jl:153); nreps = 1 is algebraically the non-block chain. */
int orc_bayesabc_block_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                             const int64_t* block_starts, int64_t nblocks, const float* grams,
                             float* r, float* alpha, float* beta, float* delta,
                             float vare, const float* var_effects, const double* pi,
                             int nreps, uint64_t seed, uint32_t iter, uint32_t marker0, int acc);

/* ---- single-trait BayesR (BayesR.jl:45-97) and block form (BayesR.jl:111-193) ------------ */
/* pi: 4 values, or p x 4 row-major when pi_is_matrix != 0.  delta holds classes 1..4 (int32).
 * gamma: 4 doubles.  Returns -1 on invalid priors (BayesR.jl:9-20), -2 if sigma_sq <= 0. */
int orc_bayesr_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                     float* r, float* alpha, int32_t* delta,
                     float vare, float sigma_sq, const double* pi, int pi_is_matrix,
                     const double* gamma, uint64_t seed, uint32_t iter, uint32_t marker0, int acc);
int orc_bayesr_block_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                           const int64_t* block_starts, int64_t nblocks, const float* grams,
                           float* r, float* alpha, int32_t* delta,
                           float vare, float sigma_sq, const double* pi, int pi_is_matrix,
                           const double* gamma, int nreps,
                           uint64_t seed, uint32_t iter, uint32_t marker0, int acc);
/* BayesR.jl:22-25 */
int orc_bayesr_block_nreps(int64_t iter, int64_t burnin, int64_t block_size);
/* variance_components.jl:68-79 */
void orc_bayesr_sigma_suffstats(const float* alpha, const int32_t* delta, int64_t p,
                                const double* gamma, double* ssq, int64_t* nnz);

/* ---- multi-trait BayesC, Gibbs sampler I (MTBayesABC.jl:57-127), non-block and block ------ */
/* t traits (<= 8).  r: t residual vectors of length ld_r each (trait-major).  alpha/beta/delta:
 * t x p trait-major.  vare: t x t row-major.  var_effect: t x t (BayesC: shared by all markers).
 * log_prior: 2^t values indexed by state = sum_k delta_k << k  (GlobalPiPrior) or p x 2^t when
 * prior_is_matrix (MarkerSpecificPiPrior). */
int orc_mtbayesc_I_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                         int t, float* r, int64_t ld_r, float* alpha, float* beta, float* delta,
                         const float* vare, const float* var_effect,
                         const double* log_prior, int prior_is_matrix,
                         uint64_t seed, uint32_t iter, uint32_t marker0, int acc);
int orc_mtbayesc_I_block_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                               const int64_t* block_starts, int64_t nblocks, const float* grams,
                               int t, float* r, int64_t ld_r, float* alpha, float* beta, float* delta,
                               const float* vare, const float* var_effect,
                               const double* log_prior, int prior_is_matrix, int nreps,
                               uint64_t seed, uint32_t iter, uint32_t marker0, int acc);

/* ---- one-block LOOKAHEAD schedule of the exact block chain (what the HIP path runs) ---------- */
/* Same chain as the *_block_sweep functions in exact arithmetic; the block RHS is assembled from the
 * residual that lacks the previous block's exit update, then corrected with the cross-Gram of the
 * previous block's changed markers (see jwas_oracle.c).  Lets the device overlap sampling of block b
 * with streaming block b+1. */
/* Grouped lookahead (the device's grouped launches, jwas_sweep_params.group_launch): m = 2 or 4 blocks per group in the
 * single-trait lookahead sweeps below; anything else = 1 = the one-block lookahead (default). */
void orc_set_lookahead_group(int m);
int orc_bayesabc_lookahead_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                                 const int64_t* block_starts, int64_t nblocks, const float* grams,
                                 float* r, float* alpha, float* beta, float* delta,
                                 float vare, const float* var_effects, const double* pi,
                                 int nreps, uint64_t seed, uint32_t iter, uint32_t marker0, int acc);
int orc_bayesr_lookahead_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                               const int64_t* block_starts, int64_t nblocks, const float* grams,
                               float* r, float* alpha, int32_t* delta,
                               float vare, float sigma_sq, const double* pi, int pi_is_matrix,
                               const double* gamma, int nreps,
                               uint64_t seed, uint32_t iter, uint32_t marker0, int acc);
int orc_mtbayesc_I_lookahead_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                                   const int64_t* block_starts, int64_t nblocks, const float* grams,
                                   int t, float* r, int64_t ld_r, float* alpha, float* beta, float* delta,
                                   const float* vare, const float* var_effect,
                                   const double* log_prior, int prior_is_matrix, int nreps,
                                   uint64_t seed, uint32_t iter, uint32_t marker0, int acc);

/* ---- multi-trait, any sampler kind: 1 = Gibbs sampler I (MTBayesABC.jl:57-127), 2 = sampler II (joint
 * state, MTBayesABC.jl:129-210), 3 = megaBayesABC! (BayesABC.jl:1-8: t independent single-trait BayesC sweeps
 * with vare[k,k], var_effect[k,k]; `log_prior` then holds the per-trait pi (probability of a zero effect),
 * t values).  Same argument meaning as the sampler-I functions; dense, block and lookahead forms. */
#define ORC_MT_SAMPLER_I  1
#define ORC_MT_SAMPLER_II 2
#define ORC_MT_MEGA       3
/* Multi-trait BayesA/B: per-marker effect covariances (p x t x t row-major; MTBayesABC.jl:66,86-90) used by every
 * multi-trait sweep until reset with NULL (harness state; sampler I). */
void orc_set_var_effect_matrix(const float* mat);
/* Rule L of sampler I (see mt1_update): 1 = on (the device's definition; default), 0 = the literal operation order. */
void orc_set_mt_linear_form(int on);
/* Rule D of single-trait BayesA/B/C sweeps under a uniform pi = 0 (see abc_update): 1 = on, 0 = the literal order (default). */
void orc_set_abc_rule_d(int on);
/* Rule T (the device's jwas_sweep_params.section_solve; see mt1_section_solve / abc_section_solve): dense 64-marker sections of
 * the lookahead forms' full blocks as triangular solves.  1 = on, 0 = the sequential chain (default). */
void orc_set_section_solve(int on);
/* The block right-hand sides in the 2-bit packed update role's own order (see dot_xr): codes [p][n], one code 0..3 per byte,
 * of the matrix X the sweeps are called with; NULL = off. */
void orc_set_packed_source(const uint8_t* codes, const float* means, int centered, const float* X, int64_t n, int64_t ld);
void orc_section_solve_counts(int64_t* solved, int64_t* fallbacks, int reset);
int64_t orc_section_solve_exceptions(void);      /* exceptions taken inside the solved sections since the last reset */
/* One InverseWishart(df, scale + b_j b_j') draw per marker (variance_components.jl:181-186; df = the reference's df + 1),
 * Bartlett on the counter RNG: the restatement the device's k_sample_marker_covariances is compared with. */
void orc_sample_marker_covariances(int t, int64_t p, const float* beta, double df, const double* scale,
                                   uint64_t seed, uint32_t iter, uint32_t marker0, float* var_mat);
/* constraint = true: diagonal only, G_kk = (scale_kk + b_jk^2) / chi2(df) (variance_components.jl:112-117) */
void orc_sample_marker_variances_diag(int t, int64_t p, const float* beta, double df, const double* scale,
                                      uint64_t seed, uint32_t iter, uint32_t marker0, float* var_mat);
int orc_mt_sweep(int kind, const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                 int t, float* r, int64_t ld_r, float* alpha, float* beta, float* delta,
                 const float* vare, const float* var_effect, const double* log_prior, int prior_is_matrix,
                 uint64_t seed, uint32_t iter, uint32_t marker0, int acc);
int orc_mt_block_sweep(int kind, const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                       const int64_t* block_starts, int64_t nblocks, const float* grams,
                       int t, float* r, int64_t ld_r, float* alpha, float* beta, float* delta,
                       const float* vare, const float* var_effect, const double* log_prior, int prior_is_matrix,
                       int nreps, uint64_t seed, uint32_t iter, uint32_t marker0, int acc);
int orc_mt_lookahead_sweep(int kind, const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                           const int64_t* block_starts, int64_t nblocks, const float* grams,
                           int t, float* r, int64_t ld_r, float* alpha, float* beta, float* delta,
                           const float* vare, const float* var_effect, const double* log_prior, int prior_is_matrix,
                           int nreps, uint64_t seed, uint32_t iter, uint32_t marker0, int acc);

/* ---- independent blocks (independent_blocks=true; BayesABC.jl:190-255, BayesR.jl:195-273, MTBayesABC.jl:335-440):
 * same arguments as the *_block_sweep functions; every block RHS is formed from the residual as it is at entry,
 * blocks are sampled independently and r += sum_b X_b*(alpha_old_b - alpha_b) is applied at the end in
 * (block, marker) order. */
int orc_bayesabc_indep_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                             const int64_t* block_starts, int64_t nblocks, const float* grams,
                             float* r, float* alpha, float* beta, float* delta,
                             float vare, const float* var_effects, const double* pi,
                             int nreps, uint64_t seed, uint32_t iter, uint32_t marker0, int acc);
int orc_bayesr_indep_sweep(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                           const int64_t* block_starts, int64_t nblocks, const float* grams,
                           float* r, float* alpha, int32_t* delta,
                           float vare, float sigma_sq, const double* pi, int pi_is_matrix,
                           const double* gamma, int nreps,
                           uint64_t seed, uint32_t iter, uint32_t marker0, int acc);
int orc_mt_indep_sweep(int kind, const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                       const int64_t* block_starts, int64_t nblocks, const float* grams,
                       int t, float* r, int64_t ld_r, float* alpha, float* beta, float* delta,
                       const float* vare, const float* var_effect, const double* log_prior, int prior_is_matrix,
                       int nreps, uint64_t seed, uint32_t iter, uint32_t marker0, int acc);

/* ---- residual weights R^-1 (n floats, borrowed; NULL = unit): all inner products become a'R^-1 b ------------ */
void orc_set_weights(const float* rinv);

/* ---- running posterior means (output.jl:568-577) ------------------------------------------ */
/* mean += (x-mean)/k ; mean2 += (x^2-mean2)/k ; freq += (ind-freq)/k, ind = delta (BayesC) or
 * delta>1 (BayesR, delta_is_class != 0). */
void orc_accumulate(const float* alpha, const void* delta, int delta_is_class, int64_t p, double k,
                    float* mean_alpha, float* mean_alpha2, float* mean_delta);

/* ---- 2-bit packed genotype codec (streaming_genotypes.jl:364-367, 978-1027) ---------------- */
/* decode marker j of a .jgb2 payload: stride = ceil(n/4) bytes per marker, individual i in byte
 * i>>2 at bit shift (i&3)<<1; codes 0/1/2 = genotype, 3 = missing -> mean; subtract mean when
 * centered. */
void orc_decode_marker_2bit(const uint8_t* payload, int64_t n, int64_t j, float mean, int centered,
                            float* out);

/* ---- CPU baseline timing helper (bench.py cpu_baseline leg) ------------------------------- */
/* Runs `sweeps` non-block BayesC sweeps (ORC_ACC_F32 dot + axpy, the reference's per-marker
 * operation order) and returns elapsed seconds; nthreads > 1 splits the dot/axpy rows over
 * POSIX threads the way a threaded BLAS would. */
double orc_time_bayesc_sweeps(const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                              float* r, float* alpha, float* beta, float* delta,
                              float vare, float var_effect, double pi,
                              uint64_t seed, int sweeps, int nthreads);

/* The same per-marker order for BayesC (kind 0), BayesR (1) and multi-trait sampler I (2), with the rows of
 * every dot / axpy split over a persistent team of `nthreads` threads meeting at one spin barrier per marker.
 * r: t x ld_r; alpha/beta/delta: t x p (BayesR: delta int32, beta unused); prior: &pi | pi[4] | log_prior[2^t].
 * max_seconds > 0 bounds the run (checked every 64 markers); *markers_done = marker updates completed (sweeps * p if
 * it ran to the end).  Returns elapsed seconds (< 0: bad arguments). */
double orc_time_sweeps_team(int kind, const float* X, int64_t n, int64_t p, int64_t ld, const float* xpx,
                            int t, float* r, int64_t ld_r, float* alpha, float* beta, void* delta,
                            const float* vare, const float* var_effect, const double* prior, const double* gamma,
                            uint64_t seed, int sweeps, int nthreads, double max_seconds, int64_t* markers_done);

/* ---- Float64 mode (runMCMC(double_precision=true)): oracle/jwas_oracle_f64.c -- the reference's scalar kernels with
 * T = Float64 in the literal non-block order (and BayesABC_block! with repetitions); same counter RNG. */
void orc64_xpx(const double* X, int64_t n, int64_t p, int64_t ld, double* out);
int  orc64_bayesabc_sweep(const double* X, int64_t n, int64_t p, int64_t ld, const double* xpx,
                          double* r, double* alpha, double* beta, double* delta,
                          double vare, const double* var_effects, const double* pi,
                          uint64_t seed, uint32_t iter, uint32_t marker0);
int  orc64_bayesabc_block_sweep(const double* X, int64_t n, int64_t p, int64_t ld, const double* xpx, int64_t bs, int nreps,
                                double* r, double* alpha, double* beta, double* delta,
                                double vare, const double* var_effects, const double* pi,
                                uint64_t seed, uint32_t iter, uint32_t marker0);
void orc64_xpx_w(const double* X, int64_t n, int64_t p, int64_t ld, const double* w, double* out);
int  orc64_bayesabc_block_sweep_ex(const double* X, int64_t n, int64_t p, int64_t ld, const double* xpx, const double* w,
                                   const int64_t* starts, int64_t nb, int nreps, int independent,
                                   double* r, double* alpha, double* beta, double* delta,
                                   double vare, const double* var_effects, const double* pi,
                                   uint64_t seed, uint32_t iter, uint32_t marker0);
int  orc64_bayesr_sweep(const double* X, int64_t n, int64_t p, int64_t ld, const double* xpx,
                        double* r, double* alpha, int32_t* delta,
                        double vare, double sigma_sq, const double* pi, int pi_is_matrix, const double* gamma,
                        uint64_t seed, uint32_t iter, uint32_t marker0);
int  orc64_mt1_sweep(int t, const double* X, int64_t n, int64_t p, int64_t ld, const double* xpx,
                     double* r, int64_t ldr, double* alpha, double* beta, double* delta,
                     const double* vare, const double* var_effect, const double* log_prior,
                     uint64_t seed, uint32_t iter, uint32_t marker0);

#ifdef __cplusplus
}
#endif
#endif
