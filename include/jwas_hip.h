/*
 * jwas_hip.h -- C ABI of libjwas_hip.so: the MI355X (gfx950) marker-effect Gibbs sweep.
 *
 * Drop-in boundary for ONE path of reworkhow/JWAS.jl (v2.3.6): the single-site marker-effect
 * update loop inside runMCMC and the genotype storage/access layer under it.  A host (the bundled
 * Python host in jwas.jl_amd/, or a Julia `ccall` shim -- see INTEGRATION.md) keeps
 * get_genotypes()/build_model()/runMCMC(); everything else (fixed effects, pedigree terms, variance
 * components, pi draws, output files) stays on the host and consumes only the O(n)+O(p)-reducible
 * statistics returned by jwas_hip_sweep().
 *
 * Reference interfaces replaced (paths relative to /root/reference/src/1.JWAS/src/):
 *   jwas_hip_load_dense_f32*   Genotypes.genotypes + GibbsMats column views
 *                              (types.jl:98-165, markers/tools4genotypes.jl:8-10,237-258)
 *   jwas_hip_setup_blocks      GibbsMats x'x and block Grams, get_column_blocks_ref
 *                              (markers/tools4genotypes.jl:28-36,80-88,259-267)
 *   jwas_hip_set/get_state     Genotypes.alpha/beta/delta (MCMC/MCMC_BayesianAlphabet.jl:85-117)
 *   jwas_hip_set/get_residual  the shared ycorr vector (MCMC/MCMC_BayesianAlphabet.jl:131-157)
 *   jwas_hip_residual_sub_xalpha   ycorr -= X*alpha0 (MCMC/MCMC_BayesianAlphabet.jl:137-146)
 *   jwas_hip_sweep             BayesABC!  (markers/BayesianAlphabet/BayesABC.jl:60-80,118-188)
 *                              BayesR!    (markers/BayesianAlphabet/BayesR.jl:45-97,111-193)
 *                              MTBayesABC! sampler I (markers/BayesianAlphabet/MTBayesABC.jl:57-127,243-333)
 *                              + the post-sweep reductions consumed by Pi.jl:7-17,20-42 and
 *                              variance_components.jl:60-112,151-189
 *   jwas_hip_accumulate        output_posterior_mean_variance, marker part (output.jl:568-577)
 *   jwas_hip_mul_alpha         getEBV's X*alpha (output.jl:281-306)
 *   jwas_hip_load_output_dense_f32 / jwas_hip_mul_alpha_output
 *                              Mi.output_genotypes (markers/tools4genotypes.jl:290-296) and getEBV over mme.output_ID
 *                              (output.jl:281-306; default output_ID = all genotyped individuals,
 *                              input_data_validation.jl:150-154)
 *   jwas_hip_window_sums       window genomic variances of a marker-effect sample (src/3.GWAS/src/GWAS.jl:152-165)
 *
 * Conventions: every entry point returns 0 on success and a negative JWAS_HIP_E* code on failure
 * (no exceptions cross the boundary; jwas_hip_last_error() returns the message -- the analogue of
 * the reference's error(...) strings).  All pointers are plain host pointers unless a parameter
 * name ends in _dev.  The context owns all device memory it allocates; host arrays are copied
 * during the call and never retained.  One host thread per context; calls are synchronous unless
 * stated.  The library never falls back to a CPU path.
 */
#ifndef JWAS_HIP_H
#define JWAS_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jwas_hip_ctx jwas_hip_ctx;

enum {
    JWAS_HIP_OK        =  0,
    JWAS_HIP_EINVAL    = -1,   /* invalid argument (message says which; mirrors error(...) text) */
    JWAS_HIP_EHIP      = -2,   /* a HIP runtime call failed                                     */
    JWAS_HIP_ESTATE    = -3,   /* call order violated (e.g. sweep before setup_blocks)          */
    JWAS_HIP_EUNSUP    = -4,   /* combination not supported by the device path                  */
    JWAS_HIP_ENOMEM    = -5    /* memory guard / allocation failure                             */
};

/* Marker-effect samplers (Mi.method / multi_trait_sampler in the reference). */
enum {
    JWAS_HIP_BAYESC    = 0,    /* single-trait BayesC: one shared effect variance               */
    JWAS_HIP_BAYESB    = 1,    /* single-trait BayesB (BayesA = BayesB with pi = 0): per-marker  */
    JWAS_HIP_BAYESR    = 2,    /* single-trait BayesR, 4-class mixture                          */
    JWAS_HIP_MTBAYESC1 = 3,    /* multi-trait BayesC, Gibbs sampler I  (MTBayesABC.jl:57-127)    */
    JWAS_HIP_MTBAYESC2 = 4,    /* multi-trait BayesC, Gibbs sampler II (MTBayesABC.jl:129-210):  */
                               /* joint indicator state; candidate states in bitmask order       */
    JWAS_HIP_MEGABAYESC = 5,   /* megaBayesABC! (BayesABC.jl:1-8, G.constraint = true): t         */
                               /* independent single-trait BayesC chains sharing one pass over X */
    JWAS_HIP_MTBAYESB1 = 6,    /* multi-trait BayesA/B, Gibbs sampler I with ONE t x t effect      */
                               /* covariance PER MARKER (locus_effect_variances[marker],           */
                               /* MTBayesABC.jl:66,86-90; variance_components.jl:181-186):          */
                               /* jwas_sweep_params.var_effect_matrix, or drawn on the device by    */
                               /* jwas_hip_sample_marker_covariances                                 */
    JWAS_HIP_MTBAYESB2 = 7,    /* multi-trait BayesA/B under Gibbs sampler II (a Pi that lists fewer */
                               /* than 2^t states with multi_trait_sampler = :auto, MTBayesABC.jl:   */
                               /* 20-25,129-210 with Ginv[marker]); covariances as for MTBAYESB1     */
    JWAS_HIP_MEGABAYESB = 8    /* megaBayesABC! with BayesA/B (BayesABC.jl:1-8, G.constraint = true): */
                               /* t independent single-trait chains, trait k of marker j with        */
                               /* variance var_effect_matrix[j][k][k] (off-diagonals ignored);       */
                               /* jwas_hip_sample_marker_covariances then draws the diagonal only    */
};

/* Genotype storage kinds (Genotypes.storage_mode in the reference, types.jl:149-150). */
enum {
    JWAS_HIP_STORAGE_DENSE_F32  = 0,   /* storage=:dense, Float32 matrix                                 */
    JWAS_HIP_STORAGE_PACKED2BIT = 1    /* storage=:stream's 2-bit packed marker-major payload (.jgb2)   */
};

/* Gram precompute modes for jwas_hip_setup_blocks. */
enum {
    JWAS_HIP_GRAM_F64  = 0,    /* fp64-accumulated (bit-reproducible vs the CPU oracle); slow    */
    JWAS_HIP_GRAM_MFMA = 1     /* fp32 MFMA (v_mfma_f32_32x32x2_f32), fp64 chunk combine         */
};

#define JWAS_HIP_MAX_TRAITS 4
#define JWAS_HIP_MAX_STATES 16          /* 2^JWAS_HIP_MAX_TRAITS */

/* Parameters of one sweep = one call of BayesABC!/BayesR!/MTBayesABC! in the reference. */
typedef struct jwas_sweep_params {
    int32_t  method;                    /* JWAS_HIP_BAYESC ...                                       */
    int32_t  ntraits;                   /* 1, or t (2..4) for the multi-trait methods                */
    int32_t  nreps;                     /* within-block repetitions: 1 = exact non-block chain;      */
                                        /* <= 0 = block size (reference fast_blocks, BayesABC.jl:153) */
    uint32_t iteration;                 /* MCMC iteration index (enters the RNG counter)             */
    uint64_t seed;                      /* runMCMC(seed=...) (JWAS.jl:239-251)                       */
    uint32_t marker_offset;             /* global index of this context's column 0 (marker shards)   */
    uint32_t independent_blocks;        /* != 0: independent-block sweep (MCMCinfo.independent_blocks,   */
                                        /* BayesABC.jl:190-255): every block starts from the same        */
                                        /* residual snapshot; reconcile r += sum_b X_b*dalpha_b afterwards */
    float    vare[JWAS_HIP_MAX_TRAITS * JWAS_HIP_MAX_TRAITS];        /* residual (co)variance, row-major t x t */
    float    var_effect[JWAS_HIP_MAX_TRAITS * JWAS_HIP_MAX_TRAITS];  /* BayesC: sigma2_alpha; BayesR: sigmaSq; MT: t x t (MEGA: diagonal used) */
    double   pi;                        /* BayesC/B scalar Pr(effect = 0); ignored if pi_vec != NULL */
    double   pi_classes[4];             /* BayesR class priors; ignored if pi_matrix != NULL.        */
                                        /* MEGABAYESC: pi_classes[k] = Pr(effect = 0) of trait k     */
    double   gamma[4];                  /* BayesR class variances (JWAS.jl:12)                       */
    double   log_prior_states[JWAS_HIP_MAX_STATES];  /* MT: log pi(state), state = sum delta_k << k  */
    const float*  var_effect_vec;       /* BayesB: p per-marker variances (host), else NULL          */
    const double* pi_vec;               /* BayesC/B: p per-marker pi (host), else NULL               */
    const double* pi_matrix;            /* BayesR: p x 4 row-major per-marker class priors, else NULL */
    const double* log_prior_states_matrix;  /* MT samplers I/II: p x 2^t row-major per-marker log pi(state) (marker-specific */
                                        /* joint priors: the reference's annotated multi-trait BayesC, MarkerSpecificPiPrior, */
                                        /* MTBayesABC.jl:22-47), else NULL; needs 2 traits and a block size <= 512 */
    const float*  var_effect_matrix;    /* MTBAYESB1: p x t x t row-major per-marker effect covariances (host); inverted on the   */
                                        /* device once per sweep; needs block_size * ntraits <= 2048.  NULL = the covariances     */
                                        /* resident on the device (an earlier sweep's, or jwas_hip_sample_marker_covariances)     */
    /* Float64 contexts (jwas_hip_set_precision(ctx, 64); runMCMC(double_precision=true)) read these instead of vare /     */
    /* var_effect / var_effect_vec -- the same quantities as Float64, as the reference holds them in that mode:           */
    double   vare_f64[JWAS_HIP_MAX_TRAITS * JWAS_HIP_MAX_TRAITS];
    double   var_effect_f64[JWAS_HIP_MAX_TRAITS * JWAS_HIP_MAX_TRAITS];
    const double* var_effect_vec_f64;   /* BayesB: p per-marker variances (host), else NULL                                */
    int32_t  section_solve;             /* != 0: Rule T -- dense chains as triangular solves.  Multi-trait sampler I (shared or     */
                                        /* per-marker covariance, <= 3 traits) on FULL 256-marker blocks: a 64-marker section's     */
                                        /* chain (MTBayesABC.jl:243-333) is evaluated as D = T y with the section's inverse         */
                                        /* T = (I + L)^-1 formed once per sweep on the device, every marker then verified with the  */
                                        /* literal evaluation (MTBayesABC.jl:85-120); a marker that is not in the model for every   */
                                        /* trait before and after is an EXCEPTION -- its literal evaluation replaces its row of the */
                                        /* solution and the rows behind it are corrected, in marker order; sections with many       */
                                        /* exceptions run the sequential chain.  The same conditional means and draws, another      */
                                        /* association (effects within float32 rounding of the sequential chain).  Needs nreps = 1, */
                                        /* no independent_blocks, no marker-specific priors; every other sweep ignores the flag.    */
                                        /* 0 = the sequential chain everywhere (bit-identical to earlier releases).                 */
    int32_t  group_launch;              /* != 0: GROUPED LAUNCHES -- every launch streams the 2 or 4 consecutive blocks that         */
                                        /* jwas_hip_setup_groups prepared and samples the previous group's blocks in order: the     */
                                        /* same chain with the per-launch cost shared by the group; the block right-hand sides are  */
                                        /* assembled from an older residual plus up to three cross-Gram corrections (float32        */
                                        /* rounding of the plain schedule).  Single-trait methods, nreps = 1, uniform blocks, no     */
                                        /* independent_blocks; every other sweep, and a context without jwas_hip_setup_groups,      */
                                        /* ignores the flag.  0 = one block per launch (bit-identical to earlier releases).         */
} jwas_sweep_params;

/* Reductions the host-side conjugate draws need (Pi.jl, variance_components.jl). */
typedef struct jwas_sweep_stats {
    double  sum_delta[JWAS_HIP_MAX_TRAITS];          /* BayesC/B, MT: sum_j delta_jk (MCMC_BayesianAlphabet.jl:309)   */
    double  alpha_ss[JWAS_HIP_MAX_TRAITS * JWAS_HIP_MAX_TRAITS];   /* alpha'alpha (t x t; [0] single trait)           */
    double  beta_ss[JWAS_HIP_MAX_TRAITS * JWAS_HIP_MAX_TRAITS];    /* beta'beta  (variance_components.jl:175-177)     */
    double  resid_ss[JWAS_HIP_MAX_TRAITS * JWAS_HIP_MAX_TRAITS];   /* r_i'r_j    (variance_components.jl:60-66,82-98) */
    double  resid_sum[JWAS_HIP_MAX_TRAITS];          /* sum_i r_ik (intercept-only location update)                   */
    double  class_counts[4];                         /* BayesR: markers per class (Pi.jl:11-17)                       */
    double  bayesr_ssq;                              /* BayesR: sum_{delta>1} alpha^2/gamma_delta                     */
    double  bayesr_nnz;                              /* BayesR: #{delta > 1}                                          */
    double  state_counts[JWAS_HIP_MAX_STATES];       /* MT: markers per joint state (Pi.jl:20-42)                     */
    double  n_events;                                /* markers whose effect changed this sweep (diagnostic)          */
    double  sweep_ms;                                /* device time of the sweep (hipEvent), milliseconds             */
    double  update_kernel_ms;                        /* sum of the sampled k_block_step event intervals (ms)          */
    double  update_kernel_samples;                   /* number of launches timed (see jwas_hip_set_kernel_timing)     */
    double  update_kernel_bytes;                     /* algorithmic bytes (4*n*b) of the timed launches               */
    double  event_overhead_ms;                       /* HIP-event interval around an EMPTY launch (dispatch gap), ms  */
} jwas_sweep_stats;

/* ---- context ------------------------------------------------------------------------------- */
int  jwas_hip_create(int device, jwas_hip_ctx** out);
void jwas_hip_destroy(jwas_hip_ctx* ctx);
const char* jwas_hip_last_error(const jwas_hip_ctx* ctx);   /* ctx may be NULL: create() errors */
/* Launch all work on an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). */
int  jwas_hip_set_stream(jwas_hip_ctx* ctx, void* hip_stream);
int  jwas_hip_device_info(jwas_hip_ctx* ctx, int* n_cu, int64_t* hbm_bytes_total, int64_t* hbm_bytes_free);

/* ---- genotype storage ------------------------------------------------------------------------ */
/* Copy a column-major (marker-major) n x p fp32 matrix from the host; ld_host >= n is the host
 * column stride in elements (Julia Matrix{Float32} / numpy order='F': ld_host = n). */
int  jwas_hip_load_dense_f32(jwas_hip_ctx* ctx, const float* X_host, int64_t n, int64_t p, int64_t ld_host);
/* Allocate an uninitialised n x p device matrix (filled later by jwas_hip_synth_genotypes). */
int  jwas_hip_alloc_dense_f32(jwas_hip_ctx* ctx, int64_t n, int64_t p);
/* ---- 2-bit packed storage: the reference's Packed2BitBackend kept packed in HBM (16x fewer bytes than fp32) ----
 * Layout (streaming_genotypes.jl:364-367,622-627): marker-major, marker j at bytes [j*stride, (j+1)*stride),
 * stride >= cld(n,4); individual i in byte i>>2 at bit shift (i&3)<<1; codes 0/1/2 = genotype, 3 = missing.
 * Every kernel decodes on the fly exactly as decode_marker! (:978-1002): v = code==3 ? mean_j : Float32(code),
 * x = centered ? v - mean_j : v, so all results equal those of the dense path on the decoded matrix.
 * jwas_hip_load_jgb2 replaces load_streaming_backend (:884-971): `path` is the prefix, or <prefix>.meta / .jgb2;
 * it reads the tab-separated manifest (nObs, nMarkers, stride_bytes, centered, data_path, mean_path) and the
 * .mean.f32 sidecar.  x'x is recomputed on the device; a host that wants the .xpRinvx.f32 sidecar values instead
 * passes them to jwas_hip_set_xpx after jwas_hip_setup_blocks. */
int  jwas_hip_load_jgb2(jwas_hip_ctx* ctx, const char* path);
int  jwas_hip_load_packed2bit(jwas_hip_ctx* ctx, const uint8_t* payload, int64_t n, int64_t p, int64_t stride_bytes,
                              const float* marker_means, int32_t centered);
int  jwas_hip_alloc_packed2bit(jwas_hip_ctx* ctx, int64_t n, int64_t p, int32_t centered);   /* + jwas_hip_synth_genotypes */
int  jwas_hip_storage_info(jwas_hip_ctx* ctx, int32_t* kind, int64_t* n, int64_t* p, int64_t* bytes);
/* Device row stride (elements) of the padded marker-major layout, and its base device pointer. */
int  jwas_hip_dense_layout(jwas_hip_ctx* ctx, int64_t* n, int64_t* p, int64_t* ld_dev, void** X_dev);
/* Copy columns [j0, j0+count) back to the host (column-major, ld = n). */
int  jwas_hip_get_columns(jwas_hip_ctx* ctx, int64_t j0, int64_t count, float* out_host);
/* Overwrite columns [j0, j0+count) of an allocated dense matrix from the host (column-major, column stride ld_host >= n):
 * a matrix that is produced in marker chunks -- impute_genotypes (single_step/SSBR.jl:112-135: 1000 markers at a time) --
 * goes to HBM chunk by chunk and never exists on the host as a whole.  Resident block configurations are dropped. */
int  jwas_hip_set_columns(jwas_hip_ctx* ctx, int64_t j0, int64_t count, const float* cols_host, int64_t ld_host);
/* Memory guard (tools4genotypes.jl:99-235 analogue for HBM): bytes the dense path needs (2 or 3 traits on 256-marker
 * blocks: incl. the per-sweep section inverses of jwas_sweep_params.section_solve, p * 64 * ntraits^2 * 4 bytes). */
int64_t jwas_hip_estimate_bytes(int64_t n, int64_t p, int32_t ntraits, int32_t block_size);
/* Same for a given storage kind (estimate_marker_memory(...; storage_mode), tools4genotypes.jl:99-235). */
int64_t jwas_hip_estimate_bytes_storage(int64_t n, int64_t p, int32_t ntraits, int32_t block_size, int32_t storage);

/* Benchmark / test data generator (benchmarks/bayesr_parity_common.jl:34-41 shape): allele
 * frequency f_j ~ U(0.1,0.4), x_ij = Bernoulli(f_j)+Bernoulli(f_j), optionally centred by the
 * exact column mean.  kind 1 = X ~ U[0,1) (benchmarks/jwas_nonblock_benchmark.jl:34-51).
 * marker_offset = global index of column 0, so marker shards of one matrix can be generated
 * independently on different GPUs. */
int  jwas_hip_synth_genotypes(jwas_hip_ctx* ctx, uint64_t seed, int32_t kind, int32_t center, int64_t marker_offset);
/* Single-step shaped input (the dense real-valued matrix impute_genotypes hands to the sweep, single_step/SSBR.jl:83-142):
 * rows [0, n_genotyped) are 0/1/2 genotypes as above, rows [n_genotyped, n) are "imputed" -- the average of two
 * genotyped rows drawn per ROW (the same linear map for every marker, as A_ng A_gg^-1 M_g is).  Dense storage only. */
int  jwas_hip_synth_single_step(jwas_hip_ctx* ctx, uint64_t seed, int64_t n_genotyped, int32_t center, int64_t marker_offset);

/* Residual weights R^-1 (n floats; NULL = unit weights): mme.invweights = 1 ./ df.weights (build_MME.jl:305-310).
 * With non-unit weights x'x becomes x'R^-1 x (getXpRinvX, tools4genotypes.jl:28-31), the Grams X_b'R^-1 X_b (:263-266),
 * the block RHS X_b'R^-1 r (block_rhs!, :59-78) and the residual statistics r'R^-1 r / 1'R^-1 r
 * (variance_components.jl:82-98); the residual update itself uses the plain column (BayesABC.jl:48).
 * Call after loading genotypes and BEFORE jwas_hip_setup_blocks (resident block configurations are dropped). */
int  jwas_hip_set_weights(jwas_hip_ctx* ctx, const float* rinv_n);
/* The same for a Float64 context (jwas_hip_set_precision(ctx, 64)): the weights as the Float64 values the reference holds under
 * double_precision = true (invweights, build_MME.jl:310) -- jwas_hip_set_weights on such a context widens Float32 values.
 * A Float32 context rejects it (JWAS_HIP_ESTATE). */
int  jwas_hip_set_weights_f64(jwas_hip_ctx* ctx, const double* rinv_n);

/* ---- precompute: x'x and block Grams ----------------------------------------------------------- */
/* block_size in {64,128,256,512,1024}; markers are processed in consecutive blocks of this size. */
int  jwas_hip_setup_blocks(jwas_hip_ctx* ctx, int32_t block_size, int32_t gram_mode);
int  jwas_hip_get_xpx(jwas_hip_ctx* ctx, float* out_p);
/* Overwrite x'x (e.g. with the .xpRinvx.f32 sidecar of a streaming backend, streaming_genotypes.jl:283-285). */
int  jwas_hip_set_xpx(jwas_hip_ctx* ctx, const float* in_p);
int  jwas_hip_get_gram(jwas_hip_ctx* ctx, int64_t block, float* out_bxb);        /* row-major b x b */
int  jwas_hip_set_gram(jwas_hip_ctx* ctx, int64_t block, const float* in_bxb);
int  jwas_hip_num_blocks(jwas_hip_ctx* ctx, int64_t* nblocks, int32_t* block_size);
/* Overwrite the cross-Gram X_{block-1}' X_block (row-major b_{block-1} x b_block; block >= 1) -- with jwas_hip_set_xpx /
 * jwas_hip_set_gram this lets a host (or the test oracle) supply every precomputed inner product. */
int  jwas_hip_set_cross_gram(jwas_hip_ctx* ctx, int64_t block, const float* in);
/* Geometry of the streaming (update) role chosen for this matrix: slices (256 rows, one wave each) per row group, row
 * groups, column groups.  The order in which a block's x'r is summed follows from it (fp64; slices of a row group in
 * order, then the row groups in order): the test oracle reproduces it to compare chains bit for bit. */
int  jwas_hip_update_geometry(jwas_hip_ctx* ctx, int32_t* slices_per_row_group, int32_t* row_groups, int32_t* column_groups);
/* Several block sizes can be resident (Grams + cross-Grams each: 8*p*block_size bytes).  The random draws do not depend
 * on the block size, so a host may switch between sweeps: large blocks amortise the per-launch cost when few markers
 * change per sweep, smaller ones keep the serial within-block chain short when many do (the reference's block size is
 * a free tuning knob too: fast_blocks=<number>, JWAS.jl:293-316).  get/set_gram and num_blocks act on the selected size. */
int  jwas_hip_add_block_size(jwas_hip_ctx* ctx, int32_t block_size, int32_t gram_mode);
int  jwas_hip_select_block_size(jwas_hip_ctx* ctx, int32_t block_size);
/* Grouped launches (jwas_sweep_params.group_launch) for the SELECTED block size: blocks_per_launch = 2 or 4 consecutive blocks
 * per launch of the step kernel (block_size * blocks_per_launch <= 4096), 0 = free the buffers again.  Builds the cross-Grams of
 * consecutive pairs (2 blocks per launch: 8 * p * block_size bytes) or of the odd pairs and the fours (4 blocks per launch:
 * 4 + 16 = 20 * p * block_size bytes) -- once; needs uniform blocks.  The
 * reference has no counterpart: like the block size it is a schedule knob of the device (BayesABC.jl:145-187 is the chain). */
int  jwas_hip_setup_groups(jwas_hip_ctx* ctx, int32_t blocks_per_launch, int32_t gram_mode);
/* Explicit, possibly non-uniform block partition: fast_blocks = a vector of block starts (JWAS.jl:298-304,
 * validate_fast_block_starts JWAS.jl:73-79).  starts: nblocks 0-based first markers, starts[0] = 0, strictly increasing;
 * block k = [starts[k], starts[k+1]) (the last one ends at p); at most 1024 markers per block and 32768 blocks.  With
 * jwas_sweep_params.nreps <= 0 every block runs its own size as repetition count (BayesABC.jl:153).  Replaces any resident
 * block configuration; independent_blocks and a second resident block size are not available on it. */
int  jwas_hip_setup_blocks_explicit(jwas_hip_ctx* ctx, const int64_t* starts, int64_t nblocks, int32_t gram_mode);

/* ---- chain state ---------------------------------------------------------------------------- */
/* Declare the sampler so state buffers can be sized: method + ntraits (delta is int32 classes for
 * BayesR, float 0/1 otherwise -- MCMC_BayesianAlphabet.jl:86,119). */
int  jwas_hip_init_state(jwas_hip_ctx* ctx, int32_t method, int32_t ntraits);
/* alpha/beta: p floats; delta: p floats (0/1) or p int32 (BayesR classes).  NULL = leave as is. */
int  jwas_hip_set_state(jwas_hip_ctx* ctx, int32_t trait, const float* alpha, const float* beta, const void* delta);
int  jwas_hip_get_state(jwas_hip_ctx* ctx, int32_t trait, float* alpha, float* beta, void* delta);
/* residual of trait k: n floats */
int  jwas_hip_set_residual(jwas_hip_ctx* ctx, int32_t trait, const float* r_host);
int  jwas_hip_get_residual(jwas_hip_ctx* ctx, int32_t trait, float* r_host);
/* Device pointer / stride of the residual block (t vectors of ld_dev floats) for in-place
 * collectives on it (marker-shard reconcile). */
int  jwas_hip_residual_dev(jwas_hip_ctx* ctx, void** r_dev, int64_t* ld_dev);
/* Device-to-device copies of residual k (n floats) from / to a caller-owned device buffer (e.g. a
 * torch tensor's data_ptr()), ordered on the context's stream -- used around the per-sweep RCCL
 * all-reduce of the residual delta. */
int  jwas_hip_residual_to_dev(jwas_hip_ctx* ctx, int32_t trait, void* dst_dev);
int  jwas_hip_residual_from_dev(jwas_hip_ctx* ctx, int32_t trait, const void* src_dev);
/* r_k[i] = fl32(fl64(r_k[i]) + shift) for the n individuals: the residual correction of a location parameter whose design
 * column is all ones (the intercept step of the host's single-site Gibbs pass, solver.jl:143-162) without a host copy of
 * the residual -- with jwas_sweep_stats.resid_sum the intercept update needs no O(n) host traffic at all.  Asynchronous
 * (ordered on the context's stream). */
int  jwas_hip_residual_add_scalar(jwas_hip_ctx* ctx, int32_t trait, double shift);
/* r_k -= X * alpha_k for the current device alpha (initial ycorr; sequential fmaf in marker order). */
int  jwas_hip_residual_sub_xalpha(jwas_hip_ctx* ctx, int32_t trait);
/* out = X * alpha_k (n floats, fp64-accumulated). */
int  jwas_hip_mul_alpha(jwas_hip_ctx* ctx, int32_t trait, float* out_host);
/* The nonzero effects of trait k as (marker index, value) lists in marker order, compacted on the device: one saved
 * marker-effect sample as a sparse record instead of the reference's dense text row of p values (output.jl:443-526).
 * idx / val: caller arrays of `capacity` entries (capacity >= p is always enough); *nnz = number of nonzero effects. */
int  jwas_hip_get_alpha_sparse(jwas_hip_ctx* ctx, int32_t trait, int64_t capacity, int32_t* idx, float* val, int64_t* nnz);
/* Rows for which EBVs are reported when they are not exactly the training rows (individuals without records, a
 * user's outputEBV(model, IDs) list): X_out is n_out x p marker-major fp32 with leading dimension ld_host, processed
 * (imputed / centred with the training column means) like the training matrix; copied. */
int  jwas_hip_load_output_dense_f32(jwas_hip_ctx* ctx, const float* X_out_host, int64_t n_out, int64_t p, int64_t ld_host);
/* out = X_out * alpha_k (n_out floats, fp64-accumulated). */
int  jwas_hip_mul_alpha_output(jwas_hip_ctx* ctx, int32_t trait, float* out_host);
/* Window genomic variances of ONE saved marker-effect sample -- the inner loop of the reference's window-based GWAS
 * (src/3.GWAS/src/GWAS.jl:152-165: genVar = var(X*alpha); per window var(X[:, w]*alpha[w])).  Window w owns the nonzero
 * effects idx[wptr[w] .. wptr[w+1]) (marker indices, any order the caller wants summed in) with values val[..];
 * windows may overlap (sliding windows) and window 0 is typically "all markers".  out_sum[w] = sum_i BV_w[i],
 * out_ss[w] = sum_i BV_w[i]^2 (fp64, deterministic) over the training individuals, or over the output rows of
 * jwas_hip_load_output_dense_f32 when use_output_rows != 0 (the reference uses Mi.output_genotypes). */
int  jwas_hip_window_sums(jwas_hip_ctx* ctx, int32_t use_output_rows, int32_t nwin, const int32_t* wptr, const int32_t* idx,
                          const float* val, double* out_sum, double* out_ss);
/* Two effect vectors over the same markers (two traits' samples of the same iteration; idx = union of their nonzero
 * effects): additionally out_cross[w] = sum_i BV1_w[i] * BV2_w[i] -- the window genetic covariance / correlation of
 * src/3.GWAS/src/GWAS.jl:199-217. */
int  jwas_hip_window_sums2(jwas_hip_ctx* ctx, int32_t use_output_rows, int32_t nwin, const int32_t* wptr, const int32_t* idx,
                           const float* val1, const float* val2, double* out_sum1, double* out_ss1, double* out_sum2,
                           double* out_ss2, double* out_cross);

/* ---- the sweep ------------------------------------------------------------------------------ */
/* Time every `stride`-th k_block_step launch of subsequent sweeps with HIP events on the
 * sweep's stream (0 = off); the sums come back in jwas_sweep_stats.update_kernel_*. */
int  jwas_hip_set_kernel_timing(jwas_hip_ctx* ctx, int32_t stride);
int  jwas_hip_sweep(jwas_hip_ctx* ctx, const jwas_sweep_params* params, jwas_sweep_stats* stats);
/* Diagnostics of the LAST sweep (no reference counterpart; what JWAS_HIP_DEBUG_PHASES prints): the sampler's counters, n <= 32
 * values -- [0] effect changes, [1] Gram rows fetched on demand, [2..6] phase cycles, [7] rounds / sections walked again,
 * [16] / [17] compact-chain blocks tried / fallen back, or (section_solve) sections solved / fallen back to the walk,
 * [23] (section_solve) exceptions taken inside the solved sections, [24] hand-over words that never arrived (the sweep fails),
 * [29] ping-pong blocks staged a second time, [30] / [31] multi-trait skip and verify: blocks in which the serial wave took the
 * chain over again / 64-marker sub-blocks evaluated by a helper wave. */
int  jwas_hip_last_sweep_counters(jwas_hip_ctx* ctx, uint64_t* out, int32_t n);

/* ---- marker shards over the GPUs of one node (one context per GPU / process) -------------------------------
 * The single-site chain is sequential in the marker index; what the reference ships for parallel blocks is
 * independent_blocks=true (BayesABC.jl:190-255): every block starts from the same residual snapshot and the residual is
 * reconciled once per sweep by r += sum_b X_b * (alpha_old_b - alpha_new_b) (BayesABC.jl:251-253).  These entry points
 * are that mode with one "block" per GPU: the context holds the rank's marker columns (marker_offset = global index of
 * its first column) and a replicated residual.
 *   jwas_hip_comm_unique_id   rank 0 creates the 128-byte RCCL id (ncclGetUniqueId) and hands it to the other ranks by
 *                             whatever means the host has (a file, a socket, MPI, torch.distributed ...);
 *   jwas_hip_comm_init        every rank joins (ncclCommInitRank on the context's device; collective);
 *   jwas_hip_sweep_sharded    jwas_hip_sweep on the own markers from the snapshot, then ON THE DEVICE, on the context's
 *                             stream: delta r = fl64(r_local) - fl64(r_snapshot) and the packed marker statistics in one
 *                             buffer, ONE ncclAllReduce(sum, fp64) over xGMI, r = fl32(r_snapshot + sum of delta r).
 *                             On return every rank holds the same residual (jwas_hip_get_residual) and `stats` holds the
 *                             ALL-RANK sums (sum_delta, alpha_ss, class / state counts, n_events ...) and the residual
 *                             statistics of the reconciled residual.  Approximate unless X_g'X_h = 0 between shards
 *                             (docs/src/manual/block_bayesc.md:95-134); world = 1 is the exact chain.
 * librccl.so is loaded on first use (dlopen); a single-GPU host never needs it. */
int  jwas_hip_comm_unique_id(void* id_out_128_bytes);
int  jwas_hip_comm_init(jwas_hip_ctx* ctx, const void* unique_id_128_bytes, int32_t rank, int32_t world);
int  jwas_hip_comm_destroy(jwas_hip_ctx* ctx);
/* Rank and size of the attached communicator as the TRANSPORT reports them (ncclCommUserRank / ncclCommCount): what a
 * host must report as its GPU count.  Without a communicator: rank 0 of 1. */
int  jwas_hip_comm_info(jwas_hip_ctx* ctx, int32_t* rank, int32_t* world);
int  jwas_hip_sweep_sharded(jwas_hip_ctx* ctx, const jwas_sweep_params* params, jwas_sweep_stats* stats);
/* ---- exact ROW shards (SURVEY 8e "exact alternative"): every rank holds a slice of the individuals and ALL markers.
 * jwas_hip_comm_row_shards(ctx, 1) after jwas_hip_comm_init and BEFORE jwas_hip_setup_blocks: x'x, the block Grams and the
 * cross-Grams are summed over the ranks at setup, every block's partial right-hand side X_b'r is summed over the ranks
 * (one small all-reduce per block launch, on the context's stream) before its sampler runs -- replicated, on identical
 * inputs, so every rank holds the same effects and its own slice of the residual.  jwas_hip_sweep then IS the exact chain
 * of the pooled data (the sums are formed in a different order than on one GPU, nothing else); r'r / sum r come back summed
 * over the ranks.  Needs the same number of 256-row groups on every rank (pad with zero rows) and the same markers.
 * jwas_hip_comm_init_loopback: test transport (both sharding modes: jwas_hip_sweep_sharded and the row shards) -- the ranks are contexts of ONE process driven by different host threads,
 * the exchange goes through host memory (slot 0..3 = one group of ranks). */
int  jwas_hip_comm_row_shards(jwas_hip_ctx* ctx, int32_t enable);
int  jwas_hip_comm_init_loopback(jwas_hip_ctx* ctx, int32_t slot, int32_t rank, int32_t world);

/* ---- multi-trait BayesA/B: the per-marker effect covariances on the device ------------------------------------------------
 * The reference redraws every marker's t x t effect covariance each iteration, G_j ~ InverseWishart(df + 1, scale + b_j b_j')
 * (sample_variance(data, 1, df, scale) per marker, variance_components.jl:181-186): O(p) small matrix draws that would
 * otherwise make the host the bottleneck.  jwas_hip_sample_marker_covariances draws all p of them on the device from the
 * CURRENT beta (Bartlett's decomposition on the counter RNG, keyed by (seed, iteration, global marker)); `df` is the
 * inverse-Wishart's degrees of freedom as passed to the distribution (the reference's df + 1), `scale` its t x t row-major
 * scale matrix.  The next jwas_hip_sweep with jwas_sweep_params.var_effect_matrix == NULL uses them in place.  Asynchronous. */
int  jwas_hip_sample_marker_covariances(jwas_hip_ctx* ctx, double df, const double* scale_txt, uint64_t seed,
                                        uint32_t iteration, uint32_t marker_offset);
int  jwas_hip_get_marker_covariances(jwas_hip_ctx* ctx, float* out_p_t_t);      /* p x t x t row-major */

/* ---- Float64 mode: runMCMC(double_precision=true) (JWAS.jl:349-366; genotypes read as Float64, readgenotypes.jl:298,345) ----
 * In that mode the reference holds EVERYTHING as Float64 -- genotypes, ycorr, alpha / beta / delta, x'x -- and its scalar
 * kernels run in Float64.  jwas_hip_set_precision(ctx, 64) BEFORE genotypes are loaded makes the context a Float64 context:
 * the data entry points below replace their Float32 namesakes (same meaning, double host arrays; delta is double 0/1, or
 * int32 classes for BayesR); jwas_hip_setup_blocks (any block size 1 .. 1024), jwas_hip_setup_blocks_explicit (blocks of at
 * most 1024 markers), jwas_hip_set_weights (the Float32 values, widened), jwas_hip_init_state, jwas_hip_sweep (reading
 * jwas_sweep_params.vare_f64 / var_effect_f64 / var_effect_vec_f64; nreps and independent_blocks as in a Float32 context),
 * jwas_hip_residual_sub_xalpha, jwas_hip_accumulate and jwas_hip_num_blocks are shared.  Samplers: single-trait BayesA/B/C,
 * BayesR, multi-trait sampler I; dense storage; block size x traits <= 2048.  Everything else of the Float32 surface (packed
 * storage, a second block size, output rows, window sums, shards, samplers II / constrained / per-marker covariances)
 * returns JWAS_HIP_EUNSUP on a Float64 context. */
int  jwas_hip_set_precision(jwas_hip_ctx* ctx, int32_t bits);                     /* 32 (default) or 64 */
int  jwas_hip_load_dense_f64(jwas_hip_ctx* ctx, const double* X_host, int64_t n, int64_t p, int64_t ld_host);
int  jwas_hip_get_xpx_f64(jwas_hip_ctx* ctx, double* out_p);
int  jwas_hip_set_state_f64(jwas_hip_ctx* ctx, int32_t trait, const double* alpha, const double* beta, const void* delta);
int  jwas_hip_get_state_f64(jwas_hip_ctx* ctx, int32_t trait, double* alpha, double* beta, void* delta);
int  jwas_hip_set_residual_f64(jwas_hip_ctx* ctx, int32_t trait, const double* r_host);
int  jwas_hip_get_residual_f64(jwas_hip_ctx* ctx, int32_t trait, double* r_host);
int  jwas_hip_mul_alpha_f64(jwas_hip_ctx* ctx, int32_t trait, double* out_host);
int  jwas_hip_get_posterior_f64(jwas_hip_ctx* ctx, int32_t trait, double* mean_alpha, double* mean_alpha2, double* mean_delta);

/* ---- posterior accumulators (output.jl:568-577) ---------------------------------------------- */
int  jwas_hip_accumulate(jwas_hip_ctx* ctx, double nsamples);
int  jwas_hip_get_posterior(jwas_hip_ctx* ctx, int32_t trait, float* mean_alpha, float* mean_alpha2, float* mean_delta);

#ifdef __cplusplus
}
#endif
#endif
