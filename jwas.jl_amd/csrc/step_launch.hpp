// step_launch.hpp -- launchers of the fused step kernel (sweep.hpp), one translation unit per sampler family so that the
// library builds in parallel and a change to one sampler recompiles one file:
//   step_st.hip    single-trait BayesA/B/C, BayesR          step_mtc1.hip  multi-trait sampler I, one shared covariance
//   step_mtb1.hip  sampler I, a covariance per marker       step_mt2.hip   sampler II (shared / per-marker covariance)
//   step_mega.hip  megaBayesABC (constraint = true)
// jwas_hip.hip (context, C ABI, every other kernel) calls these; step_launch_impl.hpp holds the shared template code.
#pragma once
#include "sweep.hpp"

namespace jw {

// What a launcher needs from the context (plain values: no context type crosses the translation units).
struct StepLaunch {
    int device, block_size, nrg;
    int64_t nblocks, p;
    hipStream_t stream;
    const int64_t* d_starts;      // explicit partition on the device, or NULL
    Events* ev_all;               // independent blocks: per-block change lists
    bool packed;                  // genotype storage: pc (2-bit packed) or dc (dense fp32)
    DenseCols dc;
    PackedCols pc;
};

// launch k of the lookahead pipeline: sampler(block k-1) || update/partial(block k)
hipError_t launch_step_st(const StepLaunch& L, int method, const UpdateArgs& U, const SamplerArgs& S, int do_sample, bool dense);
hipError_t launch_step_mtc1(const StepLaunch& L, int nt, const UpdateArgs& U, const SamplerArgs& S, int do_sample, bool dense);
hipError_t launch_step_mtb1(const StepLaunch& L, int nt, const UpdateArgs& U, const SamplerArgs& S, int do_sample, bool dense);
hipError_t launch_step_mt2(const StepLaunch& L, int method, int nt, const UpdateArgs& U, const SamplerArgs& S, int do_sample, bool dense);
hipError_t launch_step_mega(const StepLaunch& L, int method, int nt, const UpdateArgs& U, const SamplerArgs& S, int do_sample, bool dense);

// grouped launches (k_group_step): sampler(blocks of group K-1) || update/partial(group K); single trait
hipError_t launch_group_st(const StepLaunch& L, int method, const UpdateArgs& U, const int32_t* uev_idx, const float* uev_delta,
                           const GroupSamplers& SS, const GroupArgs& G);

// Rule T (jwas_sweep_params.section_solve): the inverses of all 64-marker sections of the full 256-marker blocks, once per sweep
// (k_section_inverse_mt; nsections = 4 per full block, tsec: nsections * (64 nt)^2 floats)
hipError_t launch_section_inverse_mtc1(const StepLaunch& L, int nt, const DevParams* P, const float* xpx, const float* gram, int64_t nsections, float* tsec);
hipError_t launch_section_inverse_mtb1(const StepLaunch& L, int nt, const DevParams* P, const float* xpx, const float* gram, const float* ginv_mat, int64_t nsections, float* tsec);

// independent-block sweeps: k_indep_rhs + k_indep_sample
hipError_t launch_indep_st(const StepLaunch& L, int method, const UpdateArgs& U, const SamplerArgs& S, int64_t pstride, bool dense);
hipError_t launch_indep_mtc1(const StepLaunch& L, int nt, const UpdateArgs& U, const SamplerArgs& S, int64_t pstride);
hipError_t launch_indep_mt2(const StepLaunch& L, int nt, const UpdateArgs& U, const SamplerArgs& S, int64_t pstride);      // (sampler II, shared covariance)
hipError_t launch_indep_mega(const StepLaunch& L, int nt, const UpdateArgs& U, const SamplerArgs& S, int64_t pstride);     // (megaBayesC)

}  // namespace jw
