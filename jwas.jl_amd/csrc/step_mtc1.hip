// step_mtc1.hip -- step-kernel instantiations and launchers: multi-trait Gibbs sampler I with one shared effect covariance
// (MTBayesABC.jl:57-127, block form :243-333); see step_launch.hpp.
#define JW_PLAIN_KERNEL static
#include "step_launch_impl.hpp"

namespace jw {

hipError_t launch_step_mtc1(const StepLaunch& L, int nt, const UpdateArgs& U, const SamplerArgs& S, int do_sample, bool dense)
{
    if (nt == 2) return launch_step<kMTBayesC1, 2>(L, U, S, do_sample, dense);
    if (nt == 3) return launch_step<kMTBayesC1, 3>(L, U, S, do_sample, dense);
    return launch_step<kMTBayesC1, 4>(L, U, S, do_sample, dense);
}

hipError_t launch_section_inverse_mtc1(const StepLaunch& L, int nt, const DevParams* P, const float* xpx, const float* gram, int64_t nsections, float* tsec)
{
    if (nt == 2) return launch_section_inverse<kMTBayesC1, 2>(L, P, xpx, gram, nullptr, nsections, tsec);
    if (nt == 3) return launch_section_inverse<kMTBayesC1, 3>(L, P, xpx, gram, nullptr, nsections, tsec);
    return launch_section_inverse<kMTBayesC1, 4>(L, P, xpx, gram, nullptr, nsections, tsec);
}

hipError_t launch_indep_mtc1(const StepLaunch& L, int nt, const UpdateArgs& U, const SamplerArgs& S, int64_t pstride)
{
    if (nt == 2) return launch_indep<kMTBayesC1, 2>(L, U, S, pstride);
    if (nt == 3) return launch_indep<kMTBayesC1, 3>(L, U, S, pstride);
    return launch_indep<kMTBayesC1, 4>(L, U, S, pstride);
}

}  // namespace jw
