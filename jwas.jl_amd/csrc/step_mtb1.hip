// step_mtb1.hip -- step-kernel instantiations and launchers: multi-trait Gibbs sampler I with one effect covariance PER MARKER
// (multi-trait BayesA/B: MTBayesABC.jl:66,86-90); see step_launch.hpp.
#define JW_PLAIN_KERNEL static
#include "step_launch_impl.hpp"

namespace jw {

hipError_t launch_step_mtb1(const StepLaunch& L, int nt, const UpdateArgs& U, const SamplerArgs& S, int do_sample, bool dense)
{
    if (nt == 2) return launch_step<kMTBayesB1, 2>(L, U, S, do_sample, dense);
    if (nt == 3) return launch_step<kMTBayesB1, 3>(L, U, S, do_sample, dense);
    return launch_step<kMTBayesB1, 4>(L, U, S, do_sample, dense);
}

hipError_t launch_section_inverse_mtb1(const StepLaunch& L, int nt, const DevParams* P, const float* xpx, const float* gram, const float* ginv_mat, int64_t nsections, float* tsec)
{
    if (nt == 2) return launch_section_inverse<kMTBayesB1, 2>(L, P, xpx, gram, ginv_mat, nsections, tsec);
    if (nt == 3) return launch_section_inverse<kMTBayesB1, 3>(L, P, xpx, gram, ginv_mat, nsections, tsec);
    return launch_section_inverse<kMTBayesB1, 4>(L, P, xpx, gram, ginv_mat, nsections, tsec);
}

}  // namespace jw
