// step_mt2.hip -- step-kernel instantiations and launchers: multi-trait Gibbs sampler II (MTBayesABC.jl:129-210), shared or
// per-marker effect covariance; see step_launch.hpp.
#define JW_PLAIN_KERNEL static
#include "step_launch_impl.hpp"

namespace jw {

hipError_t launch_step_mt2(const StepLaunch& L, int method, int nt, const UpdateArgs& U, const SamplerArgs& S, int do_sample, bool dense)
{
    if (method == kMTBayesB2) {
        if (nt == 2) return launch_step<kMTBayesB2, 2>(L, U, S, do_sample, dense);
        if (nt == 3) return launch_step<kMTBayesB2, 3>(L, U, S, do_sample, dense);
        return launch_step<kMTBayesB2, 4>(L, U, S, do_sample, dense);
    }
    if (nt == 2) return launch_step<kMTBayesC2, 2>(L, U, S, do_sample, dense);
    if (nt == 3) return launch_step<kMTBayesC2, 3>(L, U, S, do_sample, dense);
    return launch_step<kMTBayesC2, 4>(L, U, S, do_sample, dense);
}

hipError_t launch_indep_mt2(const StepLaunch& L, int nt, const UpdateArgs& U, const SamplerArgs& S, int64_t pstride)
{
    if (nt == 2) return launch_indep<kMTBayesC2, 2>(L, U, S, pstride);
    if (nt == 3) return launch_indep<kMTBayesC2, 3>(L, U, S, pstride);
    return launch_indep<kMTBayesC2, 4>(L, U, S, pstride);
}

}  // namespace jw
