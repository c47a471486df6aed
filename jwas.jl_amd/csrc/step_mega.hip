// step_mega.hip -- step-kernel instantiations and launchers: megaBayesABC! (constraint = true; BayesABC.jl:1-8), shared or
// per-marker variances; see step_launch.hpp.
#define JW_PLAIN_KERNEL static
#include "step_launch_impl.hpp"

namespace jw {

hipError_t launch_step_mega(const StepLaunch& L, int method, int nt, const UpdateArgs& U, const SamplerArgs& S, int do_sample, bool dense)
{
    if (method == kMegaBayesB) {
        if (nt == 2) return launch_step<kMegaBayesB, 2>(L, U, S, do_sample, dense);
        if (nt == 3) return launch_step<kMegaBayesB, 3>(L, U, S, do_sample, dense);
        return launch_step<kMegaBayesB, 4>(L, U, S, do_sample, dense);
    }
    if (nt == 2) return launch_step<kMegaBayesC, 2>(L, U, S, do_sample, dense);
    if (nt == 3) return launch_step<kMegaBayesC, 3>(L, U, S, do_sample, dense);
    return launch_step<kMegaBayesC, 4>(L, U, S, do_sample, dense);
}

hipError_t launch_indep_mega(const StepLaunch& L, int nt, const UpdateArgs& U, const SamplerArgs& S, int64_t pstride)
{
    if (nt == 2) return launch_indep<kMegaBayesC, 2>(L, U, S, pstride);
    if (nt == 3) return launch_indep<kMegaBayesC, 3>(L, U, S, pstride);
    return launch_indep<kMegaBayesC, 4>(L, U, S, pstride);
}

}  // namespace jw
