// step_st.hip -- step-kernel instantiations and launchers: single-trait BayesA/B/C and BayesR (see step_launch.hpp).
#define JW_PLAIN_KERNEL static
#include "step_launch_impl.hpp"

namespace jw {

hipError_t launch_step_st(const StepLaunch& L, int method, const UpdateArgs& U, const SamplerArgs& S, int do_sample, bool dense)
{
    if (method == kBayesB) return launch_step<kBayesB, 1>(L, U, S, do_sample, dense);
    if (method == kBayesR) return launch_step<kBayesR, 1>(L, U, S, do_sample, false);
    return launch_step<kBayesC, 1>(L, U, S, do_sample, dense);
}

hipError_t launch_group_st(const StepLaunch& L, int method, const UpdateArgs& U, const int32_t* uev_idx, const float* uev_delta,
                           const GroupSamplers& SS, const GroupArgs& G)
{
    if (method == kBayesB) return launch_group<kBayesB>(L, U, uev_idx, uev_delta, SS, G);
    if (method == kBayesR) return launch_group<kBayesR>(L, U, uev_idx, uev_delta, SS, G);
    return launch_group<kBayesC>(L, U, uev_idx, uev_delta, SS, G);
}

hipError_t launch_indep_st(const StepLaunch& L, int method, const UpdateArgs& U, const SamplerArgs& S, int64_t pstride, bool dense)
{
    if (method == kBayesB) return launch_indep<kBayesB, 1>(L, U, S, pstride, dense);
    if (method == kBayesR) return launch_indep<kBayesR, 1>(L, U, S, pstride);
    return launch_indep<kBayesC, 1>(L, U, S, pstride, dense);
}

}  // namespace jw
