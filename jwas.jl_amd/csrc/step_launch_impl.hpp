// step_launch_impl.hpp -- template code shared by the step_*.hip translation units (see step_launch.hpp).
#pragma once
#include "step_launch.hpp"
#include <atomic>
#include <cstdlib>

namespace jw {
namespace {

template <int METHOD, int NT>
inline StepSmem step_smem(int block_size, const SamplerArgs& S, bool dn)
{
    return StepSmem(block_size, NT,
                    is_mt_method(METHOD) ? mt_park_nd(block_size, NT) + (S.lpr_mat ? (1 << NT) : 0) : st_park_nd(METHOD),
                    is_mt_method(METHOD) ? mt_park_nf(block_size, NT) + (has_marker_cov(METHOD) ? NT * NT : 0) : st_park_nf(METHOD, dn));
}

template <int METHOD, int NT, class CX>
hipError_t launch_step_cx(const StepLaunch& L, const CX& cx, const UpdateArgs& U0, const SamplerArgs& S, int do_sample, bool dense)
{
    UpdateArgsT<CX> U;
    static_cast<UpdateArgs&>(U) = U0;
    U.cx = cx;
    // DENSE instantiations: single-trait sweeps under a uniform pi = 0 (Rule D), and the multi-trait samplers' dense-walk-only form
    constexpr bool kHasDense = ((METHOD == kBayesC || METHOD == kBayesB) && NT == 1) || (is_mt_method(METHOD) && !is_sampler2(METHOD));
    const bool dn = kHasDense && dense;
    const StepSmem SM = step_smem<METHOD, NT>(L.block_size, S, dn);
    static std::atomic<unsigned long long> attr_set{0ull};       // one bit per device: the attribute belongs to the device's code object
    const unsigned long long dev_bit = 1ull << (L.device & 63);
    if (!(attr_set.load(std::memory_order_acquire) & dev_bit)) {   // allow > 64 KB of dynamic LDS
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_block_step<METHOD, NT, CX, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if constexpr (CX::kCoopApply) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_block_step<METHOD, NT, CX, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        if constexpr (kHasDense) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_block_step<METHOD, NT, CX, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            if constexpr (CX::kCoopApply) {
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_block_step<METHOD, NT, CX, true, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
            }
        }
        attr_set.fetch_or(dev_bit, std::memory_order_release);
    }
    // JWAS_HIP_DEBUG_ROLE (timing experiments only; results are wrong): 1 = update role only, 2 = sampler only
    // Development builds only (build_dev.sh -DJWAS_HIP_DEV_KNOBS): the shipped library never reads these -- they break results.
#ifdef JWAS_HIP_DEV_KNOBS
    static const int dbg = std::getenv("JWAS_HIP_DEBUG_ROLE") ? std::atoi(std::getenv("JWAS_HIP_DEBUG_ROLE")) : 0;
#else
    constexpr int dbg = 0;
#endif
    const int nwork = L.nrg * U.ncg;
    unsigned grid = (dbg == 2) ? 1u : (U.quiet_xcd ? (unsigned)(1 + (nwork + 6) / 7 * 8) : (unsigned)(1 + nwork));
    if (S.xch != nullptr && do_sample && grid < 17u) grid = 17u;       // workgroup 16 is the helper the sampler publishes to (sweep.hpp)
    const int ds = (dbg == 1) ? 0 : do_sample;
    if constexpr (kHasDense) {
        if (dn) {       // uniform pi = 0: the sampler that follows Rule D (and takes dense_big_st on full 256- / 512-marker blocks)
            if constexpr (CX::kCoopApply) {
                if (U.sync_now != nullptr) {
                    hipLaunchKernelGGL((k_block_step<METHOD, NT, CX, true, true>), dim3(grid), dim3(kStepThreads), SM.bytes, L.stream, U, S, ds);
                    return hipSuccess;
                }
            }
            hipLaunchKernelGGL((k_block_step<METHOD, NT, CX, false, true>), dim3(grid), dim3(kStepThreads), SM.bytes, L.stream, U, S, ds);
            return hipSuccess;
        }
    }
    if constexpr (CX::kCoopApply) {
        if (U.sync_now != nullptr) {     // dense sweep: the instantiation whose update role shares the apply work
            hipLaunchKernelGGL((k_block_step<METHOD, NT, CX, true>), dim3(grid), dim3(kStepThreads), SM.bytes, L.stream, U, S, ds);
            return hipSuccess;
        }
    }
    hipLaunchKernelGGL((k_block_step<METHOD, NT, CX, false>), dim3(grid), dim3(kStepThreads), SM.bytes, L.stream, U, S, ds);
    return hipSuccess;
}

template <int METHOD, int NT>
hipError_t launch_step(const StepLaunch& L, const UpdateArgs& U, const SamplerArgs& S, int do_sample, bool dense)
{
    if (L.packed) return launch_step_cx<METHOD, NT, PackedCols>(L, L.pc, U, S, do_sample, dense);
    return launch_step_cx<METHOD, NT, DenseCols>(L, L.dc, U, S, do_sample, dense);
}

template <int METHOD, class CX>
hipError_t launch_group_cx(const StepLaunch& L, const CX& cx, const UpdateArgs& U0, const int32_t* uev_idx, const float* uev_delta,
                           const GroupSamplers& SS, const GroupArgs& G)
{
    UpdateArgsT<CX> U;
    static_cast<UpdateArgs&>(U) = U0;
    U.cx = cx;
    const StepSmem SM = step_smem<METHOD, 1>(L.block_size, SS.a[0], false);
    static std::atomic<unsigned long long> attr_set{0ull};
    const unsigned long long dev_bit = 1ull << (L.device & 63);
    if (!(attr_set.load(std::memory_order_acquire) & dev_bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_group_step<METHOD, CX, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_group_step<METHOD, CX, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set.fetch_or(dev_bit, std::memory_order_release);
    }
    const int nwork = L.nrg * U.ncg;
    unsigned grid = U.quiet_xcd ? (unsigned)(1 + (nwork + 6) / 7 * 8) : (unsigned)(1 + nwork);
    if (G.pp != 0 && grid < 8u * (unsigned)(G.ns - 1) + 1u) grid = 8u * (unsigned)(G.ns - 1) + 1u;      // ping-pong: block s is sampled by workgroup 8 s
    // (G.pp_kernel: the host's choice per SWEEP -- a sweep's launches share the arrival counters of the cooperative apply)
    if (G.pp_kernel != 0) hipLaunchKernelGGL((k_group_step<METHOD, CX, true>), dim3(grid), dim3(kStepThreads), SM.bytes, L.stream, U, uev_idx, uev_delta, SS, G);
    else hipLaunchKernelGGL((k_group_step<METHOD, CX, false>), dim3(grid), dim3(kStepThreads), SM.bytes, L.stream, U, uev_idx, uev_delta, SS, G);
    return hipSuccess;
}

template <int METHOD>
hipError_t launch_group(const StepLaunch& L, const UpdateArgs& U, const int32_t* uev_idx, const float* uev_delta, const GroupSamplers& SS, const GroupArgs& G)
{
    if (L.packed) return launch_group_cx<METHOD, PackedCols>(L, L.pc, U, uev_idx, uev_delta, SS, G);
    return launch_group_cx<METHOD, DenseCols>(L, L.dc, U, uev_idx, uev_delta, SS, G);
}

// Independent-block sweep (BayesABC_block_independent!, BayesABC.jl:190-255): all block RHS from the residual
// snapshot (one pass over X), all blocks sampled concurrently; the caller compacts the change lists.
template <int METHOD, int NT, class CX>
hipError_t launch_indep_cx(const StepLaunch& L, const CX& cx, const UpdateArgs& U0, const SamplerArgs& S, int64_t pstride, bool dense)
{
    UpdateArgsT<CX> U;
    static_cast<UpdateArgs&>(U) = U0;
    U.cx = cx;
    constexpr bool kHasDense = (METHOD == kBayesC || METHOD == kBayesB) && NT == 1;       // sweeps under a uniform pi = 0 (Rule D)
    const bool dn = kHasDense && dense;
    const StepSmem SM = step_smem<METHOD, NT>(L.block_size, S, dn);
    static std::atomic<unsigned long long> attr_set{0ull};       // one bit per device
    const unsigned long long dev_bit = 1ull << (L.device & 63);
    if (!(attr_set.load(std::memory_order_acquire) & dev_bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_indep_sample<METHOD, NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if constexpr (kHasDense) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_indep_sample<METHOD, NT, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        attr_set.fetch_or(dev_bit, std::memory_order_release);
    }
    const size_t red = sizeof(double) * kRowGroupSlices * kColChunk * NT;
    hipLaunchKernelGGL((k_indep_rhs<NT, CX>), dim3((unsigned)(U.nrg * U.ncg), (unsigned)L.nblocks), dim3(kStepThreads), red, L.stream,
                       U, L.p, L.block_size, pstride, L.d_starts);
    if constexpr (kHasDense) {
        if (dn) {
            hipLaunchKernelGGL((k_indep_sample<METHOD, NT, true>), dim3((unsigned)L.nblocks), dim3(kStepThreads), SM.bytes, L.stream, S, pstride, L.ev_all, L.d_starts);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((k_indep_sample<METHOD, NT>), dim3((unsigned)L.nblocks), dim3(kStepThreads), SM.bytes, L.stream,
                       S, pstride, L.ev_all, L.d_starts);
    return hipGetLastError();
}

template <int METHOD, int NT>
hipError_t launch_indep(const StepLaunch& L, const UpdateArgs& U, const SamplerArgs& S, int64_t pstride, bool dense = false)
{
    if (L.packed) return launch_indep_cx<METHOD, NT, PackedCols>(L, L.pc, U, S, pstride, dense);
    return launch_indep_cx<METHOD, NT, DenseCols>(L, L.dc, U, S, pstride, dense);
}

template <int METHOD, int NT>
hipError_t launch_section_inverse(const StepLaunch& L, const DevParams* P, const float* xpx, const float* gram, const float* ginv_mat,
                                  int64_t nsections, float* tsec)
{
    static std::atomic<unsigned long long> attr_set{0ull};
    const unsigned long long dev_bit = 1ull << (L.device & 63);
    if (!(attr_set.load(std::memory_order_acquire) & dev_bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_section_inverse_mt<METHOD, NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set.fetch_or(dev_bit, std::memory_order_release);
    }
    if (nsections <= 0) return hipSuccess;
    hipLaunchKernelGGL((k_section_inverse_mt<METHOD, NT>), dim3((unsigned)nsections), dim3(256 * NT), tsec_inverse_lds_bytes<NT>(), L.stream,
                       P, xpx, gram, ginv_mat, tsec);
    return hipGetLastError();
}

}  // namespace
}  // namespace jw
