// resident_launch.hip -- instantiations and launchers of the resident-sampler sweep (resident.hpp, resident_api.hpp).
#define JW_PLAIN_KERNEL static
#include "resident_api.hpp"
#include <atomic>

namespace jw {

namespace {

constexpr bool has_dense_inst(int method, int nt)
{
    return ((method == kBayesC || method == kBayesB) && nt == 1) || (is_mt_method(method) && !is_sampler2(method));
}

template <int METHOD, int NT, bool DENSE>
hipError_t launch_sampler_one(int device, const ResidentArgs& R, size_t lds, hipStream_t stream)
{
    static std::atomic<unsigned long long> attr_set{0ull};       // one bit per device: allow > 64 KB of dynamic LDS
    const unsigned long long dev_bit = 1ull << (device & 63);
    if (!(attr_set.load(std::memory_order_acquire) & dev_bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sampler_resident<METHOD, NT, DENSE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes - 512);
        if (e != hipSuccess) return e;
        attr_set.fetch_or(dev_bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_sampler_resident<METHOD, NT, DENSE>), dim3(kResSamplerGrid), dim3(kStepThreads), lds, stream, R);
    return hipGetLastError();
}

template <int METHOD, int NT>
hipError_t launch_sampler_m(int device, bool dense, const ResidentArgs& R, size_t lds, hipStream_t stream)
{
    if constexpr (has_dense_inst(METHOD, NT)) {
        if (dense) return launch_sampler_one<METHOD, NT, true>(device, R, lds, stream);
    }
    return launch_sampler_one<METHOD, NT, false>(device, R, lds, stream);
}

template <int NT, class CX>
hipError_t launch_update_cx(int device, bool coop, const UpdateArgs& U0, const CX& cx, const ResidentLink& L, const ResidentHelp& H,
                            unsigned grid, hipStream_t stream)
{
    UpdateArgsT<CX> U;
    static_cast<UpdateArgs&>(U) = U0;
    U.cx = cx;
    // the reduction scratch is a few KB; the dense stream asks for more than half of the CU's LDS so that the dispatcher keeps
    // ONE update workgroup per CU, the geometry it was tuned for (nrg x ncg <= 224 workgroups); the 2-bit decode loop is
    // latency-bound and takes two workgroups per CU gladly
    constexpr size_t red = (size_t)kRowGroupSlices * kColChunk * NT * 8;
    constexpr size_t lds = CX::kCoopApply ? (size_t)(81 * 1024) : red;
    static std::atomic<unsigned long long> attr_set{0ull};
    const unsigned long long dev_bit = 1ull << (device & 63);
    if (!(attr_set.load(std::memory_order_acquire) & dev_bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_step<NT, CX, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return e;
        if constexpr (CX::kCoopApply) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_step<NT, CX, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e != hipSuccess) return e;
        }
        attr_set.fetch_or(dev_bit, std::memory_order_release);
    }
    if constexpr (CX::kCoopApply) {
        if (coop && U.sync_now != nullptr) {
            hipLaunchKernelGGL((k_update_step<NT, CX, true>), dim3(grid), dim3(kStepThreads), lds, stream, U, L, H);
            return hipSuccess;
        }
    }
    hipLaunchKernelGGL((k_update_step<NT, CX, false>), dim3(grid), dim3(kStepThreads), lds, stream, U, L, H);
    return hipSuccess;
}

}  // namespace

bool resident_supported(int method, int ntraits)
{
    if (!is_mt_method(method)) return ntraits == 1;
    return ntraits >= 2 && ntraits <= 4;
}

size_t resident_sampler_lds(int method, int ntraits, int bsz, bool dense, bool lpr_mat)
{
    const bool mt = is_mt_method(method);
    const bool dn = has_dense_inst(method, ntraits) && dense;
    const StepSmem SM(bsz, ntraits, mt ? mt_park_nd(bsz, ntraits) + (lpr_mat ? (1 << ntraits) : 0) : st_park_nd(method),
                      mt ? mt_park_nf(bsz, ntraits) + (has_marker_cov(method) ? ntraits * ntraits : 0) : st_park_nf(method, dn));
    return (size_t)SM.bytes;
}

hipError_t launch_sampler_resident(int device, int method, int ntraits, bool dense, const ResidentArgs& R, hipStream_t stream)
{
    const size_t lds = resident_sampler_lds(method, ntraits, R.S.bsz, dense, R.S.lpr_mat != nullptr);
    switch (method) {
        case kBayesC: return launch_sampler_m<kBayesC, 1>(device, dense, R, lds, stream);
        case kBayesB: return launch_sampler_m<kBayesB, 1>(device, dense, R, lds, stream);
        case kBayesR: return launch_sampler_m<kBayesR, 1>(device, false, R, lds, stream);
#define JW_RES_MT(M)                                                                        \
            if (ntraits == 2) return launch_sampler_m<M, 2>(device, dense, R, lds, stream);  \
            if (ntraits == 3) return launch_sampler_m<M, 3>(device, dense, R, lds, stream);  \
            return launch_sampler_m<M, 4>(device, dense, R, lds, stream);
        case kMTBayesC1: JW_RES_MT(kMTBayesC1)
        case kMTBayesC2: JW_RES_MT(kMTBayesC2)
        case kMegaBayesC: JW_RES_MT(kMegaBayesC)
        case kMTBayesB1: JW_RES_MT(kMTBayesB1)
        case kMTBayesB2: JW_RES_MT(kMTBayesB2)
        case kMegaBayesB: JW_RES_MT(kMegaBayesB)
#undef JW_RES_MT
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_update_step(int device, int ntraits, bool coop, const UpdateArgs& U, const DenseCols* dcols, const PackedCols* pcols,
                              const ResidentLink& L, const ResidentHelp& H, unsigned grid, hipStream_t stream)
{
#define JW_RES_UPD(NT)                                                                                       \
        if (dcols) return launch_update_cx<NT, DenseCols>(device, coop, U, *dcols, L, H, grid, stream);       \
        return launch_update_cx<NT, PackedCols>(device, coop, U, *pcols, L, H, grid, stream);
    switch (ntraits) {
        case 1: JW_RES_UPD(1)
        case 2: JW_RES_UPD(2)
        case 3: JW_RES_UPD(3)
        default: JW_RES_UPD(4)
    }
#undef JW_RES_UPD
}

}  // namespace jw
