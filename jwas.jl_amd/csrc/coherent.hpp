// coherent.hpp -- accesses to data that is exchanged with a concurrently running kernel.  Included by update_role.hpp and
// sampler_common.hpp.
#pragma once
#include "kernels.hpp"

namespace jw {

// Accesses to data exchanged with a CONCURRENTLY RUNNING kernel (resident sampler, resident.hpp): COH = agent-scope relaxed
// atomics -- write-through stores / loads that do not hit a stale line of this XCD's L2; no fence, no cache invalidate (the
// sampler's XCD keeps its prefetched Gram rows).  COH = false: the plain access of the launch-per-block kernel.
template <bool COH> __device__ __forceinline__ double ld_coh(const double* p)
{
    if constexpr (COH) return __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH> __device__ __forceinline__ int32_t ld_coh(const int32_t* p)
{
    if constexpr (COH) return __hip_atomic_load(const_cast<int32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH> __device__ __forceinline__ float ld_coh(const float* p)
{
    if constexpr (COH) return __int_as_float(__hip_atomic_load(reinterpret_cast<int*>(const_cast<float*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    else return *p;
}
template <bool COH> __device__ __forceinline__ void st_coh(int32_t* p, int32_t v)
{
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool COH> __device__ __forceinline__ void st_coh(float* p, float v)
{
    if constexpr (COH) __hip_atomic_store(reinterpret_cast<int*>(p), __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool COH> __device__ __forceinline__ void st_coh(double* p, double v)
{
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

}  // namespace jw
