// rng.hpp -- counter-based random numbers for the marker sweep (device side).
//
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11), keyed by runMCMC's seed and indexed by
// (marker, iteration, within-block repetition, slot + 16*trait).  A draw therefore does not depend
// on block size, wavefront assignment, marker shard or GPU count.  The reference draws from Julia's
// task-local Xoshiro256++ in marker order (BayesABC.jl:44-54); that stream is not reproducible off
// Julia, so the counter stream is this build's definition of "same seed" (RNG contract).
//
//   slot 0: the decision uniform  u = (k + 0.5) * 2^-52, k = top 52 bits of words (1,0)
//   slot 1: the normal            z = sqrt(-2 ln u1) cos(2 pi u2), u1 from words (1,0), u2 from (3,2)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jw {

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

__device__ __forceinline__ double u52(uint32_t lo, uint32_t hi)
{
    const uint64_t k = (((uint64_t)hi << 32) | lo) >> 12;
    return ((double)k + 0.5) * 0x1.0p-52;
}

struct RngKey { uint32_t seed_lo, seed_hi, iter, rep; };

__device__ __forceinline__ double draw_uniform(const RngKey& key, uint32_t marker, uint32_t trait)
{
    const u32x4 w = philox4x32_10(marker, key.iter, key.rep, 0u + 16u * trait, key.seed_lo, key.seed_hi);
    return u52(w.x, w.y);
}

__device__ __forceinline__ double draw_normal(const RngKey& key, uint32_t marker, uint32_t trait)
{
    const u32x4 w = philox4x32_10(marker, key.iter, key.rep, 1u + 16u * trait, key.seed_lo, key.seed_hi);
    const double u1 = u52(w.x, w.y), u2 = u52(w.z, w.w);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

}  // namespace jw
