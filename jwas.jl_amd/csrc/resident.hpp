// resident.hpp -- the marker sweep with a SAMPLER WORKGROUP THAT STAYS RESIDENT ACROSS BLOCKS (gfx950, wave64).
//
// The launch-per-block step (sweep.hpp) starts the sampler of block k-1 as workgroup 0 of launch k: every block pays the
// sampler's front as cold code (its text is larger than the instruction cache it shares) after a dispatch, and a sampler
// that is slower than the stream holds the next launch back.  Here the roles are two kernels on two streams:
//
//   k_sampler_resident  (ONE launch per sweep, one workgroup, the sampler roles of sweep.hpp unchanged -- same chain, same
//                        bits): loops over the blocks; for block sb it waits until the update kernel of launch sb has
//                        reported the block's row-group partial sums (arrive[sb] = number of update workgroups), runs the
//                        block's single-site chain, writes the change list with write-through stores and publishes
//                        done = sb + 1.
//   k_update_step       (one launch per block, stream order, the UPDATE/PARTIAL role of sweep.hpp unchanged): launch k
//                        waits (inside the kernel, after its first column loads are in flight) for done >= k - 1, applies the
//                        changes of block k-2, streams block k once, stores the partial sums write-through and arrives.
//
// The exchange is the cooperative apply's (update_role.hpp): agent-scope relaxed accesses -- write-through stores, loads that
// do not hit stale lines, one counter -- and no fences (a release / acquire pair is an L2 write-back / invalidate per
// workgroup: the sampler's XCD would lose the Gram rows it prefetched).  Every wait is BOUNDED by the wall clock; whoever
// gives up sets `abort`, everybody else leaves at the next look at it, and the host restores the snapshot it took before the
// sweep (alpha, beta, delta, r) and re-runs the sweep through the launch-per-block path: a sweep can be slow, it cannot
// hang and it cannot be wrong.  Reference semantics unchanged: BayesABC.jl:118-188, BayesR.jl:111-193, MTBayesABC.jl:243-333.
#pragma once
#include "sweep.hpp"

namespace jw {

// Device-resident synchronisation block of a sweep (zeroed by the host before the sweep): done, abort and the sampler's claim
// in their own 128-byte lines, then per block: one arrival counter, one work-ticket counter (quiet sweeps), one helper claim.
//
// PLACEMENT.  The sampler wants an XCD whose L2 is not thrashed by the stream (sweep.hpp: quiet_xcd), and it needs a CU whose
// whole LDS is free.  Workgroup b of a dispatch is observed on XCC (b + offset) % 8 with an offset that moves from dispatch
// to dispatch, so "workgroup ids = 0 mod 8" names a different XCD in every update launch.  Here the XCD is FIXED instead: the
// sampler kernel is launched with 16 workgroups and the first one that finds itself on XCC kResXcc claims the role (the
// others exit at once); update workgroups on that XCC do no streaming in quiet sweeps, and since the static workgroup ->
// work mapping would then have holes, the streaming workgroups of a quiet launch draw their (row group, column group) from a
// ticket counter.  Nothing here is needed for correctness: a missing sampler or a missing worker ends in a bounded wait and
// the host's re-run.
constexpr int kResDone = 0, kResAbort = 32, kResClaim = 64, kResArrive = 96;      // int offsets
constexpr int kResXcc = 0;
constexpr int kResSamplerGrid = 16;
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 15); }      // HW_REG_XCC_ID
constexpr long long kResidentSamplerTimeout = 200000000ll;        // wall_clock64 ticks (100 MHz): 2 s without the next block's partial sums

struct ResidentArgs {
    SamplerArgs S;                // the block-independent fields (P, nrg, bstride, p, bsz, xpx, prep_*, state, counters, ...)
    int64_t nb;                   // blocks of the sweep
    const int64_t* starts;        // explicit partition (nb + 1 entries) or NULL: uniform blocks of S.bsz markers
    const double* partials;       // [2][pstride] ping-pong (block parity)
    int64_t pstride;
    const float* gram;            // block sb at sb * bsz * bsz
    const float* cross;           // cross-Gram X_{sb-1}' X_sb at sb * bsz * bsz
    float* corr;                  // [2][kMaxT][bsz]
    Events* ev;                   // [2]
    int* sync;                    // kResDone / kResAbort / kResClaim / kResArrive + block
    int ncg;                      // column groups of a full launch (a block of fewer markers than that uses one per marker)
};

// grid = kResSamplerGrid, block = 512, dynamic LDS as k_block_step.
template <int METHOD, int NT, bool DENSE>
__global__ __launch_bounds__(kStepThreads) void k_sampler_resident(ResidentArgs R)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_go;
    const int tid = threadIdx.x;
    SamplerArgs S = R.S;
    const int bs = S.bsz;
    // one workgroup on XCC kResXcc becomes the sampler (see PLACEMENT)
    if (xcc_id() != kResXcc) return;
    if (tid == 0) s_go = (atomicCAS(R.sync + kResClaim, 0, 1) == 0) ? 1 : 0;
    __syncthreads();
    if (!s_go) return;
    __syncthreads();
    if (tid == 0) {      // diagnostics: when the sampler started
        R.sync[kResAbort + 8] = xcc_id();
        *reinterpret_cast<long long*>(R.sync + kResAbort + 10) = wall_clock64();
    }
    const int64_t gs = (int64_t)bs * bs;
#pragma unroll 1
    for (int64_t sb = 0; sb < R.nb; ++sb) {
        const int64_t j0 = R.starts ? R.starts[sb] : sb * bs;
        const int b = R.starts ? (int)(R.starts[sb + 1] - j0) : (int)((j0 + bs <= S.p) ? bs : S.p - j0);
        int b_next = 0, b_after = 0;
        if (sb + 1 < R.nb) {
            const int64_t j1 = j0 + b;
            b_next = R.starts ? (int)(R.starts[sb + 2] - j1) : (int)((j1 + bs <= S.p) ? bs : S.p - j1);
            if (sb + 2 < R.nb) {
                const int64_t j2 = j1 + b_next;
                b_after = R.starts ? (int)(R.starts[sb + 3] - j2) : (int)((j2 + bs <= S.p) ? bs : S.p - j2);
            }
        }
        // ---- wait for the block's partial sums (bounded)
        const long long tw0 = clock64();
        if (tid == 0) {
            const int need = S.nrg * (R.ncg > b ? b : R.ncg);
            int* arrive = R.sync + kResArrive + 4 * sb;
            int ok = 0;
            const long long t0 = wall_clock64();
            while (true) {
                if (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) { ok = 1; break; }
                if (__hip_atomic_load(R.sync + kResAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                if (wall_clock64() - t0 > kResidentSamplerTimeout) {
                    if (__hip_atomic_exchange(R.sync + kResAbort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {      // (diagnostics: who gave up first)
                        R.sync[kResAbort + 1] = 2; R.sync[kResAbort + 2] = (int)sb; R.sync[kResAbort + 3] = __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        R.sync[kResAbort + 4] = need;
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            s_go = ok;
        }
        __syncthreads();              // (also: the previous block's final stores read LDS that the front is about to rewrite)
        if (!s_go) return;
        const long long tw1 = clock64();
        S.partials = R.partials + (sb & 1) * R.pstride;
        S.j0 = j0; S.b = b;
        S.gram = R.gram + sb * gs;
        S.b_next = b_next;
        S.cross_next = R.cross + (sb + 1 < R.nb ? sb + 1 : sb) * gs;
        S.gram_next = (sb + 1 < R.nb) ? R.gram + (sb + 1) * gs : nullptr;
        S.cross_after = (sb + 2 < R.nb) ? R.cross + (sb + 2) * gs : nullptr;
        S.lines_after = (sb + 2 < R.nb) ? (int)(((int64_t)b_next * b_after + 31) / 32) : 0;
        S.corr_in = R.corr + (sb & 1) * (size_t)kMaxT * bs;
        S.corr_out = R.corr + ((sb + 1) & 1) * (size_t)kMaxT * bs;
        S.ev_out = R.ev + (sb & 1);
        if constexpr (is_mt_method(METHOD)) sampler_role_mt<METHOD, NT, DENSE, true>(smem, S);
        else sampler_role_st<METHOD, DENSE, true>(smem, S);
        // ---- publish: every thread's stores (change list: write-through; corr_out for this workgroup's next block) are
        // acknowledged before the count moves
        const long long tw2 = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(R.sync + kResDone, (int)(sb + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(&S.counters[13], (unsigned long long)(tw1 - tw0));              // diagnostics: waiting for the partial sums
            atomicAdd(&S.counters[14], (unsigned long long)(clock64() - tw2));        //              draining the block's stores
        }
    }
}

// The update / partial role of one block as its own kernel.
//   quiet_xcd = 0: grid = nrg * ncg, workgroup w does work item w (the static mapping of k_block_step);
//   quiet_xcd = 1: grid >= 8/7 of that: workgroups on the sampler's XCC stream nothing (one of them may prefetch for the
//                  sampler, DENSE sweeps), the others draw their work item from the launch's ticket counter.
struct ResidentHelp {             // DENSE sweeps: what one idle workgroup pulls into the sampler XCD's L2 (see k_block_step)
    const float* gram_next; int nl_g;
    const float* cross_after; int nl_c;
};
template <int NT, class CX, bool COOP>
__global__ __launch_bounds__(kStepThreads) void k_update_step(UpdateArgsT<CX> U, ResidentLink L, ResidentHelp H)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int w = blockIdx.x;
    if (U.quiet_xcd) {
        int* tk = L.ticket;                                            // {arrive, ticket, helper claim, -} of this launch's block
        int* slot = reinterpret_cast<int*>(smem);
        if (threadIdx.x == 0) {
            int v;
            if (xcc_id() == kResXcc) v = ((H.nl_g + H.nl_c) > 0 && atomicCAS(tk + 1, 0, 1) == 0) ? -2 : -1;
            else v = atomicAdd(tk, 1);
            *slot = v;
        }
        __syncthreads();
        w = *slot;
        __syncthreads();                                               // (the scratch is reused by the role)
        if (w == -2) {
            float sink = 0.f;
            for (int l0 = 0; l0 < H.nl_g + H.nl_c; l0 += 8 * kStepThreads) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    int l = l0 + u * kStepThreads + (int)threadIdx.x;
                    l = l < H.nl_g + H.nl_c ? l : H.nl_g + H.nl_c - 1;
                    v[u] = (l < H.nl_g) ? H.gram_next[(int64_t)l * 32] : H.cross_after[(int64_t)(l - H.nl_g) * 32];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) sink += v[u];
            }
            asm volatile("" ::"v"(sink));
            return;
        }
        if (w < 0) return;
    }
    if (w >= U.nrg * U.ncg) return;
    update_role<NT, CX, COOP, true>(smem, w % U.nrg, w / U.nrg, U.cx, U.r_in, U.r_out, U.ev, U.j0, U.b,
                                    U.nslices, U.nrg, U.ncg, U.partials, U.bstride, U.spg, U.sync_now, U.sync_next, U.dbg, L);
}

}  // namespace jw
