// update_role.hpp -- UPDATE / PARTIAL role of the fused step kernel (apply a block's changes to the residual -- sparse exit update, or the
// cooperative dense apply -- then the block's partial right-hand side).  Included by sweep.hpp.
#pragma once
#include <type_traits>
#include "kernels.hpp"

#ifdef JWAS_HIP_COOP_RELACQ
#define JW_COOP_ARRIVE_ORDER __ATOMIC_RELEASE
#else
#define JW_COOP_ARRIVE_ORDER __ATOMIC_RELAXED
#endif

namespace jw {

// ---------------------------------------------------------------------------------------------
// UPDATE/PARTIAL role
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// UPDATE/PARTIAL role on 2-BIT PACKED storage (round 5): its own geometry.  A wave owns a 1024-row slice of the residual, a lane
// SIXTEEN rows = ONE DWORD of a column (the byte-per-lane form of rounds 2-4 moved 64 B per wave load and spent ~9 instructions
// per row and column: 16x fewer bytes than dense bought 1.19x).  Per column and lane: one dword load, then per row a bit-field
// extract, a conversion and one fp64 multiply-add; the centring is factored OUT of the loop --
//     sum_i (v_i - mu) w_i r_i  =  sum_i c_i w_i r_i  -  mu (R - M),    c_i = code (0 for a missing one), R = sum_i w_i r_i, M = sum_missing w_i r_i
// (decode_marker!, streaming_genotypes.jl:978-1002: v = code == 3 ? mu : code, x = centered ? v - mu : v; uncentred: + mu M) -- so the
// marker mean enters once per column and lane, and R once per launch.  The sum is the exact-product fp64 sum of the decoded
// column up to the rounding of fl32(code - mu) the dense matrix carries: the packed path's OWN order (oracle: dot_xr with
// orc_set_packed_source), within 1e-7 relative of the dense path's right-hand side (the reference's own stream-vs-dense
// tolerance is 1e-4, test/unit/test_streaming_codec.jl:100,104).  The residual update r += x d stays the decoded form
// fmaf(d, fl32(v - mu), r) of every other kernel.  Reductions (transposed butterfly over 8 columns, LDS combine over the
// row group's waves, one fp64 partial per column and row group) are the dense role's.
// ---------------------------------------------------------------------------------------------
constexpr int kWideRows = 1024;            // rows of r owned by one wave of the packed update role
template <int NT, class CX, class EV>
__device__ __forceinline__ void update_role_wide(char* smem, int rg, int g, const CX& cx,
                                                 const float* __restrict__ r_in, float* __restrict__ r_out,
                                                 const EV& ev,
                                                 int64_t j0, int b, int nslices, int nrg, int ncg,
                                                 double* __restrict__ partials, int bstride, int spg)
{
    typedef double RedT[kColChunk][NT];
    RedT* red = reinterpret_cast<RedT*>(smem);                 // [kRowGroupSlices][kColChunk][NT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slice = rg * spg + wave;
    const bool active = wave < spg && slice < nslices;
    const int64_t ld = cx.ld;
    const int64_t row = (int64_t)(active ? slice : 0) * kWideRows + lane * 16;        // (the last slice may reach beyond ld: masked below)
    const int ncols = (b > g) ? (b - g + ncg - 1) / ncg : 0;
    const int64_t jc0 = j0 + (ncols > 0 ? g : 0);
    const int nc1 = ncols > 0 ? ncols - 1 : 0;
    const int64_t qrow = row >> 2;                              // byte offset of the lane's dword inside a column (a multiple of 4)
    const int64_t cstride = ld >> 2;                            // bytes per column
    const uint8_t* qcol = cx.Q + jc0 * cstride + qrow;
    const int64_t qstep = (int64_t)ncg * cstride;
    auto load_code = [&](int i) -> unsigned {                   // stream element i = marker jc0 + i * ncg (clamped: always a valid address)
        return *reinterpret_cast<const unsigned*>(qcol + (int64_t)(i < ncols ? i : nc1) * qstep);
    };
    constexpr int D = 4;                                        // register batches of kU dwords in flight per wave
    unsigned xr[D][kU];
    float mnext = cx.mean[jc0 + (int64_t)(lane < ncols ? lane : nc1) * ncg];          // means of the first 64 stream elements
#pragma unroll
    for (int s = 0; s < D; ++s)
#pragma unroll
        for (int u = 0; u < kU; ++u) xr[s][u] = active ? load_code(s * kU + u) : 0u;

    // the residual slice and the weights: 16 rows per lane (rows >= ld: 0)
    float rv[NT][16], wv[16];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const bool in = active && (row + 4 * q4 < ld);
        const int64_t rr = in ? row + 4 * q4 : 0;
        const float4 w4 = *reinterpret_cast<const float4*>(cx.w + rr);
        wv[4 * q4] = in ? w4.x : 0.f; wv[4 * q4 + 1] = in ? w4.y : 0.f; wv[4 * q4 + 2] = in ? w4.z : 0.f; wv[4 * q4 + 3] = in ? w4.w : 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 r4 = *reinterpret_cast<const float4*>(r_in + t * ld + rr);
            rv[t][4 * q4] = in ? r4.x : 0.f; rv[t][4 * q4 + 1] = in ? r4.y : 0.f; rv[t][4 * q4 + 2] = in ? r4.z : 0.f; rv[t][4 * q4 + 3] = in ? r4.w : 0.f;
        }
    }
    // sparse exit update (BayesABC.jl:181-185): r += x d per changed marker in list order, x the decoded column (pad rows: 0)
    const int64_t nleft = cx.n - row;
    const unsigned vmask = nleft >= 16 ? 0xffffu : (nleft <= 0 ? 0u : ((1u << (int)nleft) - 1u));
    const int ne = ev.count();
    for (int e0 = 0; e0 < ne; e0 += 64) {
        const int el = e0 + lane, ec = el < ne ? el : ne - 1;
        const int liv = ev.idx(ec);
        const float lmu = cx.mean[liv];
        float ldv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) ldv[t] = ev.delta(t, ec);
        const int rem0 = (ne - e0) < 64 ? (ne - e0) : 64;
        for (int h = 0; h < rem0; h += 8) {
            unsigned cq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int jj = __builtin_amdgcn_readlane(liv, h + u < rem0 ? h + u : rem0 - 1);
                cq[u] = *reinterpret_cast<const unsigned*>(cx.Q + (int64_t)jj * cstride + qrow);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (h + u < rem0) {
                    const float mu = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lmu), h + u));
                    float dd[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) dd[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ldv[t]), h + u));
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float x = ((vmask >> i) & 1u) ? cx.dec((cq[u] >> (2 * i)) & 3u, mu) : 0.f;
#pragma unroll
                        for (int t = 0; t < NT; ++t) rv[t][i] = fmaf(dd[t], x, rv[t][i]);
                    }
                }
            }
        }
    }
    if (active && g == 0 && r_out != nullptr) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
            if (row + 4 * q4 < ld)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    *reinterpret_cast<float4*>(r_out + t * ld + row + 4 * q4) = float4{rv[t][4 * q4], rv[t][4 * q4 + 1], rv[t][4 * q4 + 2], rv[t][4 * q4 + 3]};
    }
    if (ncols == 0) return;
    // fl32(w r) once per launch (weights = 1: exact); one trait keeps it in double (no conversion in the loop)
    typedef typename std::conditional<NT == 1, double, float>::type RT;
    RT rd[NT][16];
    double Rl[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        Rl[t] = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float wr = rv[t][i] * wv[i]; rd[t][i] = (RT)wr; Rl[t] += (double)wr; }
    }
    const bool centered = cx.centered != 0;
    for (int i0 = 0; i0 < ncols; i0 += kColChunk) {
        const int iend = (i0 + kColChunk < ncols) ? i0 + kColChunk : ncols;
        const float mcur = mnext;                                // lane i: mean of stream element i0 + i
        if (i0 + kColChunk < ncols) mnext = cx.mean[jc0 + (int64_t)(i0 + kColChunk + lane < ncols ? i0 + kColChunk + lane : nc1) * ncg];
        for (int ib0 = i0; ib0 < iend; ib0 += kU * D) {
#pragma unroll
            for (int s = 0; s < D; ++s) {
                const int ib = ib0 + s * kU;
                if (ib >= iend) break;
                double acc[NT][kU];
                unsigned anymiss = 0u;
#pragma unroll
                for (int u = 0; u < kU; ++u) anymiss |= xr[s][u] & (xr[s][u] >> 1) & 0x55555555u;
                const bool patch = __any(anymiss != 0u);           // (wave-uniform)
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const unsigned q = xr[s][u];
                    const double mu = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(mcur), ib + u - i0));
                    double sm[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) { acc[t][u] = 0.0; sm[t] = 0.0; }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const double c = (double)((q >> (2 * i)) & 3u);
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[t][u] = fma(c, (double)rd[t][i], acc[t][u]);
                    }
                    if (patch) {                                     // a missing code was counted as 3: take it out, remember its w r
                        const unsigned m = q & (q >> 1) & 0x55555555u;
#pragma unroll
                        for (int i = 0; i < 16; ++i)
                            if ((m >> (2 * i)) & 1u) {
#pragma unroll
                                for (int t = 0; t < NT; ++t) sm[t] += (double)rd[t][i];
                            }
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[t][u] = fma(-3.0, sm[t], acc[t][u]);
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t][u] = centered ? fma(-mu, Rl[t] - sm[t], acc[t][u]) : fma(mu, sm[t], acc[t][u]);
                }
                if (ib + kU * D < ncols) {
#pragma unroll
                    for (int u = 0; u < kU; ++u) xr[s][u] = active ? load_code(ib + kU * D + u) : 0u;
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const double sum = butterfly8(acc[t], lane);
                    const int u = lane >> 3;                   // column of this 8-lane group
                    if ((lane & 7) == 0 && ib + u < ncols) red[wave][ib + u - i0][t] = sum;
                }
            }
        }
        __syncthreads();
        for (int q = tid; q < (iend - i0) * NT; q += kStepThreads) {
            const int i = q / NT, t = q - i * NT;
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < kRowGroupSlices; ++w) s += red[w][i][t];
            const int c = g + (i0 + i) * ncg;
            partials[((int64_t)t * nrg + rg) * bstride + c] = s;
        }
        __syncthreads();
    }
}

// ROLL: the rolling-window apply for 33..64 changes (below) -- only where the kernel's register budget carries it (the
// instantiation's sampler role decides: single-trait BayesA/B/C; with BayesR's or the multi-trait samplers' it spilled)
template <int NT, class CX, bool COOP = false, bool ROLL = false, class EV = EvPlain>
__device__ __forceinline__ void update_role(char* smem, int rg, int g,
                                            const CX& cx,
                                            const float* __restrict__ r_in, float* __restrict__ r_out,
                                            const EV& ev,
                                            int64_t j0, int b, int nslices, int nrg, int ncg,
                                            double* __restrict__ partials, int bstride, int spg = kRowGroupSlices,
                                            int* sync_now = nullptr, int* sync_next = nullptr, unsigned long long* dbg = nullptr)
{
    if constexpr (CX::kWide) {       // 2-bit packed storage: its own geometry (above)
        (void)sync_now; (void)sync_next; (void)dbg;
        update_role_wide<NT, CX, EV>(smem, rg, g, cx, r_in, r_out, ev, j0, b, nslices, nrg, ncg, partials, bstride, spg);
        return;
    }
#ifdef JWAS_HIP_DEV_KNOBS
#define JW_UPD_CLOCK(v) v = clock64()
#else
#define JW_UPD_CLOCK(v) (void)0
#endif
    long long tu0 = 0, tu1 = 0, tu3 = 0;
    JW_UPD_CLOCK(tu0);
    typedef double RedT[kColChunk][NT];
    RedT* red = reinterpret_cast<RedT*>(smem);                 // [kRowGroupSlices][kColChunk][NT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slice = rg * spg + wave;
    const bool active = wave < spg && slice < nslices;
    // an inactive wave (slice beyond the matrix) aliases slice 0 for addressing and contributes 0
    const int64_t row = (int64_t)(active ? slice : 0) * kSliceRows + lane * 4;
    const int ncols = (b > g) ? (b - g + ncg - 1) / ncg : 0;
    const int64_t ld = cx.ld;

    // Loads are unconditional from clamped, always-valid addresses: a select between a load and a
    // constant makes hipcc pick between pointers and emit flat/scratch accesses.
    const int64_t jc0 = j0 + (ncols > 0 ? g : 0);
    const int nc1 = ncols > 0 ? ncols - 1 : 0;
    typedef typename CX::SRaw Raw;
    constexpr int D = CX::kDepth;                      // register batches in flight per wave
    const typename CX::Stream st = cx.stream(jc0, ncg, row);     // element i = marker jc0 + i*ncg, this lane's 4 rows
    auto load_batch = [&](Raw (&dst)[kU], int ib) {
        if (active) {                                  // (wave-uniform; an idle wave streams nothing)
#pragma unroll
            for (int u = 0; u < kU; ++u) dst[u] = st.load_raw(ib + u < ncols ? ib + u : nc1);
        } else {
#pragma unroll
            for (int u = 0; u < kU; ++u) dst[u] = Raw{};
        }
    };

    // (1) the first batch(es) of column loads do not depend on r: issue them before the update.
    //     Dense: in-flight depth is ONE batch per wave (8 KB): with ~1600 waves streaming that is ~13 MB outstanding,
    //     enough for full HBM rate; doubling it only lengthens the memory queues (Little's law) and with them
    //     the latency of every dependent load of the concurrently running sampler role.
    //     2-bit packed: a batch is 8 x 64 B per wave, so the loop is latency-bound and keeps D batches in flight.
    float mnext = st.load_mean(lane < ncols ? lane : nc1);     // packed storage: marker means of the first 64 stream elements
    Raw xr[D][kU];
#pragma unroll
    for (int s = 0; s < D; ++s) load_batch(xr[s], s * kU);

    // (2) sparse exit update: sequential fmaf in marker order, bit-identical to the oracle's per-marker
    //     axpy sequence.  Every column group recomputes it (reads r_in only); group 0 stores r_out.
    float4 rv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) rv[t] = *reinterpret_cast<const float4*>(r_in + t * ld + row);
    // (the residual weights: fetched here, with the role's first loads -- behind the apply they were one more dependent
    // memory round trip before the stream could start)
    float4 wv = *reinterpret_cast<const float4*>(cx.w + row);
    // (dense priors apply a whole block of changes here: 16 column loads in flight per wave, the fmaf chain per row
    // stays in list order)
    const int ne = ev.count();
    // the head of the list, one entry per lane, issued with the count (not after it: the arrays are always there, what lies
    // beyond the count is never used as an address or a coefficient) -- the general apply below then needs ONE further
    // memory latency per 32 changes instead of two per 16 (a scalar index load, then the columns: with 30-40 changes per
    // 512-marker block -- BayesR, a fixed pi, the first sweeps of a chain -- that was 4-6 dependent round trips, ~12 us of a
    // 29 us launch)
    int liv_n = ev.idx(lane);
    float ldv_n[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) ldv_n[t] = ev.delta(t, lane);
    // ---- COOPERATIVE DENSE APPLY.  With a dense prior every launch applies a whole block of changes (ne ~ b), and every
    // column group of a row group re-reading the same ne columns makes the update role the bottleneck of the launch (8 x
    // 25.6 MB at n = 50 000, b = 128).  Here the ncg workgroups of a row group SPLIT the rows of every slice: wave w of
    // group g updates rows [g*R, (g+1)*R) of its slice (R = ceil(256 / ncg) <= 64, one row per lane, dword loads, 64 of
    // them in flight per lane), the fused multiply-add chain per row in list order -- the same operations as the float4
    // path, bit for bit.  The shares go to r_out, a counter per row group (agent scope, zeroed by the previous launch)
    // tells when all ncg shares have landed, and every group then reads its slices' new residual back.  The wait is
    // BOUNDED: if the peers do not show up (workgroups not co-resident) the group falls back to applying everything
    // itself -- same values either way, so the fallback is only slower.
    bool applied = false;
    if constexpr (COOP && CX::kCoopApply) {
        const int R = (kSliceRows + ncg - 1) / ncg;
        if (sync_now != nullptr && r_out != nullptr && ne >= 32 && ncg >= 4 && R <= 64) {
            // Rows of this group: [g*R, (g+1)*R) of every slice of the row group.  When R divides 64 (ncg = 8: R = 32) a wave
            // takes the share of 64 / R slices at once, so that all 64 lanes carry a row: the phase is bound by instruction
            // issue (one readlane + address + load and one readlane + fma per entry and row), and seven half-empty waves on
            // four SIMDs cost twice what four full ones do.
            // With at least as many column groups as slices (the usual geometry: 7-8 slices, 8-16 groups) group g takes slice g
            // of the row group WHOLE (waves 0..3: 64 consecutive rows each): 256 contiguous bytes per wave load and 1 KB per
            // column and workgroup, instead of 128-byte (or, with 16 groups, 64-byte: every line fetched by two XCDs) pieces
            // of every slice -- 9.65 -> 9.2 ms per sweep on the reference benchmark shape.  Otherwise the row-fraction split.
            const bool slice_map = ncg >= spg;
            const int pack = slice_map ? 1 : ((64 % R == 0) ? 64 / R : 1);
            const int sl_w = slice_map ? g : wave * pack + lane / R;      // slice of the row group this lane works for
            const int rloc = slice_map ? wave * 64 + lane : g * R + (pack > 1 ? lane % R : lane);        // row of the slice
            const int slice_l = rg * spg + sl_w;
            const bool mine = slice_map ? (g < spg && slice_l < nslices && wave < 4)
                                        : (sl_w < spg && slice_l < nslices && (pack > 1 || lane < R) && rloc < kSliceRows);
            const int64_t grow = (int64_t)(mine ? slice_l : 0) * kSliceRows + (mine ? rloc : 0);
            float rs[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) rs[t] = r_in[t * ld + grow];
            // 64 changes per chunk: lane l fetches entry e0 + l of the list (index and coefficients: coalesced), every
            // lane then loads its row of the 64 columns back to back (addresses from v_readlane: no scalar memory access,
            // no branch, one memory latency) and runs the chain in list order; entries past the end have coefficient 0 (an
            // exact no-op on a valid column).  (128 per pass, and two chunks in flight, were measured: not faster.)
            int iv_n; float dv_n[NT];
            auto load_list = [&](int e0) {
                const int el = e0 + lane, ec = el < ne ? el : ne - 1;
                iv_n = ev.idx(ec);
#pragma unroll
                for (int t = 0; t < NT; ++t) { const float d = ev.delta(t, ec); dv_n[t] = (el < ne) ? d : 0.f; }
            };
            if (slice_map ? (wave < 4 && g < spg) : (wave * pack < spg)) {   // (wave-uniform: the other waves have no share)
            load_list(0);
            for (int e0 = 0; e0 < ne; e0 += 64) {
                const int iv = iv_n;
                float dv[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) dv[t] = dv_n[t];
                if (e0 + 64 < ne) load_list(e0 + 64);                   // the next chunk's list: in flight behind this chunk's columns
                float x[64];
#pragma unroll
                for (int u = 0; u < 64; ++u) x[u] = cx.load1(__builtin_amdgcn_readlane(iv, u), grow);
#pragma unroll
                for (int u = 0; u < 64; ++u) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        rs[t] = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(dv[t]), u)), x[u], rs[t]);
                }
            }
            }
            // the shares and the counter travel as agent-scope accesses (write-through / coherent reads): no L2 write-back or
            // invalidate, which would cost every other workgroup of the XCD its cached columns
            if (mine)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    __hip_atomic_store(reinterpret_cast<int*>(r_out + t * ld + grow), __float_as_int(rs[t]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's share has been written ...
            JW_UPD_CLOCK(tu1);
            int* flag = reinterpret_cast<int*>(smem);                     // (the reduction scratch is not in use yet)
            __syncthreads();
            if (tid == 0) {
                // JWAS_HIP_COOP_RELACQ (measurement builds): the formally ordered variant -- release on the arrival, acquire on
                // the successful poll.  On gfx950 each is a full L2 write-back / invalidate of the workgroup's XCD.
                __hip_atomic_fetch_add(&sync_now[rg], 1, JW_COOP_ARRIVE_ORDER, __HIP_MEMORY_SCOPE_AGENT);     // ... before the count
                int ok = 0;
                for (int spin = 0; spin < 4000; ++spin) {                 // bounded: ~0.5 ms
                    if (__hip_atomic_load(&sync_now[rg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ncg) { ok = 1; break; }
                    __builtin_amdgcn_s_sleep(4);
                }
#ifdef JWAS_HIP_COOP_RELACQ
                if (ok) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");         // (the successful poll, acquired once)
#endif
                *flag = ok;
            }
            __syncthreads();
            const int ok = *flag;
            __syncthreads();                                              // (flag's bytes are reused below)
            if (ok) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int* src = reinterpret_cast<const int*>(r_out + t * ld + row);
                    rv[t].x = __int_as_float(__hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    rv[t].y = __int_as_float(__hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    rv[t].z = __int_as_float(__hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    rv[t].w = __int_as_float(__hip_atomic_load(src + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                }
                applied = true;
            }
            JW_UPD_CLOCK(tu3);
        }
    }
    if (applied) {
    } else
    if (NT == 1 && ne <= 7) {
        // header path: indices and coefficients arrived with the count (one 64-byte line)
        float4 x[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) x[u] = cx.load4(u < ne ? ev.hidx(u) : 0, row);     // (unused slots: column 0, always valid)
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            if (u < ne) {
                const float d = ev.hdelta(u);
                rv[0].x = fmaf(d, x[u].x, rv[0].x); rv[0].y = fmaf(d, x[u].y, rv[0].y);
                rv[0].z = fmaf(d, x[u].z, rv[0].z); rv[0].w = fmaf(d, x[u].w, rv[0].w);
            }
        }
    } else
    for (int e0 = 0; e0 < ne; e0 += 64) {
        // lane l: entry e0 + l of the list; the next 64 entries are fetched behind this chunk's columns
        const int liv = liv_n;
        float ldv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) ldv[t] = ldv_n[t];
        if (e0 + 64 < ne) {
            const int el = e0 + 64 + lane, ec = el < ne ? el : ne - 1;
            liv_n = ev.idx(ec);
#pragma unroll
            for (int t = 0; t < NT; ++t) ldv_n[t] = ev.delta(t, ec);
        }
        // K columns in flight per lane (addresses from v_readlane: no scalar memory access), the fused multiply-add chain per
        // row in list order; entries past the end re-read the last valid column and are skipped
        auto chunk = [&](auto kc, int h, int rem) {
            constexpr int K = decltype(kc)::value;
            float4 x[K];
#pragma unroll
            for (int u = 0; u < K; ++u) x[u] = cx.load4(__builtin_amdgcn_readlane(liv, h + (u < rem ? u : rem - 1)), row);
#pragma unroll
            for (int u = 0; u < K; ++u) {
                if (u < rem) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ldv[t]), h + u));
                        rv[t].x = fmaf(d, x[u].x, rv[t].x); rv[t].y = fmaf(d, x[u].y, rv[t].y);
                        rv[t].z = fmaf(d, x[u].z, rv[t].z); rv[t].w = fmaf(d, x[u].w, rv[t].w);
                    }
                }
            }
        };
        const int rem0 = ne - e0;                              // >= 1 entries of this list chunk (lanes 0 .. min(rem0, 64) - 1)
        if (rem0 <= 16) chunk(std::integral_constant<int, 16>{}, 0, rem0);
        else if constexpr (CX::kDepth != 1) {
            // (2-bit packed storage: the stream keeps 8 batches of bytes in registers -- 16 columns at a time here)
            for (int h = 0; h < 64 && e0 + h < ne; h += 16) chunk(std::integral_constant<int, 16>{}, h, ne - (e0 + h));
        }
        else if (rem0 <= 32) chunk(std::integral_constant<int, 32>{}, 0, rem0);
        else if constexpr (!ROLL) {
            for (int h = 0; h < 64 && e0 + h < ne; h += 32) {
                const int rem = ne - (e0 + h);
                if (rem <= 16) chunk(std::integral_constant<int, 16>{}, h, rem);
                else chunk(std::integral_constant<int, 32>{}, h, rem);
            }
        } else {
            // 33 .. 64 changes: a ROLLING window of 32 columns -- as soon as a column has been applied its registers take the
            // column 32 entries further on, so the second half's loads are in flight behind the first half's arithmetic
            // instead of one more dependent round trip later (fixed pi = 0.95: ~37 changes per 512-marker block)
            const int rem = rem0 < 64 ? rem0 : 64;
            float4 x[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) x[u] = cx.load4(__builtin_amdgcn_readlane(liv, u), row);
#pragma unroll
            for (int u = 0; u < 32; ++u) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ldv[t]), u));
                    rv[t].x = fmaf(d, x[u].x, rv[t].x); rv[t].y = fmaf(d, x[u].y, rv[t].y);
                    rv[t].z = fmaf(d, x[u].z, rv[t].z); rv[t].w = fmaf(d, x[u].w, rv[t].w);
                }
                x[u] = cx.load4(__builtin_amdgcn_readlane(liv, 32 + u < rem ? 32 + u : rem - 1), row);
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                if (32 + u < rem) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ldv[t]), 32 + u));
                        rv[t].x = fmaf(d, x[u].x, rv[t].x); rv[t].y = fmaf(d, x[u].y, rv[t].y);
                        rv[t].z = fmaf(d, x[u].z, rv[t].z); rv[t].w = fmaf(d, x[u].w, rv[t].w);
                    }
                }
            }
        }
    }
    if (!applied) { JW_UPD_CLOCK(tu1); tu3 = tu1; }               // (development builds: entry + apply | partial sums)
    if (active && g == 0 && r_out != nullptr && !applied)
#pragma unroll
        for (int t = 0; t < NT; ++t) *reinterpret_cast<float4*>(r_out + t * ld + row) = rv[t];
    // the next launch's arrival counter (its previous user is done).  Stores and loads share vmcnt: issued here, after the
    // last wait of the apply phase that is on a dependent path, the store delays nothing.
    if constexpr (COOP) { if (sync_next != nullptr && g == 0 && tid == 0) sync_next[rg] = 0; }
    if (ncols == 0) return;
    // the RHS is X_b' R^-1 r (block_rhs!, tools4genotypes.jl:59-78): the weights go onto r once per launch (weights = 1
    // when unweighted: exact), the streaming loop is untouched
    if (!active) wv = float4{0.f, 0.f, 0.f, 0.f};
    double rd[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        rd[t][0] = rv[t].x * wv.x; rd[t][1] = rv[t].y * wv.y; rd[t][2] = rv[t].z * wv.z; rd[t][3] = rv[t].w * wv.w;
    }

    // (3) partial block RHS.  (kColChunk / kU batches per chunk is a multiple of D, so ring slot = batch % D.)
    for (int i0 = 0; i0 < ncols; i0 += kColChunk) {
        const int iend = (i0 + kColChunk < ncols) ? i0 + kColChunk : ncols;
        const float mcur = mnext;                                // lane i: mean of stream element i0 + i
        if (i0 + kColChunk < ncols) mnext = st.load_mean(i0 + kColChunk + lane < ncols ? i0 + kColChunk + lane : nc1);
        for (int ib0 = i0; ib0 < iend; ib0 += kU * D) {
#pragma unroll
            for (int s = 0; s < D; ++s) {
                const int ib = ib0 + s * kU;
                if (ib >= iend) break;
                double acc[NT][kU];
                auto products = [&](auto dec) {
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const float mu = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mcur), ib + u - i0));
                        const float4 xa = dec(xr[s][u], mu);
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            acc[t][u] = (double)xa.x * rd[t][0];
                            acc[t][u] = fma((double)xa.y, rd[t][1], acc[t][u]);
                            acc[t][u] = fma((double)xa.z, rd[t][2], acc[t][u]);
                            acc[t][u] = fma((double)xa.w, rd[t][3], acc[t][u]);
                        }
                    }
                };
                unsigned fl = 0u;                              // packed storage: does any byte of the batch hold a missing code?
#pragma unroll
                for (int u = 0; u < kU; ++u) fl |= CX::Stream::flags(xr[s][u]);
                if (__any(fl != 0u)) products([&](const Raw& r, float mu) { return st.decode_patch(r, mu); });   // wave-uniform branch
                else products([&](const Raw& r, float mu) { return st.decode_fast(r, mu); });
                if (ib + kU * D < ncols) load_batch(xr[s], ib + kU * D);   // registers are free again: refill the slot
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const double sum = butterfly8(acc[t], lane);
                    const int u = lane >> 3;                   // column of this 8-lane group
                    if ((lane & 7) == 0 && ib + u < ncols) red[wave][ib + u - i0][t] = sum;
                }
            }
        }
        __syncthreads();
        for (int q = tid; q < (iend - i0) * NT; q += kStepThreads) {
            const int i = q / NT, t = q - i * NT;
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < kRowGroupSlices; ++w) s += red[w][i][t];
            const int c = g + (i0 + i) * ncg;
            partials[((int64_t)t * nrg + rg) * bstride + c] = s;
        }
        __syncthreads();
    }
#ifdef JWAS_HIP_DEV_KNOBS
    if (dbg != nullptr && rg == 0 && g == 0 && tid == 0) {               // development builds: one workgroup's phases
        const long long tu4 = clock64();
        atomicAdd(&dbg[13], (unsigned long long)(tu1 - tu0));            // cooperative apply: own share
        atomicAdd(&dbg[14], (unsigned long long)(tu3 - tu1));            //   wait for the peers + read back
        atomicAdd(&dbg[15], (unsigned long long)(tu4 - tu3));            // the rest (partial RHS)
    }
#else
    (void)dbg; (void)tu0; (void)tu1; (void)tu3;
#endif
}


}  // namespace jw
