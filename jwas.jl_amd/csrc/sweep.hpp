// sweep.hpp -- the fused per-block step of the marker sweep (gfx950, wave64).
//
// One sweep = nblocks + 1 launches of k_block_step, one per marker block, each holding TWO roles
// that run concurrently on different CUs (one-block lookahead, oracle: orc_*_lookahead_sweep):
//
//   launch k:  workgroup 0      = SAMPLER of block k-1
//              workgroups 1..   = UPDATE/PARTIAL of block k
//
//   UPDATE/PARTIAL (all other CUs, HBM-bound).  Workgroup (rg, g): row group rg = 8 wavefronts, each
//       owning a 256-row slice of the residual in registers; column group g = columns g, g+ncg, ...
//       (a) applies the net effect changes of block k-2   r += X[:,events] * d   (sparse exit update,
//           BayesABC.jl:181-185; sequential fmaf in marker order) -> residual r(k-2);
//       (b) streams the columns of block k ONCE from HBM and forms the partial block RHS
//           X_k[slice,:]' r(k-2)[slice]  (block_rhs!, tools4genotypes.jl:59-78): fp64-accumulated,
//           wave64 butterfly shuffle reduction, cross-wave combine in LDS, one partial per
//           (column, row group).
//   SAMPLER (one workgroup).  rhs_{k-1} = sum of row-group partials (computed by launch k-1 from the
//       residual r(k-3)), corrected for the changes of block k-2 with the cross-Gram X_{k-2}'X_{k-1}
//           rhs[c] = fmaf(d_j, C[j][c], rhs[c])       for the changed markers j, in marker order,
//       then ONE wavefront runs the exact single-site chain of the block (BayesABC.jl:153-179) by
//       speculative parallel evaluation: all lanes evaluate their marker against the current rhs; the
//       first lane whose effect changes commits; its Gram row corrects every rhs (BayesABC.jl:169,172);
//       the rest re-evaluate.  Lanes before the first change are final, so the result is the sequential
//       chain's.  Gram rows of the markers that look like changes at entry are staged in LDS by the
//       whole workgroup before the serial part, so a committed change costs an LDS read, not an HBM
//       round trip.
//
// Launch k depends only on launch k-1 (stream order): no in-kernel inter-workgroup communication.
#pragma once
#include "kernels.hpp"

namespace jw {

constexpr int kStepThreads = 512;
constexpr int kRowsBytes = 128 * 1024;         // LDS budget for staged Gram rows (sampler role)
// Dynamic-LDS carve of one step workgroup (bytes); B = block size, NT = traits, (nd, nf) = doubles /
// floats of per-marker sampler constants staged for the serial wave (0 = none).
constexpr int kLdsBytes = 160 * 1024;
struct StepSmem {
    int B, NT, max_cand;
    int rhs_off, acur_off, astart_off, bcur_off, dcur_off, slot_off, cand_off, wcnt_off, lpr_off, log_off, prepd_off, prepf_off, rows_off, cross_off, bytes;
    bool has_cross;
    __host__ __device__ StepSmem(int B_, int NT_, int nd, int nf) : B(B_), NT(NT_)
    {
        rhs_off  = 0;                               // float [NT][B]  running block RHS
        acur_off = rhs_off + NT * B * 4;            // float [NT][B]  current alpha
        astart_off = acur_off + NT * B * 4;         // float [NT][B]  alpha at block entry
        bcur_off = astart_off + NT * B * 4;         // float [NT][B]  current beta   (multi-trait)
        dcur_off = bcur_off + NT * B * 4;           // float [NT][B]  current delta  (multi-trait)
        slot_off = dcur_off + NT * B * 4;           // int16 [B]      LDS slot of a marker's Gram row, -1 = not staged
        cand_off = slot_off + B * 2;                // int16 [B]
        wcnt_off = (cand_off + B * 2 + 15) / 16 * 16;          // int [16]
        lpr_off  = wcnt_off + 64;                   // double [16]    multi-trait: log prior probability of each of the 2^NT states
        log_off  = lpr_off + 128;                   // int2  [B]      committed changes of this block: {slot, bits(D)}
        prepd_off = log_off + B * 8;                // double [nd][B] per-marker constants (rep 0)
        prepf_off = prepd_off + nd * B * 8;         // float  [nf][B]
        rows_off = prepf_off + nf * B * 4;          // float [max_cand + 1][B] staged Gram rows + one overflow row
        int room = (kLdsBytes - 1024 - rows_off) / (4 * B) - 1;
        if (room > kRowsBytes / (4 * B)) room = kRowsBytes / (4 * B);
        max_cand = room < B ? room : B;
        if (max_cand < 1) max_cand = 1;
        // small blocks (B <= 128, all rows staged): the cross-Gram rows X_this'X_next of the block are copied to LDS too
        // (by the waves that idle during the serial phase), so the lookahead correction at the end reads LDS only
        cross_off = rows_off + (max_cand + 1) * B * 4;
        has_cross = (B <= 128) && (max_cand >= B) && (cross_off + B * B * 4 <= kLdsBytes - 1024);
        const int samp = cross_off + (has_cross ? B * B * 4 : 0);
        const int red = kRowGroupSlices * kColChunk * NT * 8;   // update role: double [8][64][NT]
        bytes = samp > red ? samp : red;
    }
};

// Multi-trait samplers park the per-marker draws of repetition 0 (NT thresholds + NT normals, fp64) and x'x in LDS
// when the block is small enough to leave room for the staged Gram rows; otherwise the serial wave reads them from HBM.
__host__ __device__ constexpr int mt_park_nd(int B, int NT) { return (B * NT <= 2048) ? 2 * NT : 0; }
__host__ __device__ constexpr int mt_park_nf(int B, int NT) { return (B * NT <= 2048) ? 1 + NT : 0; }   // x'x, log C11 per trait

// ---------------------------------------------------------------------------------------------
// UPDATE/PARTIAL role
// ---------------------------------------------------------------------------------------------
template <int NT, class CX, bool COOP = false>
__device__ __forceinline__ void update_role(char* smem, int rg, int g,
                                            const CX& cx,
                                            const float* __restrict__ r_in, float* __restrict__ r_out,
                                            const Events* __restrict__ ev,
                                            int64_t j0, int b, int nslices, int nrg, int ncg,
                                            double* __restrict__ partials, int bstride, int spg = kRowGroupSlices,
                                            int* sync_now = nullptr, int* sync_next = nullptr, unsigned long long* dbg = nullptr)
{
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
    // (dense priors apply a whole block of changes here: 16 column loads in flight per wave, the fmaf chain per row
    // stays in list order)
    const int ne = ev->count;
    constexpr int kEB = 16;
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
            const int pack = (64 % R == 0) ? 64 / R : 1;
            const int sl_w = wave * pack + lane / R;                      // slice of the row group this lane works for
            const int rloc = g * R + (pack > 1 ? lane % R : lane);        // row of the slice
            const int slice_l = rg * spg + sl_w;
            const bool mine = sl_w < spg && slice_l < nslices && (pack > 1 || lane < R) && rloc < kSliceRows;
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
                iv_n = ev->idx[ec];
#pragma unroll
                for (int t = 0; t < NT; ++t) { const float d = ev->delta[t][ec]; dv_n[t] = (el < ne) ? d : 0.f; }
            };
            if (wave * pack < spg) {                                      // (wave-uniform: the waves beyond the packed slices have no share)
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
                __hip_atomic_fetch_add(&sync_now[rg], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ... before the count
                int ok = 0;
                for (int spin = 0; spin < 4000; ++spin) {                 // bounded: ~0.5 ms
                    if (__hip_atomic_load(&sync_now[rg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ncg) { ok = 1; break; }
                    __builtin_amdgcn_s_sleep(4);
                }
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
        for (int u = 0; u < 7; ++u) x[u] = cx.load4(u < ne ? ev->hidx[u] : 0, row);     // (unused slots: column 0, always valid)
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            if (u < ne) {
                const float d = ev->hdelta[u];
                rv[0].x = fmaf(d, x[u].x, rv[0].x); rv[0].y = fmaf(d, x[u].y, rv[0].y);
                rv[0].z = fmaf(d, x[u].z, rv[0].z); rv[0].w = fmaf(d, x[u].w, rv[0].w);
            }
        }
    } else
    for (int e0 = 0; e0 < ne; e0 += kEB) {
        float4 x[kEB];
#pragma unroll
        for (int u = 0; u < kEB; ++u) x[u] = cx.load4(ev->idx[e0 + u < ne ? e0 + u : ne - 1], row);
#pragma unroll
        for (int u = 0; u < kEB; ++u) {
            if (e0 + u < ne) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float d = ev->delta[t][e0 + u];
                    rv[t].x = fmaf(d, x[u].x, rv[t].x); rv[t].y = fmaf(d, x[u].y, rv[t].y);
                    rv[t].z = fmaf(d, x[u].z, rv[t].z); rv[t].w = fmaf(d, x[u].w, rv[t].w);
                }
            }
        }
    }
    if (active && g == 0 && r_out != nullptr && !applied)
#pragma unroll
        for (int t = 0; t < NT; ++t) *reinterpret_cast<float4*>(r_out + t * ld + row) = rv[t];
    // the next launch's arrival counter (its previous user is done).  Stores and loads share vmcnt: issued here, after the
    // last wait of the apply phase that is on a dependent path, the store delays nothing.
    if constexpr (COOP) { if (sync_next != nullptr && g == 0 && tid == 0) sync_next[rg] = 0; }
    if (ncols == 0) return;
    // the RHS is X_b' R^-1 r (block_rhs!, tools4genotypes.jl:59-78): the weights go onto r once per launch (weights = 1
    // when unweighted: exact), the streaming loop is untouched
    float4 wv = *reinterpret_cast<const float4*>(cx.w + row);
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
        atomicAdd(&dbg[15], (unsigned long long)(tu4 - (applied ? tu3 : tu0)));    // the rest (float4 apply if any, partial RHS)
    }
#else
    (void)dbg; (void)tu0; (void)tu1; (void)tu3;
#endif
}

// ---------------------------------------------------------------------------------------------
// SAMPLER role, shared front end: rhs assembly, cross-Gram correction, candidate row staging.
// Per-marker running state (rhs, alpha, ...) lives in LDS; the serial wave keeps only the active
// 64-marker sub-block in registers.
// ---------------------------------------------------------------------------------------------
struct SamplerArgs {
    const DevParams* P;
    const double* partials;       // [NT][nrg][bstride] of THIS block
    int nrg, bstride;
    int64_t j0; int b; int64_t p;
    int bsz;                      // nominal block size (LDS strides)
    const float* xpx;
    const float* gram;            // b x b, this block
    const float* cross_next;      // b x b_next: X_this' X_next (row = marker of THIS block); b_next = 0: none
    int b_next;
    const float* gram_next;       // b_next x b_next Gram of the NEXT block (L2 prefetch only), or NULL
    const float* cross_after;     // cross-Gram X_next' X_(next+1) the NEXT launch's sampler reads (L2 prefetch only), or NULL
    int lines_after;              // ... its size in 128-byte lines
    int dense_big_off;            // != 0: never take dense_big_st (tests: the same chain through the general path)
    const float* corr_in;         // [NT][bsz] lookahead correction of THIS block (written by the previous sampler)
    float* corr_out;              // [NT][bsz] lookahead correction of the NEXT block
    const double* prep_d; const float* prep_f;
    const double* mt2_tab;        // sampler II, <= 3 traits: per-marker state tables (k_prepare_mt2), else NULL
    const double* lpr_mat;        // multi-trait: p x 2^t marker-specific log prior of the joint states, else NULL
    const float* ginv_mat;        // multi-trait BayesA/B (kMTBayesB1): p x t x t per-marker G^-1 (k_prepare), else NULL
    float* alpha; float* beta; void* delta;
    Events* ev_out;
    unsigned long long* counters;
};

// fp64 sum of one column's row-group partials in fixed (ascending row group) order; the first N loads are issued
// back to back from clamped addresses (no load depends on another).
template <int N>
__device__ __forceinline__ double sum_partials_n(const double* pp, int nrg, int64_t stride)
{
    double v[N];
#pragma unroll
    for (int u = 0; u < N; ++u) v[u] = pp[(int64_t)(u < nrg ? u : nrg - 1) * stride];
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < N; ++u) if (u < nrg) sum += v[u];
    for (int rg = N; rg < nrg; rg += 16) {                        // very tall matrices only
        double w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = pp[(int64_t)(rg + u < nrg ? rg + u : nrg - 1) * stride];
#pragma unroll
        for (int u = 0; u < 16; ++u) if (rg + u < nrg) sum += w[u];
    }
    return sum;
}
__device__ __forceinline__ double sum_partials(const double* pp, int nrg, int64_t stride)
{
    if (nrg <= 8) return sum_partials_n<8>(pp, nrg, stride);      // (uniform branches: nrg is a launch constant)
    if (nrg <= 16) return sum_partials_n<16>(pp, nrg, stride);
    return sum_partials_n<32>(pp, nrg, stride);
}

// All NT traits of one column at once: the loads of a chunk of row groups are issued back to back for every trait (one
// memory latency per chunk instead of one per trait), the sums per trait in the same ascending order as sum_partials.
template <int NT>
__device__ __forceinline__ void sum_partials_traits(const double* pp, int64_t tstride, int nrg, int64_t stride, double (&sum)[NT])
{
    constexpr int kC = (NT <= 2) ? 16 : (NT == 3 ? 12 : 8);
#pragma unroll
    for (int t = 0; t < NT; ++t) sum[t] = 0.0;
    for (int rg = 0; rg < nrg; rg += kC) {
        double v[NT][kC];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int u = 0; u < kC; ++u) v[t][u] = pp[t * tstride + (int64_t)(rg + u < nrg ? rg + u : nrg - 1) * stride];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int u = 0; u < kC; ++u) if (rg + u < nrg) sum[t] += v[t][u];
    }
}

// End of the sampler role (all threads): the lookahead correction of the NEXT block from the net changes
// of this one,  corr[c] = fmaf(d_e, C[e][c], corr[c])  from 0 in marker order (C = X_this' X_next).
// fin (LDS, int2 {local column, bits(d)} per trait-0 ... ) holds the compact change list; dlds the
// per-trait changes [NT][B] indexed by local column.
template <int NT>
__device__ __forceinline__ void corr_phase(char* smem, const StepSmem& SM, const SamplerArgs& A, int nfin, bool cross_in_lds = false)
{
    const int B = SM.B;
    const int* fin = reinterpret_cast<const int*>(smem + SM.log_off);              // local columns, marker order
    const float* acur = reinterpret_cast<const float*>(smem + SM.acur_off);
    const float* astart = reinterpret_cast<const float*>(smem + SM.astart_off);
    const int bn = A.b_next;
    for (int c = threadIdx.x; c < B; c += kStepThreads) {
        float corr[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) corr[t] = 0.f;
        if (c < bn) {
            if (cross_in_lds) {                                  // rows copied by copy_cross_rows during the serial phase
                const float* crossL = reinterpret_cast<const float*>(smem + SM.cross_off);
                for (int e = 0; e < nfin; ++e) {
                    const int ce = fin[e];
                    const float g = crossL[ce * B + c];
#pragma unroll
                    for (int t = 0; t < NT; ++t) corr[t] = fmaf(astart[t * B + ce] - acur[t * B + ce], g, corr[t]);
                }
            } else
            for (int e0 = 0; e0 < nfin; e0 += 16) {
                float g[16];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    g[u] = A.cross_next[(int64_t)fin[e0 + u < nfin ? e0 + u : nfin - 1] * bn + c];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (e0 + u < nfin) {
                        const int ce = fin[e0 + u];
#pragma unroll
                        for (int t = 0; t < NT; ++t) corr[t] = fmaf(astart[t * B + ce] - acur[t * B + ce], g[u], corr[t]);
                    }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) A.corr_out[t * B + c] = corr[t];
    }
}

// Waves 1..4 of the sampler workgroup in single-pass sweeps: the lookahead correction of the NEXT block,
//   corr[c] = fmaf(d_e, C[e][c], corr[c])  from 0 over the changed markers e in marker order  (C = X_this' X_next),
// accumulated WHILE the serial wave runs.  The serial wave commits in marker order and publishes every change
// {local column, alpha_old - alpha_new} to a log in LDS with one 8-byte write (entries are pre-set to column -1; wc[13] is
// set after the last one); each helper lane owns four columns of the next block and consumes the log as it grows -- the
// cross-Gram rows come from L2 (prefetch_cross_rows) or HBM, off the critical path.  When the serial wave is done the
// correction is (nearly) done too: no dependent fetch of the changed markers' rows at the end of the block.
// Spinning on LDS inside one workgroup is safe: all its waves are resident.
// The helpers are waves 1..4 (keeping wave 4 -- the serial wave's SIMD -- idle instead was measured: no difference);
// waves 5..7 prefetch.
__device__ __forceinline__ bool is_corr_helper(int wave) { return wave >= 1 && wave <= 4; }
__device__ __forceinline__ int corr_helper_index(int wave) { return wave - 1; }
__device__ __forceinline__ float4 stream_corr_role(char* smem, const StepSmem& SM, const SamplerArgs& A)
{
    const int B = SM.B, bn = A.b_next;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int* wc = reinterpret_cast<int*>(smem + SM.wcnt_off);
    const int2* plog = reinterpret_cast<const int2*>(smem + SM.log_off);
    const int col = (corr_helper_index(wave) * 64 + lane) * 4;
    const bool vec = (bn & 3) == 0;                         // full next block: rows 16-byte aligned
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    int done = 0;
    const volatile int* vcol = reinterpret_cast<const volatile int*>(plog);       // entry e: {column, bits(d)}; column -1 = not written yet
    while (true) {
        const int fin = __hip_atomic_load(&wc[13], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        // the serial wave publishes an entry with ONE 8-byte LDS write (no counter, no wait on its side): count the
        // valid entries after `done` (at most 8 per visit)
        int n = done;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (n == done + u && n < B && vcol[2 * n] >= 0) ++n;
        if (n > done) {
            if (col < bn) {
                for (int e0 = done; e0 < n; e0 += 8) {
                    int2 le[8];
                    float4 g[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) le[u] = plog[e0 + u < n ? e0 + u : n - 1];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float* src = A.cross_next + (int64_t)le[u].x * bn + col;
                        if (vec) g[u] = *reinterpret_cast<const float4*>(src);
                        else { g[u].x = src[0]; g[u].y = src[col + 1 < bn ? 1 : 0]; g[u].z = src[col + 2 < bn ? 2 : 0]; g[u].w = src[col + 3 < bn ? 3 : 0]; }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (e0 + u < n) {
                            const float d = __int_as_float(le[u].y);
                            c0 = fmaf(d, g[u].x, c0); c1 = fmaf(d, g[u].y, c1); c2 = fmaf(d, g[u].z, c2); c3 = fmaf(d, g[u].w, c3);
                        }
                }
            }
            done = n;
        } else if (fin) break;
        else __builtin_amdgcn_s_sleep(8);                  // ~500 cycles between polls
    }
    return float4{c0, c1, c2, c3};      // stored by the caller with the role's other global stores (after the barrier)
}

// Waves 1..7 (after their other prefetch work, while wave 0 runs the serial phase): pull the Gram rows the NEXT block's
// sampler will stage into this XCD's L2 -- the whole Gram block for small (dense-prior) blocks, else the rows of the
// markers that are in the model (alpha != 0: always candidates).  A row fetch of the sampler workgroup competes with
// ~220 streaming workgroups for HBM; here it is off the critical path, in the next launch it is an L2 hit.  Speed only.
// stop (LDS, may be NULL): set by the serial wave when it is done -- prefetching is optional work and must never hold the
// workgroup's barrier back.
__device__ __forceinline__ void prefetch_next_gram(const SamplerArgs& A, bool whole_block, int w0 = 1, const int* stop = nullptr)
{
    const int lane = threadIdx.x & 63;
    const int bn = A.b_next;
    const int nw = kStepThreads / 64 - w0;                 // waves w0 .. 7 do the work
    if ((int)(threadIdx.x >> 6) < w0 || bn <= 0 || A.gram_next == nullptr) return;
    const int wave = (int)(threadIdx.x >> 6) - w0 + 1;     // 1 .. nw
    float sink = 0.f;
    if (whole_block) {
        const int nlines = (bn * bn + 31) / 32;                                    // 128-byte lines of the next Gram block
        float v[4];                                                                // (<= 4 x 448 lines: a 128 x 128 block has 512)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int l = (wave - 1) * 64 + lane + u * nw * 64;
            v[u] = A.gram_next[(int64_t)(l < nlines ? l : nlines - 1) * 32];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) sink += v[u];
    } else {
        // every lane whose marker is in the model touches the lines of ITS row: independent loads, one wait at the end
        const int lines_per_row = (bn + 31) / 32;                                  // <= 32 for 1024-marker blocks
        for (int c0 = (wave - 1) * 64; c0 < bn; c0 += nw * 64) {
            if (stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
            const int c = c0 + lane;
            const float a = A.alpha[A.j0 + A.b + (c < bn ? c : 0)];
            if (c < bn && a != 0.f) {
                const float* row = A.gram_next + (int64_t)c * bn;
                for (int l0 = 0; l0 < lines_per_row; l0 += 8) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = row[(l0 + u < lines_per_row ? l0 + u : lines_per_row - 1) * 32];
#pragma unroll
                    for (int u = 0; u < 8; ++u) sink += v[u];
                }
            }
        }
    }
    asm volatile("" ::"v"(sink));
}

// Waves 1..7: touch the cross-Gram rows (X_this' X_next) of the staged candidates so that corr_phase finds
// them in L2 instead of paying an HBM round trip at the end of the chain.
__device__ __forceinline__ void prefetch_cross_rows(char* smem, const StepSmem& SM, const SamplerArgs& A, int ncand, int w0 = 1, const int* stop = nullptr)
{
    const short* cand_list = reinterpret_cast<const short*>(smem + SM.cand_off);
    const int lane = threadIdx.x & 63;
    const int bn = A.b_next;
    const int nw = kStepThreads / 64 - w0;                 // waves w0 .. 7 do the work
    if ((int)(threadIdx.x >> 6) < w0 || bn <= 0) return;
    const int wave = (int)(threadIdx.x >> 6) - w0 + 1;     // 1 .. nw
    const int nchunk = (bn + 63) / 64, ntask = ncand * nchunk;
    float sink = 0.f;
    for (int t0 = (wave - 1) * 8; t0 < ntask; t0 += nw * 8) {
        if (stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int task = (t0 + u < ntask) ? t0 + u : ntask - 1;
            const int row = task / nchunk, c = (task - row * nchunk) * 64 + lane;
            v[u] = A.cross_next[(int64_t)cand_list[row] * bn + (c < bn ? c : 0)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) sink += v[u];
    }
    asm volatile("" ::"v"(sink));
}

// Waves 1..7 (small blocks): copy the block's cross-Gram rows X_this' X_next (b rows x bn columns) into LDS while wave 0
// runs the serial phase; corr_phase then needs no global access at the end of the chain.
__device__ __forceinline__ void copy_cross_rows(char* smem, const StepSmem& SM, const SamplerArgs& A)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bn = A.b_next, b = A.b, B = SM.B;
    if (wave == 0 || bn <= 0) return;
    float* crossL = reinterpret_cast<float*>(smem + SM.cross_off);
    const int nchunk = (bn + 63) / 64, ntask = b * nchunk;
    constexpr int kD = 16;
    for (int t0 = (wave - 1) * kD; t0 < ntask; t0 += (kStepThreads / 64 - 1) * kD) {
        float v[kD];
#pragma unroll
        for (int u = 0; u < kD; ++u) {
            const int task = (t0 + u < ntask) ? t0 + u : ntask - 1;
            const int row = task / nchunk, c = (task - row * nchunk) * 64 + lane;
            v[u] = A.cross_next[(int64_t)row * bn + (c < bn ? c : 0)];
        }
#pragma unroll
        for (int u = 0; u < kD; ++u) {
            const int task = t0 + u;
            if (task < ntask) {
                const int row = task / nchunk, c = (task - row * nchunk) * 64 + lane;
                if (c < B) crossL[row * B + c] = v[u];
            }
        }
    }
}

// Linear copy global -> LDS with direct loads (global_load_lds_dwordx4: 1 KB per wave instruction), all 8 waves, rolled
// loop, no registers; nfloats a multiple of 256.  The caller waits (s_waitcnt vmcnt(0)) and synchronises.
__device__ __forceinline__ void dma_copy_to_lds(const float* __restrict__ src, float* lds_dst, int nfloats, int w0 = 0)
{
    typedef __attribute__((address_space(3))) void lds_void;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < w0) return;                                   // (waves w0..7 share the copy)
    for (int k = (wave - w0) * 256; k < nfloats; k += (kStepThreads / 64 - w0) * 256)
        __builtin_amdgcn_global_load_lds(src + k + lane * 4, (lds_void*)(lds_dst + k), 16, 0, 0);
}

// Stage the Gram rows of the candidate markers (cand[q] for marker c = tid + q*kStepThreads) in LDS.
__device__ __forceinline__ int stage_rows(char* smem, const StepSmem& SM, const SamplerArgs& A, const bool (&cand)[2], long long* ts = nullptr)
{
    const int B = SM.B;
    short* slot_of = reinterpret_cast<short*>(smem + SM.slot_off);
    short* cand_list = reinterpret_cast<short*>(smem + SM.cand_off);
    int* wcnt = reinterpret_cast<int*>(smem + SM.wcnt_off);
    float* rows = reinterpret_cast<float*>(smem + SM.rows_off);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = A.b;
    int base = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (q * kStepThreads >= B) break;
        const int c = tid + q * kStepThreads;
        const unsigned long long m = __ballot(cand[q]);
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int pre = base, tot = base;
        for (int w = 0; w < kStepThreads / 64; ++w) { if (w < wave) pre += wcnt[w]; tot += wcnt[w]; }
        if (c < B) {
            int sl = -1;
            if (cand[q]) {
                sl = pre + __popcll(m & ((1ull << lane) - 1ull));
                if (sl < SM.max_cand) cand_list[sl] = (short)c; else sl = -1;
            }
            slot_of[c] = (short)sl;
        }
        base = tot;
        __syncthreads();
    }
    const int ncand = base < SM.max_cand ? base : SM.max_cand;
    if (ts) ts[0] = clock64();
    // ALL row loads of the workgroup are issued before the first one is consumed: the fetch costs ONE memory latency
    // (microseconds under full-rate streaming), not one per batch.  Full blocks of 256 / 512 / 1024 markers (rows 16-byte
    // aligned): (row, 256-column) tasks, one float4 per lane, task = u*8 + wave.  The task -> (row, chunk) mapping uses
    // shifts only and the candidates' row indices are fetched from LDS in one batch first: measured, the address
    // arithmetic (a runtime division and a dependent LDS read per task, 24 tasks per wave whatever the count) cost more
    // than the memory latency itself -- 20 k of the 21 k cycles this function took per 512-marker block.
    const int b4 = A.b;
    if (b4 == B && (B == 256 || B == 512 || B == 1024)) {
        // Direct global -> LDS loads (global_load_lds_dwordx4: each lane's 16 bytes land at M0 + lane*16, i.e. one task =
        // 1 KB of a row straight into its slot): no staging registers, a ROLLED loop of a few instructions with every load
        // in flight, one wait at the end.  (The unrolled register version spent 14 k cycles per block just issuing: cold
        // straight-line code is fetched at memory latency.)
        typedef __attribute__((address_space(3))) void lds_void;
        const int sh = (B == 1024) ? 2 : (B == 512 ? 1 : 0);             // log2(256-column chunks per row)
        const int ntask = ncand << sh;
        // lane u of the wave holds the marker of its u-th task (task = wave + 8u)
        const int tmine = wave + 8 * lane;
        const int mycand = (int)cand_list[(tmine < ntask ? tmine : 0) >> sh];
        int u = 0;
        for (int task = wave; task < ntask; task += kStepThreads / 64, ++u) {
            const int crow = __builtin_amdgcn_readlane(mycand, u);
            const int ch = (task & ((1 << sh) - 1)) << 8;
            __builtin_amdgcn_global_load_lds(A.gram + (crow * B + ch + lane * 4),
                                             (lds_void*)(rows + ((task >> sh) * B + ch)), 16, 0, 0);
        }
        if (ts) ts[1] = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ts) ts[2] = clock64();
    } else {
    // (row, 64-column chunk) tasks, kSL independent loads in flight per wave (ragged last block, 64/128-marker blocks)
    constexpr int kSL = 16;
    const int nchunk = B / 64, ntask = ncand * nchunk;
    for (int t0 = wave * kSL; t0 < ntask; t0 += (kStepThreads / 64) * kSL) {
        float v[kSL];
        int dst[kSL];
#pragma unroll
        for (int u = 0; u < kSL; ++u) {
            const int task = (t0 + u < ntask) ? t0 + u : ntask - 1;
            const int row = task / nchunk, c = (task - row * nchunk) * 64 + lane;
            v[u] = A.gram[(int64_t)cand_list[row] * b + (c < b ? c : 0)];
            dst[u] = row * B + c;
        }
#pragma unroll
        for (int u = 0; u < kSL; ++u) if (t0 + u < ntask) rows[dst[u]] = v[u];
    }
    }
    __syncthreads();
    return ncand;
}

// rhs[t][:] += D[t] * G[ce][:]  for the committed marker ce (BayesABC.jl:169,172); one wave.
template <int NT>
__device__ __forceinline__ void apply_gram_row(char* smem, const StepSmem& SM, const SamplerArgs& A, int ce,
                                               const float (&D)[NT], int lane)
{
    const int B = SM.B, b = A.b;
    float* rhs_lds = reinterpret_cast<float*>(smem + SM.rhs_off);
    const short* slot_of = reinterpret_cast<const short*>(smem + SM.slot_off);
    const float* rows = reinterpret_cast<const float*>(smem + SM.rows_off);
    const int sl = __builtin_amdgcn_readfirstlane((int)slot_of[ce]);
    if (sl >= 0) {                                               // staged row: LDS only (explicit branch --
        for (int c2 = lane; c2 < B; c2 += 64) {                  // a select would still issue the global load)
            const float g = rows[sl * B + c2];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (D[t] != 0.f) rhs_lds[t * B + c2] = fmaf(D[t], g, rhs_lds[t * B + c2]);
        }
    } else {
        const float* grow = A.gram + (int64_t)ce * b;             // symmetric: row = column
        for (int c2 = lane; c2 < B; c2 += 64) {
            const float g = grow[c2 < b ? c2 : 0];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (D[t] != 0.f) rhs_lds[t * B + c2] = fmaf(D[t], g, rhs_lds[t * B + c2]);
        }
        if (lane == 0) atomicAdd(&A.counters[1], 1ull);           // diagnostic: changes whose row was not staged
    }
}

// ---------------------------------------------------------------------------------------------
// SAMPLER role, single trait.  METHOD in {kBayesC, kBayesB, kBayesR}.
//
// Per-marker constants parked in LDS for the serial wave (rep 0):
//   BayesA/B/C: doubles [zs]                      floats [1/lhs, beta_excl, x'x, lo, hi]   (lo/hi: AbcMarker::thresholds)
//   BayesR    : doubles [1/lhs_k, zs_k, T_k] (9)  floats [x'x, candidate threshold]
// ---------------------------------------------------------------------------------------------
// ---- DENSE blocks: one 64-marker section of the in-lane walk (see sampler_role_st).  Lane l owns marker l of the section
// (Q = 0: running rhs r0; Q = 1: r1); at step l lane l's alpha_old - alpha_new is broadcast with one v_readlane and applied
// to the running rhs of the section's own markers (Q = 0) and of the next section's (TWO) with the marker's Gram row
// (grow: LDS, row stride B; read a batch of eight rows ahead).  rev = the rhs the lane's own marker was evaluated
// against.  ALLINC: every marker is included whatever its rhs (no compare / select on the chain).  No branch inside
// a batch; the dependent chain per step is add, mul, mul, cvt, add(f64), cvt, sub, readlane, fma.
__device__ __forceinline__ float dense_alpha_new(float x, float da, float ie, float invLhs, double zs, bool incl)
{
    const float rhs  = (x + da) * ie;                                       // BayesABC.jl:36  (da = d * alpha_old)
    const float gHat = rhs * invLhs;                                        // :39
    return incl ? (float)((double)gHat + zs) : 0.f;                         // :46 / :55
}
// RULED: the sweep runs under Rule D (uniform pi = 0): alpha_new = fmaf(kc1, x, kc0) -- the chain is fma, sub, readlane, fma.
template <int Q, bool TWO, bool ALLINC, bool RULED = false>
__device__ __forceinline__ void dense_section(const float* grow, int B, int nsteps, int lane, float ie, float lo, float hi,
                                              float il, float da, float ao, double zs, float& r0, float& r1, float& rev,
                                              float kc1 = 0.f, float kc0 = 0.f)
{
    auto step = [&](int l, float c0, float c1) {
        const float x = (Q == 0) ? r0 : r1;
        const float an = RULED ? fmaf(kc1, x, kc0) : dense_alpha_new(x, da, ie, il, zs, ALLINC ? true : abc_included(x, lo, hi));
        const float Dl = ao - an;                                           // (excluded: alpha_old - 0)
        rev = (lane == l) ? x : rev;
        const float D = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Dl), l));
        if (Q == 0) r0 = fmaf(D, c0, r0);                                   // D = 0: exact no-op
        if (TWO) r1 = fmaf(D, c1, r1);
    };
    constexpr int kBatch = 8;
    float n0[kBatch], n1[kBatch];
    auto load = [&](int l0) {
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            n0[u] = (Q == 0) ? grow[(l0 + u) * B + lane] : 0.f;
            n1[u] = TWO ? grow[(l0 + u) * B + 64 + lane] : 0.f;
        }
    };
    int l = 0;
    if (nsteps >= kBatch) load(0);
#pragma unroll 1
    for (; l + kBatch <= nsteps; l += kBatch) {
        float c0[kBatch], c1[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) { c0[u] = n0[u]; c1[u] = n1[u]; }
        if (l + 2 * kBatch <= nsteps) load(l + kBatch);                    // the next batch's rows: in flight during this one
#pragma unroll
        for (int u = 0; u < kBatch; ++u) step(l + u, c0[u], c1[u]);
    }
#pragma unroll 1
    for (; l < nsteps; ++l) step(l, (Q == 0) ? grow[l * B + lane] : 0.f, TWO ? grow[l * B + 64 + lane] : 0.f);
}

// ---- DENSE blocks of 256 / 512 markers (every marker of the block is included whatever its rhs: Pi = 0, RR-BLUP, BayesA,
// the reference's own benchmark setting).  The block chain is a forward substitution: marker c needs the changes of all
// markers before it.  Section s (64 markers) is walked by wave s exactly as dense_section walks a small block -- same
// operations, same order, bit-identical to the sequential chain -- with its 64 x 64 DIAGONAL Gram tile from LDS (all
// tiles are fetched with direct loads at the very start of the launch); everything off the diagonal runs in parallel:
// thread c owns row c (its running rhs in a register) and column c of the next block's lookahead correction, and after
// section s is done applies its 64 changes from Gram / cross-Gram values it prefetched into registers while the section
// was being walked (rhs = fmaf(D_k, G[k][c], rhs) in marker order: the sequential chain's own fmaf sequence).
// One barrier per section; the serial part per marker is dense_section's chain and nothing else.
template <int METHOD>
__device__ __forceinline__ void dense_big_st(char* smem, const StepSmem& SM, const SamplerArgs& A, float ie, long long tk0)
{
    const int B = SM.B, b = A.b, bn = A.b_next;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t j0 = A.j0;
    float* rhs_lds = reinterpret_cast<float*>(smem + SM.rhs_off);      // entry rhs; reused as D[c] = alpha_old - alpha_new once c is done
    float* acur = reinterpret_cast<float*>(smem + SM.acur_off);
    const float* astart = reinterpret_cast<const float*>(smem + SM.astart_off);
    const double* lpd = reinterpret_cast<const double*>(smem + SM.prepd_off);
    const float* lpf = reinterpret_cast<const float*>(smem + SM.prepf_off);
    const float* tiles = reinterpret_cast<const float*>(smem + SM.rows_off);      // [B/64][64][64] diagonal Gram tiles
    int* wcnt = reinterpret_cast<int*>(smem + SM.wcnt_off);
    const int c = tid;
    const bool own = c < b;
    const int cl = own ? c : 0;
    float rr = rhs_lds[cl];
    const float il = lpf[cl], dj = lpf[2 * B + cl];
    const float kc1 = lpf[B + cl], kc0 = lpf[3 * B + cl];               // Rule D: alpha_new = fmaf(kc1, x, kc0)  (see the front)
    const double zs = lpd[cl];
    const float ao = acur[cl];
    const float da = dj * ao;                                           // d * alpha_old (BayesABC.jl:36)
    float an_own = 0.f;
    const int nsec = b >> 6;
    const bool has_col = tid < bn;
    float corr = 0.f;
    float gq[64], cq[64];
    // thread c reads column c of the Gram rows / cross-Gram rows of a section: one dword per lane, coalesced (256 bytes = two
    // lines per wave instruction).  (Reading the thread's own ROW instead -- G is symmetric -- as 16 dwordx4 loads was
    // measured 3x slower: 64 different lines per wave instruction keep the texture addresser busy ~360 cycles each.)
    const float* gcol = A.gram + cl;                                    // G[k][c] = gram[k * b + c]
    const float* ccol = A.cross_next + (has_col ? tid : 0);             // C[k][c'] = cross[k * bn + c']
    auto load_g = [&](int s) {
#pragma unroll
        for (int u = 0; u < 64; ++u) gq[u] = gcol[(64 * s + u) * b];
    };
    auto load_c = [&](int s) {
#pragma unroll
        for (int u = 0; u < 64; ++u) cq[u] = ccol[(64 * s + u) * bn];
    };
    // Barrier of the section loop: LDS traffic only.  (__syncthreads() also waits for every outstanding GLOBAL load -- the
    // prefetches below are meant to stay in flight across it.)
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto prefetch = [&](int sec) {
        if (wave > sec && own) load_g(sec);
        if (has_col) load_c(sec);
    };
    prefetch(0);
#pragma unroll 1
    for (int s = 0; s < nsec; ++s) {
        if (wave == s) {
            float r0 = rr, r1 = 0.f, rev = rr;
            dense_section<0, false, true, true>(tiles + s * 4096, 64, 64, lane, ie, 0.f, 0.f, il, da, ao, zs, r0, r1, rev, kc1, kc0);
            an_own = fmaf(kc1, rev, kc0);
            acur[c] = an_own;
            rhs_lds[c] = ao - an_own;                                   // D of this marker, read by everybody after the barrier
        }
        lds_barrier();
        // (Keep this body simple: a variant in which the next walker skipped the correction and caught up in a loop after its
        // walk made the register allocator keep several copies of the prefetch arrays alive -- 218 spilled VGPRs.)
        // The section's 64 changes first (16 broadcast reads back to back: one LDS latency instead of one per group of four),
        // then the two fmaf chains in marker order, interleaved
        float dv[64];
        {
            const float4* Dv4 = reinterpret_cast<const float4*>(rhs_lds + 64 * s);
#pragma unroll
            for (int u = 0; u < 16; ++u) { const float4 d = Dv4[u]; dv[4 * u] = d.x; dv[4 * u + 1] = d.y; dv[4 * u + 2] = d.z; dv[4 * u + 3] = d.w; }
        }
        // the wave that just walked did not prefetch its cross-Gram values (the issue sat between the barrier and its walk,
        // on the critical path): it fetches them now -- its rows are done, nobody waits for it
        if (wave == s && has_col && s > 0) load_c(s);
        const bool do_r = wave > s && own;
        if (do_r && has_col) {
#pragma unroll
            for (int u = 0; u < 64; ++u) { rr = fmaf(dv[u], gq[u], rr); corr = fmaf(dv[u], cq[u], corr); }
        } else if (do_r) {
#pragma unroll
            for (int u = 0; u < 64; ++u) rr = fmaf(dv[u], gq[u], rr);
        } else if (has_col) {
#pragma unroll
            for (int u = 0; u < 64; ++u) corr = fmaf(dv[u], cq[u], corr);
        }
        if (s + 1 < nsec) {
            if (wave > s + 1 && own) load_g(s + 1);
            if (has_col && wave != s + 1) load_c(s + 1);                // (the next walker: see above)
        }
    }
    __syncthreads();
    const long long tk4 = clock64();
    // the block's change list in marker order (an effect that came out bit-equal to the old one is no change)
    const bool changed = own && (ao != an_own);
    const unsigned long long cm = __ballot(changed);
    __syncthreads();                                                    // (the candidate counts of the front are no longer read)
    if (lane == 0) wcnt[wave] = __popcll(cm);
    __syncthreads();
    int base = 0, nfin = 0;
#pragma unroll
    for (int q = 0; q < kStepThreads / 64; ++q) { const int v = wcnt[q]; base += (q < wave) ? v : 0; nfin += v; }
    // ---- global stores last
    if (tid < B && bn > 0) A.corr_out[tid] = has_col ? corr : 0.f;
    if (changed) {
        const int e = base + __popcll(cm & ((1ull << lane) - 1ull));
        const float d = ao - an_own;
        A.ev_out->idx[e] = (int32_t)(j0 + c);
        A.ev_out->delta[0][e] = d;
        if (e < 7) { A.ev_out->hidx[e] = (int32_t)(j0 + c); A.ev_out->hdelta[e] = d; }
        A.alpha[j0 + c] = an_own;
    }
    if (own) { A.beta[j0 + c] = an_own; reinterpret_cast<float*>(A.delta)[j0 + c] = 1.f; }
    if (tid == 0) {
        A.ev_out->count = nfin;
        atomicAdd(&A.counters[0], (unsigned long long)nfin);
        atomicAdd(&A.counters[5], (unsigned long long)(tk4 - tk0));      // (diagnostics: front + walk)
        atomicAdd(&A.counters[7], (unsigned long long)b);
    }
    (void)astart;
}

__host__ __device__ constexpr int st_park_nd(int method) { return method == kBayesR ? BayesRMarker::kFastD : 1; }
__host__ __device__ constexpr int st_park_nf(int method, bool dense = false) { (void)dense; return method == kBayesR ? 2 : 5; }

// DENSE: the instantiation for sweeps under a UNIFORM PRIOR pi = 0 (single-trait BayesA/B/C: RR-BLUP, BayesA, BayesL, the
// reference's benchmark setting), selected by the host: every marker follows Rule D (AbcMarker::rule_d) on every path of
// it, and full 256- / 512-marker blocks take dense_big_st.  The steady-state kernel is compiled without any of it.
template <int METHOD, bool DENSE = false>
__device__ __forceinline__ void sampler_role_st(char* smem, const SamplerArgs& A)
{
    constexpr bool kR = (METHOD == kBayesR);
    constexpr int ND = st_park_nd(METHOD), NF = st_park_nf(METHOD, DENSE);
    const StepSmem SM(A.bsz, 1, ND, NF);
    const int B = SM.B;
    const DevParams* P = A.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = A.b;
    const int64_t j0 = A.j0, p = A.p;
    const float ie = 1.0f / P->vare[0];
    double* lpd = reinterpret_cast<double*>(smem + SM.prepd_off);     // [ND][B]
    float* lpf = reinterpret_cast<float*>(smem + SM.prepf_off);       // [NF][B]
    float* rhs_lds = reinterpret_cast<float*>(smem + SM.rhs_off);
    float* acur = reinterpret_cast<float*>(smem + SM.acur_off);
    float* astart = reinterpret_cast<float*>(smem + SM.astart_off);
    float* bpark0 = reinterpret_cast<float*>(smem + SM.bcur_off);      // [B] beta / [B] delta of the block (single trait:
    float* dpark0 = reinterpret_cast<float*>(smem + SM.dcur_off);      // the multi-trait slots are free)
    const long long tk0 = clock64();

    // ---- phase A (all threads, ONE memory latency): for its marker every thread issues, back to back, the
    // loads of alpha, the sweep constants, the row-group partials and the lookahead correction; then
    //   rhs = fl32(sum of partials) + corr
    // and decides candidacy (does the effect change if evaluated against the entry rhs?) with the marker's
    // thresholds: two float compares.  Under full-rate streaming by the update role a dependent global load costs
    // microseconds, so nothing here waits twice.
    // Small blocks (B <= 128: the host's choice for dense priors): the whole Gram block fits the row slots, and it does
    // not depend on anything this launch computes -- fetch it with the very first loads instead of after the candidates
    // are known (one dependent memory latency less per block).  Slot of marker c = c.
    const bool prestage = (B <= 128) && (B <= SM.max_cand);
    // full blocks: the Gram block (and the cross-Gram rows X_this'X_next, when they have their own LDS room and the next
    // block is full too) go straight to LDS with direct loads issued before anything else: a handful of instructions
    // instead of ~200 lines of cold unrolled code (instruction fetch after a dispatch runs at memory latency)
    const bool gram_dma = prestage && b == B;
    const bool cross_dma = gram_dma && SM.has_cross && A.b_next == B;
    if (gram_dma) dma_copy_to_lds(A.gram, reinterpret_cast<float*>(smem + SM.rows_off), B * B);
    // 256- / 512-marker blocks under a prior that includes every marker (Pi = 0 without per-marker pi: known before any
    // load): dense_big_st.  Its diagonal Gram tiles (64 x 64 floats per 64-marker section: wave w fetches tile w) go to
    // LDS with direct loads issued before anything else; whether every marker really is "always included" (thresholds
    // lo = hi) is voted below, and a block that fails the vote runs the general path (which re-stages the row slots).
    bool dense_big_try = false;
    if constexpr (!kR && DENSE) {
        dense_big_try = (B == 256 || B == 512) && b == B && (A.b_next == 0 || A.b_next == B) &&
                        (P->nreps == 1) && P->pi == 0.0 && P->pi_vec == nullptr && !A.dense_big_off;
        if (dense_big_try && wave < (B >> 6)) {
            typedef __attribute__((address_space(3))) void lds_void;
            float* tile = reinterpret_cast<float*>(smem + SM.rows_off) + wave * 4096;
            const float* src = A.gram + (int64_t)(64 * wave + (lane >> 4)) * b + 64 * wave + (lane & 15) * 4;
#pragma unroll 1
            for (int r4 = 0; r4 < 16; ++r4)       // 4 rows of 64 floats per instruction: lane -> row lane / 16, float4 column lane % 16
                __builtin_amdgcn_global_load_lds(src + (int64_t)(4 * r4) * b, (lds_void*)(tile + r4 * 256), 16, 0, 0);
        }
    }
    bool always_mine = true;
    // (the cross-Gram rows are only needed after the walk: waves 1..7 fetch them while wave 0 walks)
    float4 gpre[8];
    if (prestage && !gram_dma) {
        // B*B/4 float4 elements over 512 threads: <= 8 per thread; element e -> row e / (B/4), float4 column e % (B/4)
        const int per_row = B >> 2, total = b * per_row;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + u * kStepThreads;
            const int ec = e < total ? e : 0;
            const int row = ec / per_row, c4 = (ec - row * per_row) * 4;
            // rows are b floats apart in global memory (b may be < B for the last block): element-wise clamped loads
            const float* src = A.gram + (int64_t)row * b;
            if (b == B) gpre[u] = *reinterpret_cast<const float4*>(src + c4);       // full block: rows are 16-byte aligned
            else {
                gpre[u].x = src[c4 < b ? c4 : 0]; gpre[u].y = src[c4 + 1 < b ? c4 + 1 : 0];
                gpre[u].z = src[c4 + 2 < b ? c4 + 2 : 0]; gpre[u].w = src[c4 + 3 < b ? c4 + 3 : 0];
            }
        }
    }
    bool cand[2] = {false, false};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = tid + q * kStepThreads;
        if (c >= B) continue;                     // (B may be smaller than the workgroup)
        const int cc = c < b ? c : 0;
        const int64_t j = j0 + cc;
        const float a0 = A.alpha[j];
        const float dj = A.xpx[j];
        const float co = A.corr_in[c];
        if constexpr (kR) {
            BayesRMarker bm;
            bm.load_fast_global(A.prep_d, p, j, dj, ie);
            const float thrx = A.prep_f[j];
            const double sum = sum_partials(A.partials + cc, A.nrg, A.bstride);
            const float rhs0 = (float)sum + co;
            rhs_lds[c] = rhs0;
            const float a_in = (c < b) ? a0 : 0.f;
            acur[c] = a_in; astart[c] = a_in;
            bm.store_fast(lpd, B, c);
            lpf[c] = dj; lpf[B + c] = thrx;
            cand[q] = (c < b) && ((a_in != 0.f) || (fabsf(rhs0) >= thrx));
            dpark0[c] = 1.f;                      // a marker that stays out: class 1 (see the prefix skip below)
        } else {
            const double zs = A.prep_d[3 * p + j];
            const float invLhs = A.prep_f[j], bex = A.prep_f[2 * p + j], lo = A.prep_f[3 * p + j], hi = A.prep_f[4 * p + j];
            const double sum = sum_partials(A.partials + cc, A.nrg, A.bstride);
            const float rhs0 = (float)sum + co;       // + lookahead correction formed by the previous block's sampler
            rhs_lds[c] = rhs0;
            const float a_in = (c < b) ? a0 : 0.f;
            acur[c] = a_in; astart[c] = a_in;
            lpd[c] = zs;
            lpf[c] = invLhs; lpf[B + c] = bex; lpf[2 * B + c] = dj; lpf[3 * B + c] = lo; lpf[4 * B + c] = hi;
            // Rule D sweeps: nothing is ever excluded, so the slots of the "excluded" draw and of the lower threshold carry
            // c1 and c0 instead (512-marker blocks leave no LDS for two more rows next to dense_big_st's 128 KB of tiles)
            if constexpr (DENSE) { lpf[B + c] = A.prep_f[5 * p + j]; lpf[3 * B + c] = A.prep_f[6 * p + j]; }
            cand[q] = (c < b) && ((a_in != 0.f) || abc_included(rhs0, lo, hi));
            always_mine = always_mine && ((c >= b) || (lo == hi));           // thresholds(): lo = hi <=> always included
            bpark0[c] = bex; dpark0[c] = 0.f;     // a marker that stays out: delta 0, beta = its excluded draw
        }
    }
    if (prestage) {
        float* rows_p = reinterpret_cast<float*>(smem + SM.rows_off);
        short* slot_p = reinterpret_cast<short*>(smem + SM.slot_off);
        short* cand_p = reinterpret_cast<short*>(smem + SM.cand_off);
        if (!gram_dma) {
            const int per_row = B >> 2, total = b * per_row;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + u * kStepThreads;
                if (e < total) {
                    const int row = e / per_row, c4 = (e - row * per_row) * 4;
                    *reinterpret_cast<float4*>(rows_p + row * B + c4) = gpre[u];
                }
            }
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the direct loads have landed (barrier below)
        for (int c = tid; c < B; c += kStepThreads) { slot_p[c] = (short)(c < b ? c : -1); cand_p[c] = (short)c; }
    }
    // number of markers whose effect changes against the entry rhs (block-wide count through the wave-count slots;
    // __syncthreads_count would add static LDS on top of the 160 KB dynamic carve)
    {
        int* wc = reinterpret_cast<int*>(smem + SM.wcnt_off);
        const unsigned long long mb = __ballot(cand[0] || cand[1]);
        // bits 16 / 17: this wave's markers of sub-block `wave` / `8 + wave` contain a candidate
        const int f0 = __any(cand[0]) ? 1 << 16 : 0, f1 = __any(cand[1]) ? 1 << 17 : 0;     // (votes outside the lane-0 branch)
        const int f2 = __all(always_mine) ? 1 << 18 : 0;                 // bit 18: every marker of this wave is always included
        if (lane == 0) wc[wave] = __popcll(mb) | f0 | f1 | f2;
    }
    if (dense_big_try) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the diagonal tiles have landed (barrier below)
    __syncthreads();
    int ncand_all = 0;
    // PREFIX SKIP: until the first candidate of the block commits, the running rhs IS the entry rhs, so the evaluation
    // every thread just did is final for all markers before it -- they stay out of the model (beta / delta parked
    // above) and the serial wave starts at the first sub-block that holds a candidate.  With a sparse prior that is
    // half of the sub-blocks on average, and all of them in the blocks without a candidate.
    int first_sub = 16;
    {
        const int* wc = reinterpret_cast<const int*>(smem + SM.wcnt_off);
        unsigned mask = 0u;
#pragma unroll
        for (int q = 0; q < kStepThreads / 64; ++q) {
            const int v = wc[q];
            ncand_all += v & 0xffff;
            mask |= ((v >> 16) & 1u) << q | ((v >> 17) & 1u) << (8 + q);
        }
        if (mask) first_sub = __builtin_ctz(mask);
    }
    if constexpr (!kR && DENSE) {
        if (dense_big_try) {
            const int* wc = reinterpret_cast<const int*>(smem + SM.wcnt_off);
            bool all_in = true;
#pragma unroll
            for (int q = 0; q < kStepThreads / 64; ++q) all_in = all_in && ((wc[q] >> 18) & 1);
            if (all_in) { dense_big_st<METHOD>(smem, SM, A, ie, tk0); return; }
        }
    }
    // single-pass sweeps with a next block: waves 1..4 accumulate its lookahead correction while the serial wave runs
    const bool stream_corr = ((P->nreps > 0 ? P->nreps : b) == 1) && !prestage && A.b_next > 0;
    if (tid == 0) { int* wc0 = reinterpret_cast<int*>(smem + SM.wcnt_off); wc0[12] = 0; wc0[13] = 0; wc0[14] = 0; }
    if (stream_corr) for (int c = tid; c < B; c += kStepThreads) reinterpret_cast<int2*>(smem + SM.log_off)[c] = make_int2(-1, 0);
    __syncthreads();                               // (stage_rows reuses the slots)
    const long long tk1 = clock64();
    const long long tk2 = clock64();
    // (a block without any candidate -- about a third of them with a sparse prior -- has nothing to stage or to walk)
    long long tss[3] = {tk2, tk2, tk2};
    // (a block without any candidate has nothing to stage or to walk -- in a SINGLE pass.  With within-block repetitions a
    // later repetition draws anew and may move a marker that was no candidate at entry: stage_rows must then have marked
    // every marker "not staged" (slot -1), or the winner's row would be looked up through a stale slot.)
    const bool single_pass_st = (P->nreps > 0 ? P->nreps : b) == 1;
    int nstaged = prestage ? b : ((first_sub >= 16 && single_pass_st) ? 0 : stage_rows(smem, SM, A, cand, tss));
    const bool cross_lds = prestage && SM.has_cross;
    float4 corr_mine{0.f, 0.f, 0.f, 0.f};
    if (stream_corr) {
        if (is_corr_helper(wave)) corr_mine = stream_corr_role(smem, SM, A);      // returns when the serial wave is done
        else if (wave >= 5) {                                           // waves 5, 6, 7
            const int* stop = reinterpret_cast<const int*>(smem + SM.wcnt_off) + 13;
            prefetch_cross_rows(smem, SM, A, nstaged, 5, stop);         // the helpers' loads become L2 hits
            prefetch_next_gram(A, prestage, 5, stop);
        }
    } else {
        if (cross_lds) {                                     // waves 1..7, while wave 0 runs the serial phase
            if (cross_dma) {
                dma_copy_to_lds(A.cross_next, reinterpret_cast<float*>(smem + SM.cross_off), B * B, 1);
                if (wave != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // landed before the barrier after the walk
            } else copy_cross_rows(smem, SM, A);
        }
        else prefetch_cross_rows(smem, SM, A, nstaged);      // waves 1..7: pull the rows into L2 for corr_phase at the end
        prefetch_next_gram(A, prestage);                     // waves 1..7: the next block's staging becomes an L2 hit
    }
    int* wcnt_s = reinterpret_cast<int*>(smem + SM.wcnt_off);
    long long tk3 = 0, tk4 = 0, tk5 = 0;
    int nrounds = 0, nslow = 0;
    if (wave == 0) {
    tk3 = clock64();

    // wave 0: lane l owns marker c = 64*s + l of sub-block s
    const short* slot_of = reinterpret_cast<const short*>(smem + SM.slot_off);
    const float* rows = reinterpret_cast<const float*>(smem + SM.rows_off);
    int2* evlog = reinterpret_cast<int2*>(smem + SM.log_off);
    float* bpark = bpark0;
    float* dpark = dpark0;
    const int nsub = (b + 63) / 64;
    const int nreps = P->nreps > 0 ? P->nreps : b;
    const bool lazy = (nreps == 1);     // single pass: corrections reach a sub-block when it becomes active
    RngKey key{P->seed_lo, P->seed_hi, P->iter, 0u};
    int nlog = 0;

    // The serial wave reads Gram rows from LDS ONLY (a value that may come from LDS or global compiles to
    // flat loads whose vmcnt(0) wait also drains the prefetch of the next sub-block).  A committed change
    // whose row was not staged is copied into a free slot first; when the slots are exhausted it goes to the
    // overflow row and is applied to the remaining sub-blocks at once instead of being logged.
    float* rows_w = reinterpret_cast<float*>(smem + SM.rows_off);
    auto fetch_row = [&](int ce, int slot) {
        const float* grow = A.gram + (int64_t)ce * b;
        for (int c0 = 0; c0 < B; c0 += 512) {          // 8 loads in flight per lane
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int col = c0 + 64 * u + lane; v[u] = grow[col < b ? col : 0]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int col = c0 + 64 * u + lane; if (col < B) rows_w[slot * B + col] = v[u]; }
        }
    };

    // ---- DENSE blocks (most markers of the block are candidates: Pi = 0, BayesA, the reference benchmark's setting):
    // speculation buys nothing -- every round commits exactly one marker -- so the wave walks the block sequentially.
    // Every lane evaluates ITS OWN marker against its own running rhs at every step (two float compares for the
    // decision, six operations for the new effect: no operand is broadcast); the step's marker is lane l, whose
    // alpha_old - alpha_new is broadcast with ONE v_readlane and applied to the running rhs of the whole block (two
    // registers per lane) with the marker's Gram row from LDS (all rows are staged; the read is issued a step ahead).
    // A lane's result is final at its own step: it keeps the rhs it was evaluated against and recomputes its update
    // after the walk.  Same arithmetic, same order, same results as the speculative rounds;
    // ~18 instructions per marker on a dependent chain of 11.
    bool dense_done = false;
    if constexpr (!kR) {
        if (nreps == 1 && prestage && nstaged == b && 5 * ncand_all >= 3 * b) {
            float lo[2], hi[2], il[2], da[2], ao[2], rhsq[2], rev[2], bex[2], kc1[2] = {0.f, 0.f}, kc0[2] = {0.f, 0.f};
            double zs[2];
            bool always = true;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = (64 * q + lane < B) ? 64 * q + lane : 0;
                il[q] = lpf[c]; bex[q] = lpf[B + c]; lo[q] = lpf[3 * B + c]; hi[q] = lpf[4 * B + c];
                if constexpr (DENSE) { kc1[q] = lpf[B + c]; kc0[q] = lpf[3 * B + c]; }
                zs[q] = lpd[c];
                rhsq[q] = rhs_lds[c]; ao[q] = acur[c]; rev[q] = rhsq[q];
                da[q] = lpf[2 * B + c] * ao[q];                                       // d * alpha_old (BayesABC.jl:36)
                always = always && (DENSE || lo[q] == hi[q]);                         // thresholds(): lo = hi <=> always included
            }
            // Pi = 0 / BayesA / RR-BLUP: every marker of the block is included whatever its rhs -- no decision on the chain
            const bool all_in = __all(always);
            if constexpr (DENSE) {
                // Rule D (uniform pi = 0): every marker is always included and its new effect is fmaf(kc1, x, kc0)
                if (B > 64) {
                    dense_section<0, true, true, true>(rows, B, b < 64 ? b : 64, lane, ie, lo[0], hi[0], il[0], da[0], ao[0], zs[0], rhsq[0], rhsq[1], rev[0], kc1[0], kc0[0]);
                    if (b > 64) dense_section<1, true, true, true>(rows + 64 * B, B, b - 64, lane, ie, lo[1], hi[1], il[1], da[1], ao[1], zs[1], rhsq[0], rhsq[1], rev[1], kc1[1], kc0[1]);
                } else dense_section<0, false, true, true>(rows, B, b, lane, ie, lo[0], hi[0], il[0], da[0], ao[0], zs[0], rhsq[0], rhsq[1], rev[0], kc1[0], kc0[0]);
            } else
            if (B > 64) {
                if (all_in) {
                    dense_section<0, true, true>(rows, B, b < 64 ? b : 64, lane, ie, lo[0], hi[0], il[0], da[0], ao[0], zs[0], rhsq[0], rhsq[1], rev[0]);
                    if (b > 64) dense_section<1, true, true>(rows + 64 * B, B, b - 64, lane, ie, lo[1], hi[1], il[1], da[1], ao[1], zs[1], rhsq[0], rhsq[1], rev[1]);
                } else {
                    dense_section<0, true, false>(rows, B, b < 64 ? b : 64, lane, ie, lo[0], hi[0], il[0], da[0], ao[0], zs[0], rhsq[0], rhsq[1], rev[0]);
                    if (b > 64) dense_section<1, true, false>(rows + 64 * B, B, b - 64, lane, ie, lo[1], hi[1], il[1], da[1], ao[1], zs[1], rhsq[0], rhsq[1], rev[1]);
                }
            } else {
                if (all_in) dense_section<0, false, true>(rows, B, b, lane, ie, lo[0], hi[0], il[0], da[0], ao[0], zs[0], rhsq[0], rhsq[1], rev[0]);
                else dense_section<0, false, false>(rows, B, b, lane, ie, lo[0], hi[0], il[0], da[0], ao[0], zs[0], rhsq[0], rhsq[1], rev[0]);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = 64 * q + lane;
                const bool inc = DENSE ? true : abc_included(rev[q], lo[q], hi[q]);
                const float an = DENSE ? fmaf(kc1[q], rev[q], kc0[q]) : dense_alpha_new(rev[q], da[q], ie, il[q], zs[q], inc);
                if (c < B) {
                    acur[c] = (c < b) ? an : 0.f; bpark0[c] = inc ? an : bex[q]; dpark0[c] = inc ? 1.f : 0.f;
                    rhs_lds[c] = (c < b) ? ao[q] - an : 0.f;                          // alpha_old - alpha_new, for the dense correction
                }
            }
            if (lane == 0) wcnt_s[14] = 1;
            nrounds += b;
            dense_done = true;
        }
    }

    const int s_first = (lazy && !dense_done) ? (first_sub < nsub ? first_sub : nsub) : 0;      // prefix skip (single pass only)

    // ---- SINGLE PASS (nreps = 1: the exact non-block chain; the hot path).  Everything comes from LDS; the log of
    // committed changes {row offset, D} lives in two VGPRs (lane e = entry e, written with v_writelane, read back with
    // v_readlane), so bringing a later sub-block up to date costs one LDS read per entry and no dependent second one.
    if (lazy && !dense_done) {
        int2* plog = reinterpret_cast<int2*>(smem + SM.log_off);
        int npub = 0;
        int log_off = 0;            // lane e: sl*B of entry e
        float log_D = 0.f;          // lane e: alpha_old - alpha_new of entry e
        // apply the logged changes (in commit order) to the rhs of the sub-blocks after `s` and empty the log: needed before
        // a change is applied eagerly (log full, or a row that only lives in the overflow slot) so that every rhs element
        // still sees its corrections in commit order
        auto flush_log = [&](int s) {
            for (int s2 = s + 1; s2 < nsub; ++s2) {
                const int c2 = 64 * s2 + lane;
                float r2 = rhs_lds[c2];
                for (int e = 0; e < nlog; ++e)
                    r2 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(log_D), e)),
                              rows[__builtin_amdgcn_readlane(log_off, e) + c2], r2);
                rhs_lds[c2] = r2;
            }
            nlog = 0;
        };
#pragma unroll 1
        for (int s = s_first; s < nsub; ++s) {
            const int c = 64 * s + lane;
            const bool valid = c < b;
            const int cl = valid ? c : 0;
            const float a_cur = acur[c];                  // (fixed while the marker is pending; a committed lane leaves `pending`)
            float rhs = rhs_lds[c];
            const int my_slot = slot_of[c];
            float c_lo = 0.f, c_hi = 0.f, c_il = 0.f, c_bex = 0.f, c_d = 0.f, c_thrx = 0.f, c_k1 = 0.f, c_k0 = 0.f;
            double c_zs = 0.0;
            double r_il1 = 0.0, r_il2 = 0.0, r_il3 = 0.0, r_zs1 = 0.0, r_zs2 = 0.0, r_zs3 = 0.0, r_T0 = 0.0, r_T1 = 0.0, r_T2 = 0.0;
            if constexpr (kR) {
                c_d = lpf[cl]; c_thrx = lpf[B + cl];
                r_il1 = lpd[0 * B + cl]; r_il2 = lpd[1 * B + cl]; r_il3 = lpd[2 * B + cl];
                r_zs1 = lpd[3 * B + cl]; r_zs2 = lpd[4 * B + cl]; r_zs3 = lpd[5 * B + cl];
                r_T0 = lpd[6 * B + cl]; r_T1 = lpd[7 * B + cl]; r_T2 = lpd[8 * B + cl];
            } else {
                c_il = lpf[cl]; c_bex = lpf[B + cl]; c_d = lpf[2 * B + cl]; c_lo = lpf[3 * B + cl]; c_hi = lpf[4 * B + cl];
                c_zs = lpd[cl];
                if constexpr (DENSE) { c_k1 = lpf[B + cl]; c_k0 = lpf[3 * B + cl]; }
            }
            // bring this sub-block up to date: the changes committed so far, in commit order (the same fmaf sequence
            // per element as an immediate update); 8 independent LDS reads in flight
            for (int e0 = 0; e0 < nlog; e0 += 8) {
                float gv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ee = e0 + u < nlog ? e0 + u : nlog - 1;
                    gv[u] = rows[__builtin_amdgcn_readlane(log_off, ee) + c];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u < nlog) rhs = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(log_D), e0 + u)), gv[u], rhs);
            }
            unsigned long long pending = __ballot(valid);
            const bool nz = a_cur != 0.f;
            // speculative rounds: every pending lane tests ITS marker against the current rhs (float compares only);
            // the first lane whose effect changes commits, the rest are re-tested after its Gram row corrected the rhs.
            // Lanes before the winner stay out of the model with the values parked for them; only the winner writes.
            while (true) {
                ++nrounds;
                bool inc = false, ev = false;
                float an = 0.f;
                if constexpr (kR) ev = nz || (fabsf(rhs) >= c_thrx);
                else {
                    inc = DENSE ? true : abc_included(rhs, c_lo, c_hi); ev = inc || nz;
                    // every lane's new effect BEFORE the vote: the six dependent operations run beside the vote's
                    // compare / ballot / find-first chain instead of after it (wasted only in a sub-block's last round)
                    if constexpr (DENSE) an = fmaf(c_k1, rhs, c_k0);         // Rule D (uniform pi = 0: inc is always true)
                    else an = abc_alpha_new(rhs, a_cur, c_d, ie, c_il, c_zs, inc);
                    asm volatile("" : "+v"(an));         // (keeps the compiler from sinking it below the vote's branch)
                }
                const unsigned long long m = __builtin_amdgcn_ballot_w64(ev) & pending;
                if (m == 0ull) break;                     // no further change in this sub-block
                const int k = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
                if constexpr (kR) {
                    bool sure = true;
                    int cls = bayesr_eval_thr(rhs, a_cur, ie, c_d, r_il1, r_il2, r_il3, r_zs1, r_zs2, r_zs3, r_T0, r_T1, r_T2, an, sure);
                    if (__builtin_amdgcn_readlane(sure ? 0 : 1, k)) {       // (practically never: s within 1e-6 of a class threshold)
                        BayesRMarker bm;
                        bm.load(A.prep_d, A.prep_f, p, j0 + cl, c_d, ie);   // full constants from global
                        cls = bm.evaluate(rhs, a_cur, ie, an);
                        ++nslow;
                    }
                    if (cls == 0) an = 0.f;
                    if (lane == k) { acur[c] = an; dpark[c] = (float)(cls + 1); }             // stored as class 1..4
                } else {
                    if (lane == k) acur[c] = an;          // (beta / delta follow from alpha at the end: derive_bd)
                }
                const float Dl = a_cur - an;
                pending = (k == 63) ? 0ull : (pending & ~((2ull << k) - 1ull));
                const float D = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Dl), k));
                if (D != 0.f) {
                    // the block's change list {local column, alpha_old - alpha_new}: marker order = commit order; read by
                    // the correction helpers while it grows and by the final stores
                    if (lane == 0) plog[npub] = make_int2(64 * s + k, __float_as_int(D));
                    ++npub;
                    // rhs += D * G[ce][:] (BayesABC.jl:169,172).  The active sub-block is corrected in its register
                    // copy -- the only value the next round waits for.
                    int sl = __builtin_amdgcn_readlane(my_slot, k);
                    bool overflow = false;
                    if (sl < 0) {                                            // rare: not staged at entry
                        if (nstaged < SM.max_cand) sl = nstaged++; else { sl = SM.max_cand; overflow = true; }
                        fetch_row(64 * s + k, sl);
                        if (lane == 0) atomicAdd(&A.counters[1], 1ull);      // diagnostic
                    }
                    const int off = sl * B;
                    rhs = fmaf(D, rows[off + c], rhs);
                    if (overflow || nlog >= 64) {
                        flush_log(s);
                        for (int s2 = s + 1; s2 < nsub; ++s2) {
                            const int c2 = 64 * s2 + lane;
                            rhs_lds[c2] = fmaf(D, rows[off + c2], rhs_lds[c2]);
                        }
                    } else {
                        log_off = (lane == nlog) ? off : log_off;
                        log_D = (lane == nlog) ? D : log_D;
                        ++nlog;
                    }
                }
                if (pending == 0ull) break;
            }
        }
        if (lane == 0) {
            wcnt_s[11] = npub;                                              // (read after the barrier)
            __hip_atomic_store(&wcnt_s[13], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);     // the helpers may finish
        }
    }

    for (int rep = 0; rep < ((dense_done || lazy) ? 0 : nreps); ++rep) {
        key.rep = (uint32_t)rep;
#pragma unroll 1
        for (int s = s_first; s < nsub; ++s) {
            const int c = 64 * s + lane;
            const bool valid = c < b;
            const int cl = valid ? c : 0;
            const int64_t j = j0 + cl;
            const uint32_t marker = P->marker0 + (uint32_t)j;
            unsigned long long pending = __ballot(valid);
            const float a_cur = acur[c];                  // (fixed while the marker is pending; a committed lane leaves `pending`)
            float rhs = rhs_lds[c];                       // register copy of the active sub-block's rhs
            const int my_slot = slot_of[c];
            if (lazy) {
                // bring this sub-block up to date: the changes committed so far, in commit order
                // (same fmaf sequence per entry as an immediate update)
                // two-phase chunks so the dependent LDS reads (log entry -> row element) are pipelined
                for (int e0 = 0; e0 < nlog; e0 += 8) {
                    int2 le[8];
                    float gv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) le[u] = evlog[e0 + u < nlog ? e0 + u : nlog - 1];
#pragma unroll
                    for (int u = 0; u < 8; ++u) gv[u] = rows[le[u].x * B + c];
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (e0 + u < nlog) rhs = fmaf(__int_as_float(le[u].y), gv[u], rhs);
                }
            }

            // the marker's constants: parked in LDS by the parallel phase (rep 0) or recomputed for a later repetition
            float c_lo = 0.f, c_hi = 0.f, c_il = 0.f, c_bex = 0.f, c_d = 0.f, c_thrx = 0.f, c_k1 = 0.f, c_k0 = 0.f;
            double c_zs = 0.0;
            BayesRMarker bm;
            double r_il1 = 0.0, r_il2 = 0.0, r_il3 = 0.0, r_zs1 = 0.0, r_zs2 = 0.0, r_zs3 = 0.0, r_T0 = 0.0, r_T1 = 0.0, r_T2 = 0.0;
            if (rep == 0) {
                if constexpr (kR) {
                    c_d = lpf[cl]; c_thrx = lpf[B + cl];
                    r_il1 = lpd[0 * B + cl]; r_il2 = lpd[1 * B + cl]; r_il3 = lpd[2 * B + cl];
                    r_zs1 = lpd[3 * B + cl]; r_zs2 = lpd[4 * B + cl]; r_zs3 = lpd[5 * B + cl];
                    r_T0 = lpd[6 * B + cl]; r_T1 = lpd[7 * B + cl]; r_T2 = lpd[8 * B + cl];
                } else {
                    c_il = lpf[cl]; c_bex = lpf[B + cl]; c_d = lpf[2 * B + cl]; c_lo = lpf[3 * B + cl]; c_hi = lpf[4 * B + cl];
                    c_zs = lpd[cl];
                    if constexpr (DENSE) { c_k1 = lpf[B + cl]; c_k0 = lpf[3 * B + cl]; }
                }
            } else {
                const float dj = A.xpx[j];
                const double u = draw_uniform(key, marker, 0u);
                const double z = draw_normal(key, marker, 0u);
                c_d = dj;
                if constexpr (kR) {
                    double pj[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) pj[k] = P->pi_mat ? P->pi_mat[4 * j + k] : P->pi4[k];
                    bm.prepare(dj, P->var_effect[0], pj, P->gamma, ie, u, z);
                    r_il1 = bm.invLhs[1]; r_il2 = bm.invLhs[2]; r_il3 = bm.invLhs[3];
                    r_zs1 = bm.zs[1]; r_zs2 = bm.zs[2]; r_zs3 = bm.zs[3];
                    r_T0 = bm.T[0]; r_T1 = bm.T[1]; r_T2 = bm.T[2];
                    c_thrx = (a_cur != 0.f) ? 0.f : bayesr_candidate_threshold(bm, ie);
                } else {
                    float var_j = P->var_effect[0];
                    if constexpr (METHOD == kBayesB) var_j = P->var_vec[j];
                    double pi_j = P->pi;
                    if (P->pi_vec) pi_j = P->pi_vec[j];
                    AbcMarker am;
                    am.prepare(dj, var_j, pi_j, ie, u, z);
                    am.thresholds(a_cur, ie, c_lo, c_hi);
                    c_il = am.invLhs; c_bex = am.beta_excl; c_zs = am.zs;
                    if constexpr (DENSE) am.rule_d(a_cur, ie, c_k1, c_k0);       // Rule D with this repetition's alpha_old and draw
                }
                // a marker that is not touched in this repetition gets the repetition's "out of the model" draw
                if (valid) { if constexpr (kR) dpark[c] = 1.f; else { bpark[c] = c_bex; dpark[c] = 0.f; } }
            }
            const bool nz = a_cur != 0.f;
            // speculative rounds: every pending lane tests ITS marker against the current rhs (float compares only);
            // the first lane whose effect changes commits, the rest are re-tested after its Gram row corrected the rhs.
            // Lanes before the winner stay out of the model with the values parked for them; only the winner writes.
            while (true) {
                ++nrounds;
                bool inc = false, ev = false;
                if constexpr (kR) ev = nz || (fabsf(rhs) >= c_thrx);
                else { inc = DENSE ? true : abc_included(rhs, c_lo, c_hi); ev = inc || nz; }
                const unsigned long long m = __ballot(ev && valid) & pending;
                if (m == 0ull) break;                     // no further change in this sub-block
                const int k = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
                float an = 0.f;
                if constexpr (kR) {
                    bool sure = true;
                    int cls = bayesr_eval_thr(rhs, a_cur, ie, c_d, r_il1, r_il2, r_il3, r_zs1, r_zs2, r_zs3, r_T0, r_T1, r_T2, an, sure);
                    if (__builtin_amdgcn_readlane(sure ? 0 : 1, k)) {       // (practically never: s within 1e-6 of a class threshold)
                        if (rep == 0) bm.load(A.prep_d, A.prep_f, p, j, c_d, ie);          // full constants from global
                        cls = bm.evaluate(rhs, a_cur, ie, an);
                        ++nslow;
                    }
                    if (cls == 0) an = 0.f;
                    if (lane == k) { acur[c] = an; dpark[c] = (float)(cls + 1); }             // stored as class 1..4
                } else {
                    if constexpr (DENSE) an = fmaf(c_k1, rhs, c_k0);         // Rule D (inc is always true)
                    else an = abc_alpha_new(rhs, a_cur, c_d, ie, c_il, c_zs, inc);
                    if (lane == k) { acur[c] = an; bpark[c] = inc ? an : c_bex; dpark[c] = inc ? 1.f : 0.f; }
                }
                const float Dl = a_cur - an;
                pending = (k == 63) ? 0ull : (pending & ~((2ull << k) - 1ull));
                const float D = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Dl), k));
                if (D != 0.f) {
                    // rhs += D * G[ce][:] (BayesABC.jl:169,172).  The active sub-block is corrected in its register
                    // copy -- the only value the next round waits for.
                    const int ce = 64 * s + k;
                    int sl = __builtin_amdgcn_readlane(my_slot, k);
                    bool overflow = false;
                    if (sl < 0) {                                            // rare: not staged at entry
                        if (nstaged < SM.max_cand) sl = nstaged++; else { sl = SM.max_cand; overflow = true; }
                        fetch_row(ce, sl);
                        if (lane == 0) atomicAdd(&A.counters[1], 1ull);      // diagnostic
                    }
                    const float* grow = rows + sl * B;
                    rhs = fmaf(D, grow[c], rhs);
                    if (lazy && !overflow) {
                        if (lane == 0) evlog[nlog] = make_int2(sl, __float_as_int(D));
                        ++nlog;
                    } else {
                        for (int s2 = lazy ? s + 1 : 0; s2 < nsub; ++s2) {
                            const int c2 = 64 * s2 + lane;
                            if (s2 != s) rhs_lds[c2] = fmaf(D, grow[c2], rhs_lds[c2]);
                        }
                    }
                }
                if (pending == 0ull) break;
            }
            rhs_lds[c] = rhs;
        }
    }
    // the net changes of this block as a compact list in LDS (nothing changed before the first candidate's sub-block);
    // single-pass sweeps: the published change log IS that list
    tk4 = clock64();
    if (!(lazy && !dense_done)) {
        int base = 0;
#pragma unroll 1
        for (int s = s_first; s < nsub; ++s) {
            const int c = 64 * s + lane;
            const bool changed = (c < b) && (astart[c] != acur[c]);
            const unsigned long long cm = __ballot(changed);
            if (changed) reinterpret_cast<int*>(smem + SM.log_off)[base + __popcll(cm & ((1ull << lane) - 1ull))] = c;
            base += __popcll(cm);
        }
        if (lane == 0) { wcnt_s[15] = base; wcnt_s[11] = -1; }
    }
    tk5 = clock64();
    }   // wave 0
    __syncthreads();
    const bool from_log = wcnt_s[11] >= 0;                  // single pass: {column, d} pairs published by the serial wave
    const int nfin = from_log ? wcnt_s[11] : wcnt_s[15];
    const long long tk6 = clock64();
    if (A.b_next > 0 && !stream_corr && cross_lds && wcnt_s[14] != 0) {
        // dense walk: every marker of the block is an entry (alpha_old - alpha_new left in rhs_lds; 0 = exact no-op), the
        // cross-Gram rows are in LDS: one thread per column of the next block, a chain of b fused multiply-adds in marker
        // order fed by broadcast reads of four changes and conflict-free reads of the rows
        const float* crossL = reinterpret_cast<const float*>(smem + SM.cross_off);
        if (tid < B) {
            float corr = 0.f;
            if (tid < A.b_next) {
                int e = 0;
#pragma unroll 1
                for (; e + 8 <= b; e += 8) {
                    const float4 d0 = *reinterpret_cast<const float4*>(rhs_lds + e), d1 = *reinterpret_cast<const float4*>(rhs_lds + e + 4);
                    float g[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) g[u] = crossL[(e + u) * B + tid];
                    corr = fmaf(d0.x, g[0], corr); corr = fmaf(d0.y, g[1], corr); corr = fmaf(d0.z, g[2], corr); corr = fmaf(d0.w, g[3], corr);
                    corr = fmaf(d1.x, g[4], corr); corr = fmaf(d1.y, g[5], corr); corr = fmaf(d1.z, g[6], corr); corr = fmaf(d1.w, g[7], corr);
                }
                for (; e < b; ++e) corr = fmaf(rhs_lds[e], crossL[e * B + tid], corr);
            }
            A.corr_out[tid] = corr;
        }
    } else if (A.b_next > 0 && !stream_corr) {
        if (from_log) {                                     // (small blocks with a sparse prior: corr_phase wants plain columns)
            const int2* plog = reinterpret_cast<const int2*>(smem + SM.log_off);
            int cols[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) cols[q] = (tid + q * kStepThreads < nfin) ? plog[tid + q * kStepThreads].x : 0;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q) if (tid + q * kStepThreads < nfin) reinterpret_cast<int*>(smem + SM.log_off)[tid + q * kStepThreads] = cols[q];
            __syncthreads();
        }
        corr_phase<1>(smem, SM, A, nfin, cross_lds);
    }
    const long long tk7 = clock64();
    // ---- global stores LAST (nothing in this launch waits for them; a barrier after a global store waits for the store):
    // the change list for the next update role, alpha of the changed markers, beta / delta of the whole block
    if (stream_corr && is_corr_helper(wave)) {              // the lookahead correction accumulated by this helper lane
        const int col = (corr_helper_index(wave) * 64 + lane) * 4, bn = A.b_next;
        const float cv[4] = {corr_mine.x, corr_mine.y, corr_mine.z, corr_mine.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) if (col + i < B) A.corr_out[col + i] = (col + i < bn) ? cv[i] : 0.f;
    }
    {
        const int* fin = reinterpret_cast<const int*>(smem + SM.log_off);
        const bool pairs = from_log && (stream_corr || A.b_next <= 0);          // (else the list was turned into plain columns)
        for (int e = tid; e < nfin; e += kStepThreads) {
            const int ce = pairs ? fin[2 * e] : fin[e];
            const float d = astart[ce] - acur[ce];
            A.ev_out->idx[e] = (int32_t)(j0 + ce);
            A.ev_out->delta[0][e] = d;
            if (e < 7) { A.ev_out->hidx[e] = (int32_t)(j0 + ce); A.ev_out->hdelta[e] = d; }
            A.alpha[j0 + ce] = acur[ce];
        }
        // single-pass BayesA/B/C: a marker is in the model iff its effect is nonzero, beta = the effect, else its
        // "excluded" draw (parked at entry) -- the serial wave only wrote alpha
        const bool derive_bd = !kR && from_log;
        for (int c = tid; c < b; c += kStepThreads) {
            if constexpr (kR) reinterpret_cast<int32_t*>(A.delta)[j0 + c] = (int32_t)dpark0[c];
            else if (derive_bd) {
                const float a = acur[c];
                A.beta[j0 + c] = (a != 0.f) ? a : bpark0[c];
                reinterpret_cast<float*>(A.delta)[j0 + c] = (a != 0.f) ? 1.f : 0.f;
            }
            else { A.beta[j0 + c] = bpark0[c]; reinterpret_cast<float*>(A.delta)[j0 + c] = dpark0[c]; }
        }
    }
    if (tid == 0) {
        A.ev_out->count = nfin;
        atomicAdd(&A.counters[0], (unsigned long long)nfin);
        atomicAdd(&A.counters[2], (unsigned long long)(tk1 - tk0));      // phase cycle counts (diagnostics)
        atomicAdd(&A.counters[3], (unsigned long long)(tk2 - tk1));
        atomicAdd(&A.counters[4], (unsigned long long)(tk3 - tk2));
        atomicAdd(&A.counters[5], (unsigned long long)(tk4 - tk3));
        atomicAdd(&A.counters[6], (unsigned long long)(tk5 - tk4));
        atomicAdd(&A.counters[7], (unsigned long long)nrounds);
        if (nslow) atomicAdd(&A.counters[8], (unsigned long long)nslow);      // BayesR: rounds that needed the double-precision evaluation
        atomicAdd(&A.counters[9], (unsigned long long)(tk7 - tk6));           // the lookahead-correction phase
        atomicAdd(&A.counters[10], (unsigned long long)(tss[0] - tk2));       // staging: slot assignment | load issue | LDS stores
        atomicAdd(&A.counters[11], (unsigned long long)(tss[1] - tss[0]));
        atomicAdd(&A.counters[12], (unsigned long long)(tss[2] - tss[1]));
    }
}

// ---------------------------------------------------------------------------------------------
// Per-marker evaluation of the multi-trait samplers.  Inputs: w[k] = rhs_k + d*alpha_k (fp32), the
// marker's current (alpha, beta, delta), its draws.  Outputs: new (an, bn, dn) and the axpy
// coefficients Dl[k] (alpha_old - alpha_new; 0 = no change).  Operation for operation the oracle's
// mt1_update / mt2_update / mega_update.
// ---------------------------------------------------------------------------------------------
template <int NT>
struct MtConsts {
    float Rinv[NT][NT], Ginv[NT][NT];
    float invG[NT], lG[NT], sG[NT];                   // sampler I: 1/Ginv_kk, log Ginv_kk, sqrt(1/Ginv_kk)
    // mega (constraint = true): per-trait single-trait BayesC constants
    float ie[NT], var[NT], iv[NT], lv[NT], sv[NT];   // sv = sqrt(var)
    double lp0[NT], lp1[NT];
};

// Per-marker quantities that depend only on x'x (not on the running rhs): computed once per marker, SIMD across the
// markers of a sub-block, instead of inside every evaluation (each holds a double-precision log).
template <int NT>
struct MtPre {
    float C11[NT], invLhs1[NT], lC11[NT], s1[NT];     // sampler I: C11, 1/C11, log C11, sqrt(1/C11)
                                                      // mega:      lhs, 1/lhs, log(lhs) + log(var), sqrt(1/lhs)
};
// lc = the logs k_prepare took for this marker (prep_f rows 0..NT-1)
template <int METHOD, int NT>
__device__ __forceinline__ MtPre<NT> mt_precompute(const MtConsts<NT>& K, float dj, const float (&lc)[NT])
{
    MtPre<NT> R;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        if constexpr (METHOD == kMegaBayesC) {
            R.C11[k] = dj * K.ie[k] + K.iv[k];                                     // BayesABC.jl:37
            R.invLhs1[k] = 1.0f / R.C11[k];                                        // :38
        } else {
            R.C11[k] = K.Ginv[k][k] + K.Rinv[k][k] * dj;                           // MTBayesABC.jl:89
            R.invLhs1[k] = 1.0f / R.C11[k];                                        // :95
        }
        R.lC11[k] = lc[k];
        R.s1[k] = sqrtf(R.invLhs1[k]);
    }
    return R;
}

// Log prior probabilities of the two joint states sampler I compares for trait k (delta_k = 0 / 1, the other traits as
// they are now).  PriorMem: a table in memory (LDS; stride ls between states: 1 = the shared table, block size = this
// marker's column of the marker-specific priors).
struct PriorMem {
    const double* lpr; int ls;
    template <int NT>
    __device__ __forceinline__ void pair(int k, const float (&dn)[NT], double& l0, double& l1) const
    {
        unsigned s0 = 0u;
#pragma unroll
        for (int m = 0; m < NT; ++m) if (m != k && dn[m] != 0.f) s0 |= 1u << m;
        l0 = lpr[s0 * ls];
        l1 = lpr[(s0 | (1u << k)) * ls];
    }
};
// ---- Rule L (sampler I): the LINEAR FORM of a marker that is in the model for every trait and stays there.
// With every delta = 1 before and after, the marker's NT conditionals are one small triangular system:
//     beta_k = 1/C11_k * ( sum_m Rinv[m][k] w_m - sum_{m<k} C12[k][m] beta_m - sum_{m>k} C12[k][m] beta_old_m ) + z_k sqrt(1/C11_k)
// i.e.  beta = A w + c  with A (NT x NT) and c (NT) functions of the marker's constants, its old beta and its draws only --
// NOT of the running rhs.  The dense walk (every marker in the model: the reference's default all-ones prior) precomputes
// A, c for all 64 markers of a section in parallel and is left with NT^2 fused multiply-adds per marker on the serial
// chain instead of the ~26 dependent operations of the conditional-by-conditional order (430 -> ~110 cycles per 3-trait
// marker).  So that every path (dense walk, speculative rounds, the oracle) produces the SAME numbers, the rule is part
// of the sampler's definition: whenever the exact evaluation (below, the reference's operation order) says that a marker
// which entered with every delta = 1 leaves with every delta = 1, its new effects are the linear form's
//     beta_k = fmaf(A[k][NT-1], w[NT-1], ... fmaf(A[k][0], w[0], c[k]))      (A, c: double recurrence, rounded to float)
// -- the same conditional means and the same draws, another association: <= a few ulp from the reference's order.  Every
// other marker keeps the exact order's values.  The oracle applies the same rule (orc mt1_update; the literal order stays
// available there: orc_set_mt_linear_form(0), compared in tests/test_oracle_kat.py).
template <int NT>
__device__ __forceinline__ void mt1_linear_coeffs(const MtConsts<NT>& K, const MtPre<NT>& Q, float dj, const float (&b_old)[NT],
                                                  const double (&z)[NT], float (&A)[NT][NT], float (&cc)[NT])
{
    double Ad[NT][NT], cd[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const double il = (double)Q.invLhs1[k];
        double C12[NT];
#pragma unroll
        for (int m = 0; m < NT; ++m) C12[m] = (double)(K.Ginv[k][m] + (dj * 1.f) * K.Rinv[k][m]);       // MTBayesABC.jl:90 with delta_m = 1
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            double acc = (double)K.Rinv[m][k];
#pragma unroll
            for (int j = 0; j < k; ++j) acc = acc - C12[j] * Ad[j][m];
            Ad[k][m] = il * acc;
        }
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < k; ++j) acc = acc - C12[j] * cd[j];
#pragma unroll
        for (int j = k + 1; j < NT; ++j) acc = acc - C12[j] * (double)b_old[j];
        cd[k] = il * acc + z[k] * (double)Q.s1[k];
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        cc[k] = (float)cd[k];
#pragma unroll
        for (int m = 0; m < NT; ++m) A[k][m] = (float)Ad[k][m];
    }
}
template <int NT>
__device__ __forceinline__ void mt1_linear_beta(const float (&A)[NT][NT], const float (&cc)[NT], const float (&w)[NT], float (&b)[NT])
{
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        float v = cc[k];
#pragma unroll
        for (int m = 0; m < NT; ++m) v = fmaf(A[k][m], w[m], v);
        b[k] = v;
    }
}

// Gibbs sampler I (MTBayesABC.jl:85-120); LIN: apply Rule L to the result (off only where the result's VALUES are not kept).
template <int NT, bool LIN = true, class LP>
__device__ __forceinline__ void mt1_eval(const MtConsts<NT>& K, const MtPre<NT>& Q, const LP& lp, const float (&w)[NT], float dj,
                                         const double (&thr)[NT], const double (&z)[NT],
                                         float (&an)[NT], float (&bn)[NT], float (&dn)[NT], float (&Dl)[NT])
{
    float a_in[NT], b_in[NT];
    bool all1 = LIN;
#pragma unroll
    for (int k = 0; k < NT; ++k) { a_in[k] = an[k]; b_in[k] = bn[k]; all1 = all1 && (dn[k] == 1.f); }
#pragma unroll
    for (int k = 0; k < NT; ++k) {                                                  // :85
        const float Ginv11 = K.Ginv[k][k];
        const float C11 = Q.C11[k];                                                 // :89
        float rhs0 = 0.f, c12b = 0.f, wR = 0.f;
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            wR = wR + w[m] * K.Rinv[m][k];
            if (m == k) continue;
            const float C12m = K.Ginv[k][m] + (dj * dn[m]) * K.Rinv[k][m];          // :90
            rhs0 = rhs0 + K.Ginv[k][m] * bn[m];
            c12b = c12b + C12m * bn[m];
        }
        rhs0 = -rhs0;                                                               // :93
        const float invLhs0 = K.invG[k];
        const float gHat0 = rhs0 * invLhs0;
        const float invLhs1 = Q.invLhs1[k];
        const float rhs1 = wR - c12b;                                               // :96
        const float gHat1 = rhs1 * invLhs1;
        double lp0, lp1;
        lp.template pair<NT>(k, dn, lp0, lp1);
        const float in0 = K.lG[k] - (gHat0 * gHat0) * Ginv11;                       // :104
        const float in1 = Q.lC11[k] - (gHat1 * gHat1) * C11;                        // :105
        const double logDelta0 = -0.5 * (double)in0 + lp0;
        const double logDelta1 = -0.5 * (double)in1 + lp1;
        if ((logDelta0 - logDelta1) < thr[k]) {                                     // :107-111
            dn[k] = 1.f;
            bn[k] = (float)((double)gHat1 + z[k] * (double)Q.s1[k]);
            Dl[k] = an[k] - bn[k];
            an[k] = bn[k];
        } else {                                                                    // :112-119
            bn[k] = (float)((double)gHat0 + z[k] * (double)K.sG[k]);
            dn[k] = 0.f;
            Dl[k] = an[k];
            an[k] = 0.f;
        }
    }
    if constexpr (LIN) {
#pragma unroll
        for (int k = 0; k < NT; ++k) all1 = all1 && (dn[k] == 1.f);
        if (all1) {                                                                 // Rule L
            float A[NT][NT], cc[NT];
            mt1_linear_coeffs<NT>(K, Q, dj, b_in, z, A, cc);
            mt1_linear_beta<NT>(A, cc, w, bn);
#pragma unroll
            for (int k = 0; k < NT; ++k) { an[k] = bn[k]; Dl[k] = a_in[k] - bn[k]; }
        }
    }
}

// megaBayesABC! (BayesABC.jl:1-8): trait k is an independent single-trait BayesC update (BayesABC.jl:24-58)
template <int NT>
__device__ __forceinline__ void mega_eval(const MtConsts<NT>& K, const MtPre<NT>& Q, const float (&w)[NT], float dj,
                                          const double (&thr)[NT], const double (&z)[NT],
                                          float (&an)[NT], float (&bn)[NT], float (&dn)[NT], float (&Dl)[NT])
{
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const float rhs    = w[k] * K.ie[k];                                        // :36
        const float invLhs = Q.invLhs1[k];                                          // :37-38
        const float gHat   = rhs * invLhs;                                          // :39
        const float inner  = Q.lC11[k] - gHat * rhs;                                // (log lhs + log var) - gHat*rhs
        const double l1    = -0.5 * (double)inner + K.lp1[k];                       // :40
        if ((K.lp0[k] - l1) < thr[k]) {                                             // :41,:44
            dn[k] = 1.f;
            bn[k] = (float)((double)gHat + z[k] * (double)Q.s1[k]);                 // :46
            Dl[k] = an[k] - bn[k];
            an[k] = bn[k];
        } else {
            dn[k] = 0.f;
            bn[k] = (float)(z[k] * (double)K.sv[k]);                                // :54
            Dl[k] = an[k];
            an[k] = 0.f;
        }
    }
    (void)dj;
}

// lower Cholesky factor of an SPD NT x NT matrix (fixed operation order, shared with the oracle's chol_lower)
template <int NT>
__device__ __forceinline__ void chol_lower(const double (&A)[NT][NT], double (&L)[NT][NT])
{
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        double s = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) s = s - L[j][k] * L[j][k];
        L[j][j] = sqrt(s);
#pragma unroll
        for (int i = j + 1; i < NT; ++i) {
            double v = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v = v - L[i][k] * L[j][k];
            L[i][j] = v / L[j][j];
        }
    }
}

// ---- multi-trait BayesA/B: one InverseWishart(df, scale + b_j b_j') draw per marker (variance_components.jl:181-186:
// sample_variance(data, 1, df, scale) per marker -- the host's 100 000 draws per iteration were ~35 ms of numpy and made
// the multi-trait BayesB iteration host-bound).  One thread per marker, Bartlett's decomposition on the counter RNG:
//   S = scale + b b' = C C' (chol_lower);  A lower-triangular, A_ii = sqrt(chi2(df - i)), A_ik ~ N(0,1) (k < i);
//   K' = A^-1 C' (forward substitution);  G = K K', symmetrised, rounded to float.
// W = C'^-1 A A' C^-1 ~ Wishart(df, S^-1) and G = W^-1.  Counter of a draw: (global marker, iteration, 0x80000000 | attempt,
// slot): slot 32 + 2i (+1) the chi-square of row i (Marsaglia-Tsang gamma: one normal + one uniform per attempt), slot
// 64 + 4i + k the normal A_ik -- disjoint from the sweep's own draws (repetition index < 2^31, slots 0 / 1 + 16 trait).
// Operation for operation the oracle's orc_sample_marker_covariances.
__device__ __forceinline__ double iw_chi2(uint32_t marker, uint32_t iter, uint32_t slot, uint32_t k0, uint32_t k1, double nu)
{
    double a = 0.5 * nu, boost = 1.0;
    if (a < 1.0) {                                   // gamma(a) = gamma(a + 1) * u^(1/a)
        const u32x4 w = philox4x32_10(marker, iter, 0x80000000u | 0xFFFFu, slot, k0, k1);
        boost = exp(log(u52(w.x, w.y)) / a);
        a = a + 1.0;
    }
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    double g = d;
    for (uint32_t attempt = 0; attempt < 64u; ++attempt) {
        const u32x4 w = philox4x32_10(marker, iter, 0x80000000u | attempt, slot, k0, k1);
        const u32x4 w2 = philox4x32_10(marker, iter, 0x80000000u | attempt, slot + 1u, k0, k1);
        const double x = sqrt(-2.0 * log(u52(w.x, w.y))) * cos(6.283185307179586476925286766559 * u52(w.z, w.w));
        const double u = u52(w2.x, w2.y);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        g = d * v;
        if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) break;
    }
    return 2.0 * g * boost;
}

struct IwParams { double df; double scale[kMaxT * kMaxT]; uint32_t seed_lo, seed_hi, iter, marker0; };

template <int NT>
__global__ __launch_bounds__(256) void k_sample_marker_covariances(IwParams Q, int64_t p, const float* __restrict__ beta, float* __restrict__ var_mat)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= p) return;
    const uint32_t marker = Q.marker0 + (uint32_t)j;
    double b[NT], S[NT][NT], C[NT][NT], A[NT][NT], Kt[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) b[a] = (double)beta[(int64_t)a * p + j];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c) { S[a][c] = Q.scale[a * NT + c] + b[a] * b[c]; C[a][c] = 0.0; A[a][c] = 0.0; }
    chol_lower<NT>(S, C);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        A[i][i] = sqrt(iw_chi2(marker, Q.iter, 32u + 2u * (uint32_t)i, Q.seed_lo, Q.seed_hi, Q.df - (double)i));
#pragma unroll
        for (int k = 0; k < i; ++k) {
            const u32x4 w = philox4x32_10(marker, Q.iter, 0x80000000u, 64u + 4u * (uint32_t)i + (uint32_t)k, Q.seed_lo, Q.seed_hi);
            A[i][k] = sqrt(-2.0 * log(u52(w.x, w.y))) * cos(6.283185307179586476925286766559 * u52(w.z, w.w));
        }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            double acc = C[c][i];                                       // C'[i][c]
#pragma unroll
            for (int k = 0; k < i; ++k) acc = acc - A[i][k] * Kt[k][c];
            Kt[i][c] = acc / A[i][i];
        }
    double G[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < NT; ++i) s = s + Kt[i][a] * Kt[i][c];
            G[a][c] = s;
        }
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c) var_mat[(j * NT + a) * NT + c] = (float)(0.5 * (G[a][c] + G[c][a]));
}

// Gibbs sampler II, one candidate state (MTBayesABC.jl:178-185).  st: bit k = trait k in the model.
// q = -0.5*(log det lhs - rhs'gHat); cand = gHat + chol(lhs^-1)*z only when want_cand.
// The evaluation of one state is split in three: the part that depends on the marker's x'x and the sweep's variances
// only (lhs, its inverse and log determinant -- both Cholesky factorisations' worth of divisions and square roots),
// the part that depends on the running rhs (a handful of multiply-adds), and the candidate effects of the chosen state.
// mt2_state = pre + post (+ cand): one operation order, shared with the oracle's mt2_state.
template <int NT>
__device__ __forceinline__ void mt2_state_pre(const MtConsts<NT>& K, unsigned st, float dj, double (&inv)[NT][NT], double& logdet)
{
    double lhs[NT][NT], L[NT][NT], M[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        const double Da = ((st >> a) & 1u) ? 1.0 : 0.0;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const double Dc = ((st >> c) & 1u) ? 1.0 : 0.0;
            const double rl = (Da * (double)K.Rinv[a][c]) * Dc;                     // D*Rinv*D  :159
            lhs[a][c] = rl * (double)dj + (double)K.Ginv[a][c];                     // :179
        }
    }
    chol_lower<NT>(lhs, L);
#pragma unroll
    for (int j = 0; j < NT; ++j) {                                                  // M = L^-1
        M[j][j] = 1.0 / L[j][j];
#pragma unroll
        for (int i = j + 1; i < NT; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = j; k < i; ++k) s = s + L[i][k] * M[k][j];
            M[i][j] = -s / L[i][i];
        }
    }
#pragma unroll
    for (int a = 0; a < NT; ++a)                                                    // inv(lhs) = M'M  :181
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            double s = 0.0;
#pragma unroll
            for (int k = (a > c ? a : c); k < NT; ++k) s = s + M[k][a] * M[k][c];
            inv[a][c] = s;                                                          // (bitwise symmetric: products commute)
        }
    double det = 1.0;
#pragma unroll
    for (int j = 0; j < NT; ++j) det = det * (L[j][j] * L[j][j]);
    logdet = log(det);
}
template <int NT>
__device__ __forceinline__ void mt2_state_post(const MtConsts<NT>& K, unsigned st, const float (&w)[NT],
                                               const double (&inv)[NT][NT], double logdet, double& q, double (&gHat)[NT])
{
    double rhs[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        const double Da = ((st >> a) & 1u) ? 1.0 : 0.0;
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < NT; ++m) s = s + ((double)K.Rinv[m][a] * Da) * (double)w[m];   // (Rinv*D)'w :180
        rhs[a] = s;
    }
    double quad = 0.0;
#pragma unroll
    for (int a = 0; a < NT; ++a) {                                                  // gHat = invLhs*rhs :183
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < NT; ++c) s = s + inv[a][c] * rhs[c];
        gHat[a] = s;
        quad = quad + rhs[a] * s;
    }
    q = -0.5 * (logdet - quad);                                                     // :184
}
template <int NT>
__device__ __forceinline__ void mt2_state_cand(const double (&inv)[NT][NT], const double (&gHat)[NT], const double (&z)[NT],
                                               double (&cand)[NT])
{
    double C[NT][NT];
    chol_lower<NT>(inv, C);                                                         // cholesky(Hermitian(invLhs)).L :182
#pragma unroll
    for (int a = 0; a < NT; ++a) {                                                  // gHat + L*z  :185
        double s = gHat[a];
#pragma unroll
        for (int c = 0; c <= a; ++c) s = s + C[a][c] * z[c];
        cand[a] = s;
    }
}
template <int NT>
__device__ __forceinline__ void mt2_state(const MtConsts<NT>& K, unsigned st, const float (&w)[NT], float dj,
                                          const double (&z)[NT], bool want_cand, double& q, double (&cand)[NT])
{
    double inv[NT][NT], gHat[NT], logdet;
    mt2_state_pre<NT>(K, st, dj, inv, logdet);
    mt2_state_post<NT>(K, st, w, inv, logdet, q, gHat);
    if (want_cand) mt2_state_cand<NT>(inv, gHat, z, cand);
}

// Per-marker table of the state-dependent, rhs-independent quantities (sampler II, NT <= 3): for each of the 2^NT
// states the NT(NT+1)/2 unique entries of inv(lhs) (row-major upper triangle) and log det lhs.  Filled once per sweep
// for all markers in parallel (k_prepare_mt2); layout [state][value][p].
template <int NT>
struct Mt2Tab {
    static constexpr int NS = 1 << NT, NV = NT * (NT + 1) / 2 + 1, kRows = NS * NV;
};
template <int NT>
__device__ __forceinline__ void mt2_unpack(const double (&row)[Mt2Tab<NT>::NV], double (&inv)[NT][NT], double& logdet)
{
    int v = 0;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int c = a; c < NT; ++c) { inv[a][c] = row[v]; inv[c][a] = row[v]; ++v; }
    logdet = row[v];
}
template <int NT>
__global__ __launch_bounds__(256) void k_prepare_mt2(const DevParams* __restrict__ P, int64_t p, const float* __restrict__ xpx,
                                                     double* __restrict__ tab)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= p) return;
    MtConsts<NT> K;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c) { K.Rinv[a][c] = P->Rinv[a * NT + c]; K.Ginv[a][c] = P->Ginv[a * NT + c]; }
    const float dj = xpx[j];
#pragma unroll 1
    for (int st = 0; st < Mt2Tab<NT>::NS; ++st) {
        double inv[NT][NT], logdet;
        mt2_state_pre<NT>(K, (unsigned)st, dj, inv, logdet);
        int v = 0;
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int c = a; c < NT; ++c) { tab[((int64_t)st * Mt2Tab<NT>::NV + v) * p + j] = inv[a][c]; ++v; }
        tab[((int64_t)st * Mt2Tab<NT>::NV + v) * p + j] = logdet;
    }
}
template <int NT>
__device__ __forceinline__ void mt2_load_tab(const double* __restrict__ tab, int64_t p, int64_t j,
                                             double (&T)[Mt2Tab<NT>::NS][Mt2Tab<NT>::NV])
{
#pragma unroll
    for (int st = 0; st < Mt2Tab<NT>::NS; ++st)
#pragma unroll
        for (int v = 0; v < Mt2Tab<NT>::NV; ++v) T[st][v] = tab[((int64_t)st * Mt2Tab<NT>::NV + v) * p + j];
}

// Gibbs sampler II, one marker, from its state table (same results as mt2_eval).
template <int NT>
__device__ __forceinline__ void mt2_eval_tab(const MtConsts<NT>& K, const double* lpr, int ls, const float (&w)[NT],
                                             const double (&T)[Mt2Tab<NT>::NS][Mt2Tab<NT>::NV],
                                             double u, const double (&z)[NT],
                                             float (&an)[NT], float (&bn)[NT], float (&dn)[NT], float (&Dl)[NT])
{
    constexpr int NS = Mt2Tab<NT>::NS, NV = Mt2Tab<NT>::NV;
    double ld[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        double inv[NT][NT], lg, q, gh[NT];
        mt2_unpack<NT>(T[s], inv, lg);
        mt2_state_post<NT>(K, (unsigned)s, w, inv, lg, q, gh);
        ld[s] = q + lpr[s * ls];
    }
    int which = NS - 1;
    {                                                                               // :188-198
        double mx = -INFINITY;
#pragma unroll
        for (int s = 0; s < NS; ++s) if (ld[s] > mx) mx = ld[s];
        double den = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) { ld[s] = exp(ld[s] - mx); den += ld[s]; }
        double cp = 0.0;
        bool found = false;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            cp += ld[s] / den;
            if (!found && u < cp) { which = s; found = true; }
        }
    }
    double row[NV];                                                                 // the chosen state's row: select chain
#pragma unroll
    for (int v = 0; v < NV; ++v) row[v] = T[0][v];
#pragma unroll
    for (int s = 1; s < NS; ++s)
#pragma unroll
        for (int v = 0; v < NV; ++v) row[v] = (which == s) ? T[s][v] : row[v];
    double inv[NT][NT], lg, q, gh[NT], cand[NT];
    mt2_unpack<NT>(row, inv, lg);
    mt2_state_post<NT>(K, (unsigned)which, w, inv, lg, q, gh);
    mt2_state_cand<NT>(inv, gh, z, cand);
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const double dk = ((which >> k) & 1) ? 1.0 : 0.0;
        const double a_new = dk * cand[k];                                          // diagm(delta)*beta :201
        Dl[k] = (float)((double)an[k] - a_new);                                     // oldα-newα -> axpy :204
        bn[k] = (float)cand[k];
        dn[k] = (float)dk;
        an[k] = (float)a_new;
    }
}

// Gibbs sampler II, one marker (MTBayesABC.jl:160-208).  u = the marker's uniform (slot 0).
template <int NT>
__device__ __forceinline__ void mt2_eval(const MtConsts<NT>& K, const double* lpr, int ls, const float (&w)[NT], float dj,
                                         double u, const double (&z)[NT],
                                         float (&an)[NT], float (&bn)[NT], float (&dn)[NT], float (&Dl)[NT])
{
    constexpr int NS = 1 << NT;
    double ld[NS], cand[NT];
#pragma unroll
    for (int s = 0; s < NS; ++s) ld[s] = 0.0;
    int which = NS - 1;
    // passes 0..NS-1 evaluate the states; pass NS re-evaluates the chosen one for its candidate effects
#pragma unroll 1
    for (int pass = 0; pass <= NS; ++pass) {
        const unsigned st = pass < NS ? (unsigned)pass : (unsigned)which;
        double q;
        mt2_state<NT>(K, st, w, dj, z, pass == NS, q, cand);
        if (pass < NS) {
            const double v = q + lpr[pass * ls];
#pragma unroll
            for (int s = 0; s < NS; ++s) ld[s] = (s == pass) ? v : ld[s];
        }
        if (pass == NS - 1) {                                                       // :188-198
            double mx = -INFINITY;
#pragma unroll
            for (int s = 0; s < NS; ++s) if (ld[s] > mx) mx = ld[s];
            double den = 0.0;
#pragma unroll
            for (int s = 0; s < NS; ++s) { ld[s] = exp(ld[s] - mx); den += ld[s]; }
            double cp = 0.0;
            bool found = false;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                cp += ld[s] / den;
                if (!found && u < cp) { which = s; found = true; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const double dk = ((which >> k) & 1) ? 1.0 : 0.0;
        const double a_new = dk * cand[k];                                          // diagm(delta)*beta :201
        Dl[k] = (float)((double)an[k] - a_new);                                     // oldα-newα -> axpy :204
        bn[k] = (float)cand[k];
        dn[k] = (float)dk;
        an[k] = (float)a_new;
    }
}

// ---------------------------------------------------------------------------------------------
// SAMPLER role, multi-trait: Gibbs sampler I (MTBayesABC.jl:57-127, block form :243-333), sampler II
// (:129-210) and megaBayesABC! (BayesABC.jl:1-8) share the schedule; only the per-marker evaluation differs.
// ---------------------------------------------------------------------------------------------
template <int METHOD, int NT>
__device__ __forceinline__ void sampler_role_mt(char* smem, const SamplerArgs& A)
{
    const bool pm = A.lpr_mat != nullptr;           // marker-specific joint priors (host: only with parked draws)
    constexpr bool kPG = (METHOD == kMTBayesB1);    // a t x t effect covariance per marker (host: only with parked draws)
    const StepSmem SM(A.bsz, NT, mt_park_nd(A.bsz, NT) + (pm ? (1 << NT) : 0), mt_park_nf(A.bsz, NT) + (kPG ? NT * NT : 0));
    const int B = SM.B;
    const bool parked = mt_park_nd(B, NT) != 0;
    constexpr bool kTab = (METHOD == kMTBayesC2) && (NT <= 3);       // sampler II from per-marker state tables
    constexpr int kTS = kTab ? (1 << NT) : 1, kTV = kTab ? NT * (NT + 1) / 2 + 1 : 1;
    const DevParams* P = A.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = A.b;
    const int64_t j0 = A.j0, p = A.p;
    float* rhs_lds = reinterpret_cast<float*>(smem + SM.rhs_off);
    float* acur = reinterpret_cast<float*>(smem + SM.acur_off);
    float* astart = reinterpret_cast<float*>(smem + SM.astart_off);
    float* bcur = reinterpret_cast<float*>(smem + SM.bcur_off);
    float* dcur = reinterpret_cast<float*>(smem + SM.dcur_off);
    double* lpd = reinterpret_cast<double*>(smem + SM.prepd_off);     // [2 NT][B] thresholds, normals (if parked)
    float* lpf = reinterpret_cast<float*>(smem + SM.prepf_off);       // [B] x'x (if parked)
    float* lpg = lpf + (1 + NT) * B;                                  // [NT*NT][B] the marker's own G^-1 (kPG)
    float* delta = reinterpret_cast<float*>(A.delta);
    const long long tk0 = clock64();

    MtConsts<NT> K;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
#pragma unroll
        for (int c = 0; c < NT; ++c) { K.Rinv[a][c] = P->Rinv[a * NT + c]; K.Ginv[a][c] = P->Ginv[a * NT + c]; }
        K.invG[a] = 1.0f / K.Ginv[a][a];                            // MTBayesABC.jl:92
        K.lG[a] = logf_via_double(K.Ginv[a][a]);
        K.sG[a] = sqrtf(K.invG[a]);
        if constexpr (METHOD == kMegaBayesC) {
            K.ie[a]  = 1.0f / P->vare[a * NT + a];                  // invVarRes          BayesABC.jl:69
            K.var[a] = P->var_effect[a * NT + a];
            K.iv[a]  = 1.0f / K.var[a];                             // invVarEffects[j]   :70
            K.lv[a]  = logf_via_double(K.var[a]);                   // logVarEffects[j]   :71
            K.sv[a]  = sqrtf(K.var[a]);
            K.lp0[a] = log(P->pi4[a]);                              // logPi              :67
            K.lp1[a] = log(1.0 - P->pi4[a]);                        // logPiComp          :68
        }
    }
    // multi-trait BayesA/B: the constants that depend on G are the marker's own (its inverse was formed by k_prepare)
    auto with_ginv = [&](const float (&g)[NT * NT]) {
        MtConsts<NT> Kj = K;
#pragma unroll
        for (int a = 0; a < NT; ++a) {
#pragma unroll
            for (int c2 = 0; c2 < NT; ++c2) Kj.Ginv[a][c2] = g[a * NT + c2];
            Kj.invG[a] = 1.0f / Kj.Ginv[a][a];                      // MTBayesABC.jl:92
            Kj.lG[a] = logf_via_double(Kj.Ginv[a][a]);
            Kj.sG[a] = sqrtf(Kj.invG[a]);
        }
        return Kj;
    };
    auto consts_of = [&](int c) {                                     // marker c of the block (after the front's barrier)
        if constexpr (kPG) {
            float g[NT * NT];
#pragma unroll
            for (int i = 0; i < NT * NT; ++i) g[i] = lpg[i * B + c];
            return with_ginv(g);
        } else { (void)c; return K; }
    };
    // the 2^NT log prior state probabilities are indexed by the running state inside every evaluation: a global load
    // there would put a memory latency (microseconds under full-rate streaming) on each trait of each round -- LDS copy
    double* lpr = reinterpret_cast<double*>(smem + SM.lpr_off);
    const double lpr_mine = P->log_prior[tid < (1 << NT) ? tid : 0];

    // ---- front (all threads, ONE memory latency): every thread issues the loads of its marker's state, draws, x'x,
    // lookahead correction and row-group partials back to back, forms  rhs = fl32(sum of partials) + corr,  parks
    // everything the serial wave needs in LDS, and decides candidacy: a marker already in the model for some trait
    // (its effects always change) or one whose evaluation against the entry rhs changes an effect.  Candidates get
    // their Gram row staged in LDS; a change of a non-candidate reads its row from HBM inside the serial phase.
    // small blocks (the host's choice for dense priors): the whole Gram block with the very first loads, as in the
    // single-trait sampler (slot of marker c = c)
    const bool prestage = (B <= 128) && (B <= SM.max_cand);
    const bool gram_dma = prestage && b == B;              // full block: direct global -> LDS loads (see sampler_role_st)
    const bool cross_dma = gram_dma && SM.has_cross && A.b_next == B;
    if (gram_dma) dma_copy_to_lds(A.gram, reinterpret_cast<float*>(smem + SM.rows_off), B * B);
    float4 gpre[8];
    if (prestage && !gram_dma) {
        const int per_row = B >> 2, total = b * per_row;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + u * kStepThreads;
            const int ec = e < total ? e : 0;
            const int row = ec / per_row, c4 = (ec - row * per_row) * 4;
            const float* src = A.gram + (int64_t)row * b;
            if (b == B) gpre[u] = *reinterpret_cast<const float4*>(src + c4);
            else {
                gpre[u].x = src[c4 < b ? c4 : 0]; gpre[u].y = src[c4 + 1 < b ? c4 + 1 : 0];
                gpre[u].z = src[c4 + 2 < b ? c4 + 2 : 0]; gpre[u].w = src[c4 + 3 < b ? c4 + 3 : 0];
            }
        }
    }
    bool cand[2] = {false, false};
    float djq_[2], a0[2][NT], b0[2][NT], d0[2][NT], w0[2][NT], lc0[2][NT], gq_[2][kPG ? NT * NT : 1];
    double thr0[2][NT], z0[2][NT];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = tid + q * kStepThreads;
        if (c >= B) continue;
        const int cc = c < b ? c : 0;
        const int64_t j = j0 + cc;
        const float dj = A.xpx[j];
        float co[NT];
        djq_[q] = dj;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            a0[q][t] = A.alpha[(int64_t)t * p + j]; b0[q][t] = A.beta[(int64_t)t * p + j]; d0[q][t] = delta[(int64_t)t * p + j];
            co[t] = A.corr_in[t * B + c];
            thr0[q][t] = A.prep_d[(int64_t)t * p + j]; z0[q][t] = A.prep_d[(int64_t)(NT + t) * p + j];
            lc0[q][t] = A.prep_f[(int64_t)t * p + j];
        }
        if constexpr (kPG) {
#pragma unroll
            for (int i = 0; i < NT * NT; ++i) gq_[q][i] = A.ginv_mat[j * (NT * NT) + i];
        }
        double lpm[1 << NT];
        if (pm) {
#pragma unroll
            for (int st = 0; st < (1 << NT); ++st) lpm[st] = A.lpr_mat[(int64_t)(1 << NT) * j + st];
        }
        double psum[NT];
        sum_partials_traits<NT>(A.partials + cc, (int64_t)A.nrg * A.bstride, A.nrg, A.bstride, psum);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const double sum = psum[t];
            const float rhs0 = (float)sum + co[t];
            const float a_in = (c < b) ? a0[q][t] : 0.f;
            rhs_lds[t * B + c] = rhs0;
            acur[t * B + c] = a_in; astart[t * B + c] = a_in;
            bcur[t * B + c] = b0[q][t]; dcur[t * B + c] = d0[q][t];
            w0[q][t] = rhs0 + dj * a_in;                                                             // :82
            if (parked) { lpd[t * B + c] = thr0[q][t]; lpd[(NT + t) * B + c] = z0[q][t]; lpf[(1 + t) * B + c] = lc0[q][t]; }
            a0[q][t] = a_in;
        }
        if (parked) lpf[c] = dj;
        if constexpr (kPG) {
#pragma unroll
            for (int i = 0; i < NT * NT; ++i) lpg[i * B + c] = gq_[q][i];
        }
        if (pm) {
#pragma unroll
            for (int st = 0; st < (1 << NT); ++st) lpd[(2 * NT + st) * B + c] = lpm[st];
        }
    }
    if (tid < (1 << NT)) lpr[tid] = lpr_mine;
    if (tid == 0) reinterpret_cast<int*>(smem + SM.wcnt_off)[14] = 0;      // set by the dense walk
    if (prestage) {
        float* rows_p = reinterpret_cast<float*>(smem + SM.rows_off);
        short* slot_p = reinterpret_cast<short*>(smem + SM.slot_off);
        short* cand_p = reinterpret_cast<short*>(smem + SM.cand_off);
        if (!gram_dma) {
            const int per_row = B >> 2, total = b * per_row;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + u * kStepThreads;
                if (e < total) {
                    const int row = e / per_row, c4 = (e - row * per_row) * 4;
                    *reinterpret_cast<float4*>(rows_p + row * B + c4) = gpre[u];
                }
            }
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int c = tid; c < B; c += kStepThreads) { slot_p[c] = (short)(c < b ? c : -1); cand_p[c] = (short)c; }
    }
    __syncthreads();
    // marker c's table of log prior state probabilities: the shared one (stride 1) or its own column of the parked
    // marker-specific priors (stride B)
    const int ls = pm ? B : 1;
    auto lpr_of = [&](int c) -> const double* { return pm ? lpd + 2 * NT * B + c : lpr; };
    bool stay[2] = {false, false};
    float pb[2][NT], pd[2][NT];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = tid + q * kStepThreads;
        if (c >= b) continue;
        bool in_model = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) in_model = in_model || (a0[q][t] != 0.f);
        bool moves = false;
        if (!in_model) {
            const float dj = djq_[q];
            MtConsts<NT> Kc = K;
            if constexpr (kPG) Kc = with_ginv(gq_[q]);
            const MtPre<NT> Q0 = mt_precompute<METHOD, NT>(Kc, dj, lc0[q]);
            float an[NT], bn[NT], dn[NT], Dl[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) { an[t] = a0[q][t]; bn[t] = b0[q][t]; dn[t] = d0[q][t]; Dl[t] = 0.f; }
            if constexpr (is_sampler1(METHOD)) mt1_eval<NT, false>(Kc, Q0, PriorMem{lpr_of(c), ls}, w0[q], dj, thr0[q], z0[q], an, bn, dn, Dl);
            else if constexpr (kTab) {
                double T[kTS][kTV];
                mt2_load_tab<NT>(A.mt2_tab, p, j0 + c, T);
                mt2_eval_tab<NT>(K, lpr_of(c), ls, w0[q], T, thr0[q][0], z0[q], an, bn, dn, Dl);
            }
            else if constexpr (METHOD == kMTBayesC2) mt2_eval<NT>(K, lpr_of(c), ls, w0[q], dj, thr0[q][0], z0[q], an, bn, dn, Dl);
            else mega_eval<NT>(K, Q0, w0[q], dj, thr0[q], z0[q], an, bn, dn, Dl);
#pragma unroll
            for (int t = 0; t < NT; ++t) moves = moves || (Dl[t] != 0.f);
            if (!moves) {
                stay[q] = true;
#pragma unroll
                for (int t = 0; t < NT; ++t) { pb[q][t] = bn[t]; pd[q][t] = dn[t]; }
            }
        }
        cand[q] = in_model || moves;
    }
    // PREFIX SKIP (as in the single-trait sampler): until the first candidate of the block commits the running rhs is the
    // entry rhs, so the evaluation above is final for every marker before it.  Their freshly drawn beta / delta are parked
    // (only theirs: a later marker is re-evaluated from its OLD state) and the serial wave starts at the first sub-block
    // that holds a candidate.  Single pass only.
    int first_sub = 16, ncand_all = 0;
    {
        int* wc = reinterpret_cast<int*>(smem + SM.wcnt_off);
        const int f0 = __any(cand[0]) ? 1 : 0, f1 = __any(cand[1]) ? 2 : 0;
        const int npop = __popcll(__ballot(cand[0])) + __popcll(__ballot(cand[1]));
        if (lane == 0) wc[wave] = f0 | f1 | (npop << 8);
        __syncthreads();
        unsigned mask = 0u;
#pragma unroll
        for (int q = 0; q < kStepThreads / 64; ++q) { const int v = wc[q]; ncand_all += v >> 8; mask |= (unsigned)(v & 1) << q | (unsigned)((v >> 1) & 1) << (8 + q); }
        if (mask) first_sub = __builtin_ctz(mask);
        const bool single_pass = (P->nreps > 0 ? P->nreps : b) == 1;
        if (!single_pass) first_sub = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = tid + q * kStepThreads;
            if (stay[q] && c < b && (c >> 6) < first_sub) {
#pragma unroll
                for (int t = 0; t < NT; ++t) { bcur[t * B + c] = pb[q][t]; dcur[t * B + c] = pd[q][t]; }
            }
        }
        __syncthreads();                                   // (stage_rows reuses the slots)
    }
    const long long tk1 = clock64();
    const int nstaged_mt = prestage ? b : (first_sub >= 16 ? 0 : stage_rows(smem, SM, A, cand));
    if (cross_dma) {                                       // waves 1..7: the cross-Gram rows to LDS while wave 0 walks the block
        dma_copy_to_lds(A.cross_next, reinterpret_cast<float*>(smem + SM.cross_off), B * B, 1);
        if (wave != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else prefetch_cross_rows(smem, SM, A, nstaged_mt);
    prefetch_next_gram(A, prestage);
    int* wcnt_s = reinterpret_cast<int*>(smem + SM.wcnt_off);
    long long tk3 = 0, tk4 = 0, tk5 = 0;
    int nrounds = 0;
    if (wave == 0) {
    tk3 = clock64();


    const int nsub = (b + 63) / 64;
    const int nreps = P->nreps > 0 ? P->nreps : b;
    RngKey key{P->seed_lo, P->seed_hi, P->iter, 0u};

    // ---- DENSE blocks (every marker of a <= 128-marker block is in the model for some trait -- the default all-ones
    // multi-trait prior): sequential walk instead of speculative rounds, as in the single-trait sampler.  Every lane
    // evaluates ITS OWN marker against its own running rhs at every step -- no operand is broadcast; the step's marker
    // is lane l, whose per-trait alpha_old - alpha_new are broadcast with NT v_readlane and applied to the running rhs of
    // the whole block (NT x 2 registers per lane) with the marker's Gram row from LDS (read a step ahead).  A lane's
    // result is final at its own step: it keeps the w it was evaluated with and recomputes its update after the walk.
    bool dense_done = false;
    // Sampler I, every marker of the block in the model for every trait at entry (the reference's default prior keeps it
    // that way: the states with a trait missing have probability ~0): the walk SPECULATES that every delta stays 1 and
    // evaluates a marker with Rule L's linear form (mt1_linear_coeffs: A, c of all 64 markers of a section formed in
    // parallel, NT^2 fused multiply-adds per marker on the chain).  After a 64-marker section ONE full mt1_eval per lane
    // (all 64 markers at once, each with the w it was walked with) both verifies the speculation and yields the final
    // state -- by Rule L the linear form's own numbers whenever the speculation held; if any marker left the model for a
    // trait the section is walked again from its saved rhs with those markers evaluated the general way.
    if (METHOD != kMTBayesC2 && nreps == 1 && prestage && nstaged_mt == b && 5 * ncand_all >= 3 * b) {
        const float* rows = reinterpret_cast<const float*>(smem + SM.rows_off);
        float rhsq[NT][2], aq[NT][2], bq[NT][2], dq[NT][2], djq[2], wev[2][NT];
        double thrq[NT][2], zq[NT][2];
        MtPre<NT> Qq[2];
        MtConsts<NT> Kq[kPG ? 2 : 1];                               // (kPG: the two markers' own constants)
        auto KQ = [&](int q) -> const MtConsts<NT>& { if constexpr (kPG) return Kq[q]; else { (void)q; return K; } };
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = (64 * q + lane < b) ? 64 * q + lane : 0;
            djq[q] = lpf[c];                                        // (B <= 128: the draws are always parked in LDS)
            float lcq[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) lcq[t] = lpf[(1 + t) * B + c];
            if constexpr (kPG) Kq[q] = consts_of(c);
            Qq[q] = mt_precompute<METHOD, NT>(KQ(q), djq[q], lcq);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                rhsq[t][q] = rhs_lds[t * B + c]; aq[t][q] = acur[t * B + c]; bq[t][q] = bcur[t * B + c]; dq[t][q] = dcur[t * B + c];
                thrq[t][q] = lpd[t * B + c]; zq[t][q] = lpd[(NT + t) * B + c];
                wev[q][t] = 0.f;
            }
        }
        const bool speculate = is_sampler1(METHOD);
        // one marker evaluated in-lane from (w, its state at block entry, its draws)
        auto eval_own = [&](int q, const float (&w)[NT], float (&an)[NT], float (&bn)[NT], float (&dn)[NT], float (&Dl)[NT]) {
            const int c = (64 * q + lane < b) ? 64 * q + lane : 0;
            double thr[NT], z[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) { an[t] = aq[t][q]; bn[t] = bq[t][q]; dn[t] = dq[t][q]; Dl[t] = 0.f; thr[t] = thrq[t][q]; z[t] = zq[t][q]; }
            // (the shared prior table from registers -- v_cndmask trees instead of the LDS lookup -- was measured: 79 ms
            // per sweep instead of 55 at 3 traits x 20k x 100k; the LDS read overlaps the trait's arithmetic well enough)
            if constexpr (is_sampler1(METHOD)) mt1_eval<NT>(KQ(q), Qq[q], PriorMem{lpr_of(c), ls}, w, djq[q], thr, z, an, bn, dn, Dl);
            else mega_eval<NT>(K, Qq[q], w, djq[q], thr, z, an, bn, dn, Dl);
        };
        // the speculative conditionals = Rule L's linear form (mt1_linear_coeffs): A, c of the lane's own marker are formed once
        // per section; a step is NT^2 fused multiply-adds.  Dl = alpha_old - alpha_new
        auto eval_fast = [&](int q, const float (&w)[NT], const float (&Al)[NT][NT], const float (&cl)[NT], float (&bn)[NT], float (&Dl)[NT]) {
            mt1_linear_beta<NT>(Al, cl, w, bn);
#pragma unroll
            for (int k = 0; k < NT; ++k) Dl[k] = aq[k][q] - bn[k];
        };
        auto linear_of = [&](int q, float (&Al)[NT][NT], float (&cl)[NT]) {
            float b_old[NT];
            double zz[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) { b_old[t] = bq[t][q]; zz[t] = zq[t][q]; }
            mt1_linear_coeffs<NT>(KQ(q), Qq[q], djq[q], b_old, zz, Al, cl);
        };
        // 64-marker section q: eight steps per batch without a branch, the Gram rows read a batch ahead
        auto section = [&](auto qc, auto fastc, const float* grow, int nsteps) {
            constexpr int Q = decltype(qc)::value;
            constexpr bool FAST = decltype(fastc)::value;
            float Al[NT][NT], cl[NT], da[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) da[t] = djq[Q] * aq[t][Q];                                 // :82
            if constexpr (FAST) linear_of(Q, Al, cl);
            auto step = [&](int l, float c0, float c1) {
                float w[NT], an[NT], bn[NT], dn[NT], Dl[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) w[t] = rhsq[t][Q] + da[t];
                if constexpr (FAST) eval_fast(Q, w, Al, cl, bn, Dl);
                else eval_own(Q, w, an, bn, dn, Dl);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    wev[Q][t] = (lane == l) ? w[t] : wev[Q][t];      // lane l: what it was evaluated with
                    const float D = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Dl[t]), l));
                    if (Q == 0) rhsq[t][0] = fmaf(D, c0, rhsq[t][0]);                                   // D = 0: exact no-op
                    if (B > 64) rhsq[t][1] = fmaf(D, c1, rhsq[t][1]);
                }
            };
            constexpr int kBatch = FAST ? 8 : 2;
            float n0[kBatch], n1[kBatch];
            auto load = [&](int l0) {
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    n0[u] = (Q == 0) ? grow[(l0 + u) * B + lane] : 0.f;
                    n1[u] = (B > 64) ? grow[(l0 + u) * B + 64 + lane] : 0.f;
                }
            };
            int l = 0;
            if (nsteps >= kBatch) load(0);
#pragma unroll 1
            for (; l + kBatch <= nsteps; l += kBatch) {
                float c0[kBatch], c1[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u) { c0[u] = n0[u]; c1[u] = n1[u]; }
                if (l + 2 * kBatch <= nsteps) load(l + kBatch);
#pragma unroll
                for (int u = 0; u < kBatch; ++u) step(l + u, c0[u], c1[u]);
            }
#pragma unroll 1
            for (; l < nsteps; ++l) step(l, (Q == 0) ? grow[l * B + lane] : 0.f, (B > 64) ? grow[l * B + 64 + lane] : 0.f);
        };
        using std::integral_constant;
        // the same section with some markers (bit l of `slow`) evaluated the general way and the others speculatively: one
        // step per loop trip (used when the speculation missed, or when a marker is not in the model for every trait at entry)
        auto section_mixed = [&](auto qc, const float* grow, int nsteps, unsigned long long slow) {
            constexpr int Q = decltype(qc)::value;
            float Al[NT][NT], cl[NT], da[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) da[t] = djq[Q] * aq[t][Q];
            linear_of(Q, Al, cl);
            float g0 = (Q == 0) ? grow[lane] : 0.f;
            float g1 = (B > 64) ? grow[64 + lane] : 0.f;
#pragma unroll 1
            for (int l = 0; l < nsteps; ++l) {
                float w[NT], an[NT], bn[NT], dn[NT], Dl[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) w[t] = rhsq[t][Q] + da[t];
                if ((slow >> l) & 1ull) eval_own(Q, w, an, bn, dn, Dl);                               // (wave-uniform)
                else eval_fast(Q, w, Al, cl, bn, Dl);
                const float c0 = g0, c1 = g1;
                grow += B;                                           // next marker's row (one past the block: the overflow row)
                if (Q == 0) g0 = grow[lane];
                if (B > 64) g1 = grow[64 + lane];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    wev[Q][t] = (lane == l) ? w[t] : wev[Q][t];
                    const float D = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Dl[t]), l));
                    if (Q == 0) rhsq[t][0] = fmaf(D, c0, rhsq[t][0]);
                    if (B > 64) rhsq[t][1] = fmaf(D, c1, rhsq[t][1]);
                }
            }
        };
        auto run_section = [&](auto qc) {
            constexpr int Q = decltype(qc)::value;
            const int nsteps = (b < 64 * (Q + 1) ? b : 64 * (Q + 1)) - 64 * Q;
            if (nsteps <= 0) return;
            const float* grow = rows + 64 * Q * B;                   // (all rows staged in marker order: slot = marker)
            const int c = 64 * Q + lane;
            float an[NT], bn[NT], dn[NT], Dl[NT];
            if (speculate) {
                float rs[NT][2];
#pragma unroll
                for (int t = 0; t < NT; ++t) { rs[t][0] = rhsq[t][0]; rs[t][1] = rhsq[t][1]; }
                // markers that are not in the model for every trait at entry cannot be speculated on
                bool in_all = true;
#pragma unroll
                for (int t = 0; t < NT; ++t) in_all = in_all && (dq[t][Q] == 1.f);
                unsigned long long slow = __ballot(!in_all && c < b);
                if (__popcll(slow) * 4 > nsteps) slow = ~0ull;       // not a block to speculate on: everything the general way
                if (slow == 0ull) section(qc, integral_constant<bool, true>{}, grow, nsteps);
                else section_mixed(qc, grow, nsteps, slow);
                for (int pass = 0; pass < 64; ++pass) {
                    eval_own(Q, wev[Q], an, bn, dn, Dl);             // the exact evaluation of every marker of the section
                    bool ok = true;
#pragma unroll
                    for (int t = 0; t < NT; ++t) ok = ok && (dn[t] == 1.f);
                    // a speculated marker that leaves the model for a trait: its broadcast changes were wrong -- evaluate it
                    // (and whatever else looks wrong now) the general way and walk the section again from its saved rhs
                    const unsigned long long bad = __ballot(!ok && c < b) & ~slow;
                    if (bad == 0ull) break;
                    slow |= bad;
                    if (__popcll(slow) * 4 > nsteps) slow = ~0ull;   // (misses are not rare here: stop speculating)
#pragma unroll
                    for (int t = 0; t < NT; ++t) { rhsq[t][0] = rs[t][0]; rhsq[t][1] = rs[t][1]; }
                    section_mixed(qc, grow, nsteps, slow);
                    ++nrounds;                                       // (diagnostics: sections walked again)
                }
            } else {
                section(qc, integral_constant<bool, false>{}, grow, nsteps);
                eval_own(Q, wev[Q], an, bn, dn, Dl);
            }
            if (c < B)
#pragma unroll
                for (int t = 0; t < NT; ++t) { acur[t * B + c] = (c < b) ? an[t] : 0.f; bcur[t * B + c] = bn[t]; dcur[t * B + c] = dn[t]; }
        };
        run_section(integral_constant<int, 0>{});
        run_section(integral_constant<int, 1>{});
        dense_done = true;
    }
    if (dense_done && lane == 0) wcnt_s[14] = 1;

    const int s_first = (nreps == 1 && !dense_done) ? (first_sub < nsub ? first_sub : nsub) : 0;       // prefix skip (single pass only)
    for (int rep = 0; rep < (dense_done ? 0 : nreps); ++rep) {
        key.rep = (uint32_t)rep;
#pragma unroll 1
        for (int s = s_first; s < nsub; ++s) {
            const int c = 64 * s + lane;
            const bool valid = c < b;
            const int64_t j = j0 + (valid ? c : 0);
            const uint32_t marker = P->marker0 + (uint32_t)j;
            unsigned long long pending = __ballot(valid);
            const float dj = parked ? lpf[c] : A.xpx[j];
            double thr[NT], z[NT];
            float a_cur[NT], b_cur[NT], d_cur[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                a_cur[t] = acur[t * B + c]; b_cur[t] = bcur[t * B + c]; d_cur[t] = dcur[t * B + c];
                if (rep == 0) {
                    if (parked) { thr[t] = lpd[t * B + c]; z[t] = lpd[(NT + t) * B + c]; }
                    else { thr[t] = A.prep_d[(int64_t)t * p + j]; z[t] = A.prep_d[(int64_t)(NT + t) * p + j]; }
                }
                else {
                    const double u = draw_uniform(key, marker, (uint32_t)t);
                    thr[t] = (METHOD == kMTBayesC2) ? u : log((1.0 - u) / u);
                    z[t] = draw_normal(key, marker, (uint32_t)t);
                }
            }
            float lcm[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) lcm[t] = parked ? lpf[(1 + t) * B + c] : A.prep_f[(int64_t)t * p + j];
            const MtConsts<NT> Km = consts_of(valid ? c : 0);          // (kPG: this marker's own G-dependent constants)
            const MtPre<NT> Qm = mt_precompute<METHOD, NT>(Km, dj, lcm); // x'x-only terms, once per marker (SIMD over the sub-block)
            double T[kTS][kTV];
            if constexpr (kTab) mt2_load_tab<NT>(A.mt2_tab, p, j, T);
            while (true) {
                const bool live = valid && ((pending >> lane) & 1ull);
                float an[NT], bn[NT], dn[NT], Dl[NT];
                bool is_event = false;
#pragma unroll
                for (int t = 0; t < NT; ++t) { an[t] = a_cur[t]; bn[t] = b_cur[t]; dn[t] = d_cur[t]; Dl[t] = 0.f; }
                if (live) {
                    float w[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) w[t] = rhs_lds[t * B + c] + dj * a_cur[t];           // :82
                    if constexpr (is_sampler1(METHOD)) mt1_eval<NT>(Km, Qm, PriorMem{lpr_of(c), ls}, w, dj, thr, z, an, bn, dn, Dl);
                    else if constexpr (kTab) mt2_eval_tab<NT>(K, lpr_of(c), ls, w, T, thr[0], z, an, bn, dn, Dl);
                    else if constexpr (METHOD == kMTBayesC2) mt2_eval<NT>(K, lpr_of(c), ls, w, dj, thr[0], z, an, bn, dn, Dl);
                    else mega_eval<NT>(K, Qm, w, dj, thr, z, an, bn, dn, Dl);
#pragma unroll
                    for (int t = 0; t < NT; ++t) is_event = is_event || (Dl[t] != 0.f);
                }
                ++nrounds;
                const unsigned long long m = __ballot(is_event) & pending;
                const int k = m ? __builtin_ctzll(m) : 64;
                if (live && lane <= k) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) { a_cur[t] = an[t]; b_cur[t] = bn[t]; d_cur[t] = dn[t]; }
                }
                if (k == 64) break;
                pending = (k == 63) ? 0ull : (pending & ~((2ull << k) - 1ull));
                float D[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) D[t] = __shfl(Dl[t], k, 64);
                apply_gram_row<NT>(smem, SM, A, 64 * s + k, D, lane);                               // :311,317
                if (pending == 0ull) break;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) { acur[t * B + c] = a_cur[t]; bcur[t * B + c] = b_cur[t]; dcur[t * B + c] = d_cur[t]; }
        }
    }

    tk4 = clock64();
    int base = 0;
#pragma unroll 1
    for (int s = s_first; s < nsub; ++s) {                    // (no change before the first candidate's sub-block)
        const int c = 64 * s + lane;
        bool changed = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) changed = changed || ((c < b) && astart[t * B + c] != acur[t * B + c]);
        const unsigned long long cm = __ballot(changed);
        if (changed) reinterpret_cast<int*>(smem + SM.log_off)[base + __popcll(cm & ((1ull << lane) - 1ull))] = c;
        base += __popcll(cm);
    }
    if (lane == 0) wcnt_s[15] = base;
    tk5 = clock64();
    }   // wave 0
    __syncthreads();
    const int nfin = wcnt_s[15];
    if (A.b_next > 0 && cross_dma && wcnt_s[14] != 0) {
        // dense walk with the cross-Gram rows in LDS: every marker is an entry (alpha_old - alpha_new = 0: exact no-op); one
        // thread per (trait, column of the next block), the chain in marker order as in corr_phase
        const float* crossL = reinterpret_cast<const float*>(smem + SM.cross_off);
        for (int i = tid; i < NT * B; i += kStepThreads) rhs_lds[i] = astart[i] - acur[i];
        __syncthreads();
        for (int i = tid; i < NT * B; i += kStepThreads) {
            const int t = i / B, c = i - t * B;
            const float* dl = rhs_lds + t * B;
            float corr = 0.f;
            int e = 0;
#pragma unroll 1
            for (; e + 8 <= b; e += 8) {
                const float4 d0 = *reinterpret_cast<const float4*>(dl + e), d1 = *reinterpret_cast<const float4*>(dl + e + 4);
                float g[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) g[u] = crossL[(e + u) * B + c];
                corr = fmaf(d0.x, g[0], corr); corr = fmaf(d0.y, g[1], corr); corr = fmaf(d0.z, g[2], corr); corr = fmaf(d0.w, g[3], corr);
                corr = fmaf(d1.x, g[4], corr); corr = fmaf(d1.y, g[5], corr); corr = fmaf(d1.z, g[6], corr); corr = fmaf(d1.w, g[7], corr);
            }
            for (; e < b; ++e) corr = fmaf(dl[e], crossL[e * B + c], corr);
            A.corr_out[i] = corr;
        }
    } else if (A.b_next > 0) corr_phase<NT>(smem, SM, A, nfin);
    // ---- global stores LAST (a barrier after a global store waits for the store): the change list for the next update
    // role, then the block's state (beta / delta of every marker are new draws; alpha changes only where an event happened)
    {
        const int* fin = reinterpret_cast<const int*>(smem + SM.log_off);
        for (int e = tid; e < nfin; e += kStepThreads) {
            const int ce = fin[e];
            A.ev_out->idx[e] = (int32_t)(j0 + ce);
#pragma unroll
            for (int t = 0; t < NT; ++t) A.ev_out->delta[t][e] = astart[t * B + ce] - acur[t * B + ce];
        }
    }
    for (int c = tid; c < b; c += kStepThreads) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float a_fin = acur[t * B + c];
            if (a_fin != astart[t * B + c]) A.alpha[(int64_t)t * p + j0 + c] = a_fin;
            A.beta[(int64_t)t * p + j0 + c] = bcur[t * B + c];
            delta[(int64_t)t * p + j0 + c]  = dcur[t * B + c];
        }
    }
    if (tid == 0) {
        A.ev_out->count = nfin;
        atomicAdd(&A.counters[0], (unsigned long long)nfin);
        atomicAdd(&A.counters[2], (unsigned long long)(tk1 - tk0));      // phase cycle counts (diagnostics)
        atomicAdd(&A.counters[4], (unsigned long long)(tk3 - tk1));
        atomicAdd(&A.counters[5], (unsigned long long)(tk4 - tk3));
        atomicAdd(&A.counters[6], (unsigned long long)(tk5 - tk4));
        atomicAdd(&A.counters[7], (unsigned long long)nrounds);
    }
}

// ---------------------------------------------------------------------------------------------
// The fused step.  grid = 1 + nrg*ncg, block = 512.
// ---------------------------------------------------------------------------------------------
struct UpdateArgs {
    const float* r_in; float* r_out;
    const Events* ev;             // changes to apply (block k-2)
    int64_t j0; int b;            // block whose partial RHS is formed (b = 0: none)
    int nslices, nrg, ncg;
    int spg;                      // slices (waves that stream) per row group: <= 8
    double* partials; int bstride;
    int quiet_xcd;                // 1: ids = 0 mod 8 (the sampler's XCD) do no update work
    unsigned long long* dbg;      // phase cycle counters (diagnostics) or NULL
    int* sync_now; int* sync_next; // [nrg] arrival counters of the cooperative dense apply: this launch's / the next one's (or NULL)
    int dbg_throttle;             // > 1: only every n-th update workgroup runs (timing experiments; results wrong)
};
template <class CX>
struct UpdateArgsT : UpdateArgs {
    CX cx;                        // genotype storage accessor
};

// COOP: the update role may split a dense change list among the column groups of a row group (see update_role); a separate
// instantiation so that the steady-state kernel's code is exactly the one without it (its presence alone cost 1 %).
template <int METHOD, int NT, class CX, bool COOP = false, bool DENSE = false>
__global__ __launch_bounds__(kStepThreads) void k_block_step(UpdateArgsT<CX> U, SamplerArgs S, int do_sample)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (blockIdx.x == 0) {
        if (!do_sample) return;
#ifdef JWAS_HIP_POISON_LDS
        // development builds (-DJWAS_HIP_POISON_LDS=0x11111111): the sampler starts from a known-bad LDS image instead of
        // whatever the previous kernel left there -- a read of something this launch did not write shows up as a parity failure
        for (int i = threadIdx.x; i < 150000 / 4; i += kStepThreads) reinterpret_cast<unsigned*>(smem)[i] = (unsigned)(JWAS_HIP_POISON_LDS);
        __syncthreads();
#endif
        if constexpr (is_mt_method(METHOD)) sampler_role_mt<METHOD, NT>(smem, S);
        else sampler_role_st<METHOD, DENSE>(smem, S);
        return;
    }
    int w = blockIdx.x - 1;
    if (U.quiet_xcd) {
        // Speed heuristic only (never correctness): workgroup ids are observed to round-robin over the 8 XCDs,
        // so ids = 0 mod 8 share the sampler's XCD/L2.  Leaving them idle keeps that L2 free of the streaming
        // traffic, which shortens every dependent load of the sampler chain.
        if constexpr (DENSE) {
            // ... and ONE of them (id 8) pulls what the NEXT launch's sampler will read -- the next block's Gram and its
            // cross-Gram, 2 MB at 512 markers -- into that L2: dense_big_st moves ~1.5 MB per block through one CU, and with
            // HBM latency under the update role's streaming (~5 us) a CU's 63 loads in flight per wave cap it at ~25 GB/s.
            if (blockIdx.x == 8 && do_sample && (S.bsz == 256 || S.bsz == 512) && !S.dense_big_off) {
                const int nl_g = (S.gram_next != nullptr) ? (S.b_next * S.b_next + 31) / 32 : 0;
                const int nl_c = (S.cross_after != nullptr) ? S.lines_after : 0;
                float sink = 0.f;
                for (int l0 = 0; l0 < nl_g + nl_c; l0 += 8 * kStepThreads) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        int l = l0 + u * kStepThreads + (int)threadIdx.x;
                        l = l < nl_g + nl_c ? l : nl_g + nl_c - 1;
                        v[u] = (l < nl_g) ? S.gram_next[(int64_t)l * 32] : S.cross_after[(int64_t)(l - nl_g) * 32];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) sink += v[u];
                }
                asm volatile("" ::"v"(sink));
                return;
            }
        }
        if ((blockIdx.x & 7) == 0) return;
        w = (int)(blockIdx.x - 1) - (int)((blockIdx.x - 1) >> 3);
    }
    if (w >= U.nrg * U.ncg) return;
    if (U.dbg_throttle > 1 && (w % U.dbg_throttle) != 0) return;       // timing experiments only
    update_role<NT, CX, COOP>(smem, w % U.nrg, w / U.nrg, U.cx, U.r_in, U.r_out, U.ev, U.j0, U.b,
                              U.nslices, U.nrg, U.ncg, U.partials, U.bstride, U.spg, U.sync_now, U.sync_next, U.dbg);
}

// Cross-Gram of consecutive blocks, exact (fp64-accumulated): C[a][c] = x_{jp+a}' x_{j0+c}.
// grid = (bsize, nblocks-1), block = 256; workgroup (a, i) writes row a of cross block i+1.
template <class CX>
__global__ __launch_bounds__(256) void k_cross_f64(CX cx, int64_t p, int bsize,
                                                   float* __restrict__ cross, const int64_t* __restrict__ starts = nullptr)
{
    const int64_t ld = cx.ld;
    const int64_t blk = (int64_t)blockIdx.y + 1;
    const int64_t j0 = starts ? starts[blk] : blk * bsize, jp = starts ? starts[blk - 1] : j0 - bsize;
    const int b = starts ? (int)(starts[blk + 1] - j0) : (int)((j0 + bsize <= p) ? bsize : (p - j0));
    const int a = blockIdx.x;
    if (a >= (int)(j0 - jp)) return;                   // (explicit starts: the previous block may be shorter than bsize)
    float* C = cross + blk * (int64_t)bsize * bsize;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < b; c += 4) {
        double s = 0.0;
        for (int64_t i = (int64_t)lane * 4; i < ld; i += 256) {
            const float4 qa = cx.load4(jp + a, i);
            const float4 qc = cx.load4(j0 + c, i);
            const float4 wv = *reinterpret_cast<const float4*>(cx.w + i);
            s = fma((double)qa.x, (double)(qc.x * wv.x), s);
            s = fma((double)qa.y, (double)(qc.y * wv.y), s);
            s = fma((double)qa.z, (double)(qc.z * wv.z), s);
            s = fma((double)qa.w, (double)(qc.w * wv.w), s);
        }
        s = wave_sum(s);
        if (lane == 0) C[(int64_t)a * b + c] = (float)s;
    }
}

// ---------------------------------------------------------------------------------------------
// INDEPENDENT-BLOCK mode (BayesABC_block_independent!, BayesABC.jl:190-255; BayesR.jl:195-273;
// MTBayesABC.jl:335-440): every block's RHS comes from the SAME residual snapshot, the blocks are
// sampled independently (here: all at once, one workgroup each), and the residual is reconciled
// afterwards with r += sum_b X_b * (alpha_old_b - alpha_new_b) in (block, marker) order.
// ---------------------------------------------------------------------------------------------
// Block RHS of ALL blocks from the snapshot.  grid = (nrg*ncg, nblocks), block = 512.
template <int NT, class CX>
__global__ __launch_bounds__(kStepThreads) void k_indep_rhs(UpdateArgsT<CX> U, int64_t p, int bsz, int64_t pstride)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t blk = blockIdx.y;
    const int64_t j0 = blk * bsz;
    const int b = (int)((j0 + bsz <= p) ? bsz : p - j0);
    const int ncg = U.ncg < b ? U.ncg : b;
    const int w = blockIdx.x;
    if (w >= U.nrg * ncg) return;
    update_role<NT, CX>(smem, w % U.nrg, w / U.nrg, U.cx, U.r_in, nullptr, U.ev, j0, b,
                        U.nslices, U.nrg, ncg, U.partials + blk * pstride, U.bstride, U.spg);
}

// All blocks sampled concurrently.  grid = nblocks, block = 512, dynamic LDS as k_block_step.
template <int METHOD, int NT, bool DENSE = false>
__global__ __launch_bounds__(kStepThreads) void k_indep_sample(SamplerArgs S, int64_t pstride, Events* ev_all)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t blk = blockIdx.x;
    S.j0 = blk * S.bsz;
    S.b = (int)((S.j0 + S.bsz <= S.p) ? S.bsz : S.p - S.j0);
    S.partials += blk * pstride;
    S.gram += blk * (int64_t)S.bsz * S.bsz;
    S.ev_out = ev_all + blk;
    S.b_next = 0;                                   // no lookahead correction in this mode
    if constexpr (is_mt_method(METHOD)) sampler_role_mt<METHOD, NT>(smem, S);
    else sampler_role_st<METHOD, DENSE>(smem, S);
}

// Exclusive scan of the per-block change counts.  grid = 1, block = 1024.
__global__ __launch_bounds__(1024) void k_indep_scan(const Events* __restrict__ ev_all, int nblocks,
                                                     int32_t* __restrict__ offs, int32_t* __restrict__ total)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + tid;
        const int v = i < nblocks ? ev_all[i].count : 0;
        int x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int wpre = 0, tot = 0;
        for (int q = 0; q < 16; ++q) { if (q < wave) wpre += wsum[q]; tot += wsum[q]; }
        if (i < nblocks) offs[i] = carry + wpre + x - v;
        __syncthreads();
        if (tid == 0) carry += tot;
        __syncthreads();
    }
    if (tid == 0) *total = carry;
}

// Compact the per-block change lists into one list in (block, marker) order.  grid = nblocks, block = 256.
__global__ __launch_bounds__(256) void k_indep_gather(const Events* __restrict__ ev_all, const int32_t* __restrict__ offs,
                                                      int nt, int32_t* __restrict__ idx_all, float* __restrict__ delta_all,
                                                      int64_t dstride)
{
    const Events* ev = ev_all + blockIdx.x;
    const int ne = ev->count, off = offs[blockIdx.x];
    for (int e = threadIdx.x; e < ne; e += 256) {
        idx_all[off + e] = ev->idx[e];
        for (int t = 0; t < nt; ++t) delta_all[t * dstride + off + e] = ev->delta[t][e];
    }
}

}  // namespace jw
