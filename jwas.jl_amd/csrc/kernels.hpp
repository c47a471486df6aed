// kernels.hpp -- gfx950 (CDNA4, wave64) kernels of the marker-effect Gibbs sweep.
//
// Data layout in HBM (DESIGN.md "Layout"):
//   X      float [p][ld]   marker-major; ld = n rounded up to 256 rows, pad rows are zero
//   r      float [t][ld]   residual vectors, pad rows zero
//   alpha/beta float [t][p], delta float [t][p] (0/1) or int32 [p] (BayesR classes)
//   gram   float           consecutive b x b row-major symmetric blocks, one per marker block
//
// One sweep = for each marker block B_i of b columns:
//   k_update_partial  (all CUs, HBM-bound): every wavefront owns a 256-row slice of r in
//                     registers; it first applies the previous block's effect changes
//                     r += X[:,events] * d  (sparse exit update of BayesABC.jl:181-185), then forms
//                     the partial block RHS X_b[slice,:]' r[slice] (block_rhs!,
//                     tools4genotypes.jl:59-78) with wave64 shuffle reductions, fp64-accumulated.
//   k_sample_block    (one workgroup): reduces the row-group partials to rhs_b, then ONE wavefront
//                     runs the exact single-site chain of the block (BayesABC.jl:153-179) by
//                     speculative parallel evaluation: all lanes evaluate their marker against the
//                     current rhs; the first lane whose effect changes commits, its Gram column
//                     corrects every rhs (BayesABC.jl:169,172), the rest re-evaluate.  Lanes before
//                     the first change are final, so the result is the sequential chain's.
//
// Arithmetic contract (must match oracle/jwas_oracle.c): compiled with -ffp-contract=off; every
// fused operation is an explicit fmaf/fma; transcendentals are evaluated in double and rounded to
// the type the reference holds; inner products accumulate exact fp32 products in double.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rng.hpp"

// non-template kernels live in this header: the translation unit that owns them (jwas_hip.hip) leaves this empty, any other one
// (resident_launch.hip) defines it `static` so that the two object files do not both export them
#ifndef JW_PLAIN_KERNEL
#define JW_PLAIN_KERNEL
#endif

namespace jw {

constexpr int kSliceRows = 256;      // rows of r owned by one workgroup of k_update_partial
constexpr int kMaxBlock  = 1024;     // largest marker block
constexpr int kMaxT      = 4;        // traits
constexpr int kMaxStates = 16;

enum Method { kBayesC = 0, kBayesB = 1, kBayesR = 2, kMTBayesC1 = 3, kMTBayesC2 = 4, kMegaBayesC = 5, kMTBayesB1 = 6, kMTBayesB2 = 7, kMegaBayesB = 8 };
__host__ __device__ constexpr bool is_mt_method(int m) { return m >= kMTBayesC1; }
// Gibbs sampler I; kMTBayesB1 = the same sampler with a t x t effect covariance PER MARKER (multi-trait BayesA/B:
// locus_effect_variances[marker], MTBayesABC.jl:66,86-90)
__host__ __device__ constexpr bool is_sampler1(int m) { return m == kMTBayesC1 || m == kMTBayesB1; }
// sampler II (joint state); kMTBayesB2 = sampler II with a t x t effect covariance per marker (multi-trait BayesA/B, multi_trait_sampler = :II)
__host__ __device__ constexpr bool is_sampler2(int m) { return m == kMTBayesC2 || m == kMTBayesB2; }
__host__ __device__ constexpr bool has_marker_cov(int m) { return m == kMTBayesB1 || m == kMTBayesB2 || m == kMegaBayesB; }
// megaBayesABC! (t independent single-trait chains); kMegaBayesB: with every marker's own diagonal variances (BayesA/B)
__host__ __device__ constexpr bool is_mega(int m) { return m == kMegaBayesC || m == kMegaBayesB; }

// Effect changes of one marker block, consumed by the next k_update_partial.
struct Events {
    int32_t count;
    // single-trait header: the first 7 changes again, in the same 64-byte line as the count, so that the consumer's
    // count -> index -> column chain of dependent loads is one step shorter in the common case (<= 7 changes per block)
    int32_t hidx[7];
    float   hdelta[8];
    int32_t idx[kMaxBlock];                 // local column index of the changed marker
    float   delta[kMaxT][kMaxBlock];        // alpha_old - alpha_new per trait (the axpy coefficient)
};
// How the update role reads a change list.  EvPlain: one block's Events.  EvGroup (grouped launches, sweep.hpp k_group_step:
// single trait): the merged list of a GROUP of consecutive blocks -- the header line of an Events (count and the first 7 changes)
// plus index / coefficient arrays of the group's capacity.
struct EvPlain {
    const Events* ev;
    __device__ __forceinline__ int count() const { return ev->count; }
    __device__ __forceinline__ int idx(int e) const { return ev->idx[e]; }
    __device__ __forceinline__ float delta(int t, int e) const { return ev->delta[t][e]; }
    __device__ __forceinline__ int hidx(int u) const { return ev->hidx[u]; }
    __device__ __forceinline__ float hdelta(int u) const { return ev->hdelta[u]; }
};
struct EvGroup {
    const Events* hdr; const int32_t* gidx; const float* gdelta;
    __device__ __forceinline__ int count() const { return hdr->count; }
    __device__ __forceinline__ int idx(int e) const { return gidx[e]; }
    __device__ __forceinline__ float delta(int, int e) const { return gdelta[e]; }
    __device__ __forceinline__ int hidx(int u) const { return hdr->hidx[u]; }
    __device__ __forceinline__ float hdelta(int u) const { return hdr->hdelta[u]; }
};

// Per-sweep scalars, device resident (rewritten before every sweep).
struct DevParams {
    int32_t  method, ntraits, nreps;
    uint32_t iter, seed_lo, seed_hi, marker0, pad0;
    float    vare[16], var_effect[16];
    float    Rinv[16], Ginv[16];            // t x t inverses (host: double Gauss-Jordan -> float)
    double   pi;
    double   pi4[4], gamma[4];
    double   log_prior[kMaxStates];
    const float*  var_vec;                  // p, BayesB
    const float*  var_mat;                  // p x t x t, multi-trait BayesA/B: per-marker effect covariances
    float*        ginv_mat;                 // p x t x t, their inverses (written by k_prepare, read by the sampler)
    const double* pi_vec;                   // p
    const double* pi_mat;                   // p x 4
};

// ---------------------------------------------------------------------------------------------
// Genotype storage accessors.  Every kernel that reads genotypes is templated on one of these.
//   DenseCols : fp32, marker-major, column stride ld (a multiple of 256 rows, pad rows zero).
//   PackedCols: the reference's 2-bit packed backend (Packed2BitBackend, streaming_genotypes.jl:7-25):
//               marker j occupies bytes [j*sb, (j+1)*sb); individual i is in byte i>>2 at bit shift (i&3)<<1;
//               codes 0/1/2 = genotype, 3 = missing -> the marker mean; decoded exactly as decode_marker!
//               (streaming_genotypes.jl:978-1002):  v = code == 3 ? mu : Float32(code);  x = centered ? v - mu : v.
//               On the device sb = ld/4 (>= cld(n,4), padded to 64 bytes); rows >= n decode to 0.
// load4(j, row): rows row..row+3 (row % 4 == 0) of column j;  load1(j, row): one element.
// ---------------------------------------------------------------------------------------------
struct DenseCols {
    const float* X; int64_t ld;
    const float* w; int32_t weighted;      // residual weights R^-1 (ld floats, pad rows 0; all ones when unweighted)
    // streaming interface of the update role: raw load now, decode when consumed; kDepth = register batches kept
    // in flight per wave (dense: one 8 KB batch per wave already saturates HBM, see update_role)
    typedef float4 Raw;
    typedef float4 SRaw;
    static constexpr int kDepth = 1;
    static constexpr bool kCoopApply = true;          // update role: the cooperative dense apply is available (load1 is one dword)
    static constexpr bool kWide = false;              // (the update role's 256-row slices, four rows per lane)
    // column stream of the update role: element i = rows row..row+3 of marker j0 + i*jstride
    struct Stream {
        const float* p; int64_t stride;
        // non-temporal: every element of X is read exactly once per sweep, keeping it out of the caches' replacement
        // order is worth 3-5 % of streaming bandwidth (scripts/micro/read_bw.hip: 6.37 -> 6.69 TB/s)
        __device__ __forceinline__ SRaw load_raw(int i) const
        {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p + (int64_t)i * stride));
            return float4{t.x, t.y, t.z, t.w};
        }
        __device__ __forceinline__ float load_mean(int) const { return 0.f; }
        static __device__ __forceinline__ unsigned flags(const SRaw&) { return 0u; }
        __device__ __forceinline__ float4 decode_fast(const SRaw& r, float) const { return r; }
        __device__ __forceinline__ float4 decode_patch(const SRaw& r, float) const { return r; }
    };
    __device__ __forceinline__ Stream stream(int64_t j0, int jstride, int64_t row) const
    {
        return Stream{X + j0 * ld + row, (int64_t)jstride * ld};
    }
    __device__ __forceinline__ float4 load4(int64_t j, int64_t row) const
    {
        return *reinterpret_cast<const float4*>(X + j * ld + row);
    }
    __device__ __forceinline__ float load1(int64_t j, int64_t row) const { return X[j * ld + row]; }
};

struct PackedCols {
    const uint8_t* Q; int64_t ld;            // ld = padded row count; byte stride of a marker = ld / 4
    const float* mean; int64_t n; int32_t centered;
    const float* w; int32_t weighted;        // residual weights R^-1 (as DenseCols)
    __device__ __forceinline__ float dec(unsigned code, float mu) const
    {
        const float v = (code == 3u) ? mu : (float)code;           // streaming_genotypes.jl:994
        return centered ? v - mu : v;                              // :995
    }
    // a lane's share of a column is ONE byte, so many columns can be in flight for the price of a few registers:
    // the streaming loop is latency-bound, not bandwidth-bound, and wants depth
    struct Raw { unsigned byte; float mu; };
    static constexpr int kDepth = 8;
    static constexpr bool kCoopApply = false;
    static constexpr bool kWide = true;               // update role: 1024-row slices, one dword = 16 rows per lane (update_role_wide)
    __device__ __forceinline__ Raw load_raw(int64_t j, int64_t row) const
    {
        return Raw{Q[j * (ld >> 2) + (row >> 2)], mean[j]};
    }
    __device__ __forceinline__ float4 decode(const Raw& r, int64_t row) const
    {
        // Float32(code) - mu for the four codes of the byte; the (rare) missing code 3 is patched under a branch the
        // whole wave skips when no lane holds one
        const unsigned b = r.byte;
        const float sub = centered ? r.mu : 0.f;
        float4 x;
        x.x = (float)(b & 3u) - sub; x.y = (float)((b >> 2) & 3u) - sub;
        x.z = (float)((b >> 4) & 3u) - sub; x.w = (float)(b >> 6) - sub;
        const unsigned miss = b & (b >> 1) & 0x55u;                // bit 2k set <=> code k == 3
        if (miss) {
            const float mv = centered ? 0.f : r.mu;                // v = mu  ->  x = centered ? mu - mu : mu
            if (miss & 0x01u) x.x = mv;
            if (miss & 0x04u) x.y = mv;
            if (miss & 0x10u) x.z = mv;
            if (miss & 0x40u) x.w = mv;
        }
        if (row + 4 > n) {                                         // pad rows (last slice only)
            if (row >= n) x.x = 0.f;
            if (row + 1 >= n) x.y = 0.f;
            if (row + 2 >= n) x.z = 0.f;
            if (row + 3 >= n) x.w = 0.f;
        }
        return x;
    }
    __device__ __forceinline__ float4 load4(int64_t j, int64_t row) const { return decode(load_raw(j, row), row); }
    // column stream of the update role (no pad-row masking: the consumer multiplies pad rows by r = 0).
    // The marker means are NOT fetched per column: a scalar load in the streaming loop forces s_waitcnt lgkmcnt(0)
    // (SMEM returns out of order), i.e. a full memory round trip per batch.  Lane i of the wave loads the mean of
    // stream element i0 + i once per 64-column chunk; columns read it with v_readlane.
    typedef unsigned SRaw;
    struct Stream {
        const uint8_t* q; int64_t qstride; const float* m; int64_t mstride; int32_t centered;
        __device__ __forceinline__ SRaw load_raw(int i) const { return q[(int64_t)i * qstride]; }
        __device__ __forceinline__ float load_mean(int i) const { return m[(int64_t)i * mstride]; }
        static __device__ __forceinline__ unsigned flags(const SRaw& b) { return b & (b >> 1) & 0x55u; }   // any code == 3
        __device__ __forceinline__ float4 decode_fast(const SRaw& b, float mu) const           // no missing code in the byte
        {
            const float sub = centered ? mu : 0.f;
            float4 x;
            x.x = (float)(b & 3u) - sub; x.y = (float)((b >> 2) & 3u) - sub;
            x.z = (float)((b >> 4) & 3u) - sub; x.w = (float)(b >> 6) - sub;
            return x;
        }
        __device__ __forceinline__ float4 decode_patch(const SRaw& b, float mu) const
        {
            float4 x = decode_fast(b, mu);
            const unsigned miss = flags(b);
            const float mv = centered ? 0.f : mu;
            x.x = (miss & 0x01u) ? mv : x.x; x.y = (miss & 0x04u) ? mv : x.y;
            x.z = (miss & 0x10u) ? mv : x.z; x.w = (miss & 0x40u) ? mv : x.w;
            return x;
        }
    };
    __device__ __forceinline__ Stream stream(int64_t j0, int jstride, int64_t row) const
    {
        return Stream{Q + j0 * (ld >> 2) + (row >> 2), (int64_t)jstride * (ld >> 2), mean + j0, (int64_t)jstride, centered};
    }
    __device__ __forceinline__ float load1(int64_t j, int64_t row) const
    {
        if (row >= n) return 0.f;
        const unsigned byte = Q[j * (ld >> 2) + (row >> 2)];
        return dec((byte >> ((row & 3) << 1)) & 3u, mean[j]);
    }
};

// ---------------------------------------------------------------------------------------------
// wave64 reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* lds /* [nwaves][NV] */, int nwaves)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) lds[wave * NV + i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0.0;
        for (int w = 0; w < nwaves; ++w) s += lds[w * NV + i];
        v[i] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// Update/partial role (see sweep.hpp): apply a block's events to the residual, then the partial block RHS.
//
// grid = (nrg, ncg), block = 512 (8 waves).  Row group rg owns 8 consecutive 256-row slices, one
// per wave (lane l holds rows 4l..4l+3 of its slice as a float4 / four doubles in registers).
// Column group g owns columns g, g+ncg, ... of the block.  All 8 waves walk the same columns, so
// one column step reads 8 KB contiguous of X; the 8 per-wave dot products are combined through
// LDS and ONE partial per (column, row group) goes to HBM: partials[(t*nrg + rg)*bstride + c].
//
// Loads are software-pipelined (two register batches of U float4 per wave: ~16 KB in flight per
// wave, 128 KB per workgroup), and the U per-lane column sums are reduced with a transposed
// butterfly (U-1 + 3 shuffle-adds instead of 6U).
//
// The sparse exit update r += X[:,events]*d (BayesABC.jl:181-185) is recomputed by every column
// group of a row group (reads r_in, never r_out, so there is no read/write race between groups);
// only column group 0 stores the updated slice to r_out.
// ---------------------------------------------------------------------------------------------
constexpr int kRowGroupSlices = 8;
constexpr int kColChunk = 64;        // columns per LDS flush
constexpr int kU = 8;                // columns per register batch

// Transposed butterfly: in = kU per-lane values (one per column); out: lane l holds the wave-wide
// sum of column ((l>>5)&1)*4 + ((l>>4)&1)*2 + ((l>>3)&1) in every lane of its 8-lane group.
__device__ __forceinline__ double butterfly8(const double (&v)[kU], int lane)
{
    double a[4], b[2], c;
    const bool h5 = (lane & 32) != 0, h4 = (lane & 16) != 0, h3 = (lane & 8) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double send = h5 ? v[i] : v[4 + i];
        const double keep = h5 ? v[4 + i] : v[i];
        a[i] = keep + __shfl_xor(send, 32, 64);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double send = h4 ? a[i] : a[2 + i];
        const double keep = h4 ? a[2 + i] : a[i];
        b[i] = keep + __shfl_xor(send, 16, 64);
    }
    {
        const double send = h3 ? b[0] : b[1];
        const double keep = h3 ? b[1] : b[0];
        c = keep + __shfl_xor(send, 8, 64);
    }
    c += __shfl_xor(c, 4, 64);
    c += __shfl_xor(c, 2, 64);
    c += __shfl_xor(c, 1, 64);
    return c;
}

// ---------------------------------------------------------------------------------------------
// K_F: apply the last block's events (r_in -> r_out, may alias) and reduce r'r / sum(r) per
// slice.  out[slice][NT*NT + NT]
// ---------------------------------------------------------------------------------------------
// A list of effect changes in global memory: count, column indices, per-trait coefficients (trait stride dstride).
struct EventList {
    const int32_t* count; const int32_t* idx; const float* delta; int64_t dstride;
};
__host__ __device__ inline EventList event_list(const Events* ev)
{
    return EventList{&ev->count, ev->idx, &ev->delta[0][0], (int64_t)kMaxBlock};
}

template <int NT, class CX>
__global__ __launch_bounds__(256) void k_finish(CX cx, const float* r_in,
                                                float* r_out,
                                                EventList ev, double* __restrict__ out)
{
    const int64_t ld = cx.ld;
    __shared__ double red[4 * (NT * NT + NT)];
    const int tid = threadIdx.x;
    const int64_t row = (int64_t)blockIdx.x * kSliceRows + tid;
    float rv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) rv[t] = r_in[t * ld + row];
    const int ne = *ev.count;
    // sequential fmaf in list order (= marker order); 16 independent column loads in flight per thread
    for (int e0 = 0; e0 < ne; e0 += 16) {
        float x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = cx.load1(ev.idx[e0 + u < ne ? e0 + u : ne - 1], row);
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (e0 + u < ne)
#pragma unroll
                for (int t = 0; t < NT; ++t) rv[t] = fmaf(ev.delta[t * ev.dstride + e0 + u], x[u], rv[t]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) r_out[t * ld + row] = rv[t];
    // r_a' R^-1 r_c and 1' R^-1 r_a (sample_variance with invweights, variance_components.jl:82-98; weights = 1: r'r, sum r)
    const double wr = (double)cx.w[row];
    double v[NT * NT + NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
#pragma unroll
        for (int c = 0; c < NT; ++c) v[a * NT + c] = ((double)rv[a] * (double)rv[c]) * wr;
        v[NT * NT + a] = (double)rv[a] * wr;
    }
    block_sum<NT * NT + NT>(v, red, 4);
    if (tid == 0)
#pragma unroll
        for (int i = 0; i < NT * NT + NT; ++i) out[(int64_t)blockIdx.x * (NT * NT + NT) + i] = v[i];
}

// ---------------------------------------------------------------------------------------------
// scalar samplers (must mirror oracle/jwas_oracle.c operation for operation)
//
// Everything about a marker that does not depend on the running block RHS (its two random draws,
// the prior logs, lhs, 1/lhs, log lhs ...) is computed ONCE PER SWEEP for all p markers in parallel
// by k_prepare and stored struct-of-arrays in prep_d [kPrepD][p] / prep_f [kPrepF][p]; the serial
// sampler wave only loads them.  (Within-block repetitions > 0 recompute them in place.)
// ---------------------------------------------------------------------------------------------
constexpr int kPrepD = 17;
constexpr int kPrepF = 7;          // (rows 5, 6: Rule D's c1, c0 -- sweeps with a uniform pi = 0 only)

// Floats as ordered integers (the bisections below walk the float number line).
__device__ __forceinline__ uint32_t float_key(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
constexpr float kFltMax = 3.402823466e+38f;

__device__ __forceinline__ float logf_via_double(float x) { return (float)log((double)x); }

// BayesA/B/C -- bayesabc_update_marker! (BayesABC.jl:24-58).  The inclusion test
// u < 1/(1+exp(lp0-l1)) is evaluated in the equivalent log-odds form (lp0-l1) < log((1-u)/u),
// so no exp sits on the serial path.
struct AbcMarker {
    float  d, invLhs, c1, beta_excl;        // c1 = log(lhs) + log(var_j) (Float32 sum of :40)
    double lp0, lp1, thr, zs;               // zs = z*sqrt(1/lhs)
    __device__ __forceinline__ void prepare(float d_, float var_, double pi_, float ie, double u, double z)
    {
        d = d_;
        const float iv = 1.0f / var_;                       // invVarEffects[j]   :70
        const float lv = logf_via_double(var_);             // logVarEffects[j]   :71
        lp0 = log(pi_);                                     // logPi[j]           :67
        lp1 = log(1.0 - pi_);                               // logPiComp[j]       :68
        const float lhs = d_ * ie + iv;                     // :37
        invLhs = 1.0f / lhs;                                // :38
        c1  = logf_via_double(lhs) + lv;
        zs  = z * (double)sqrtf(invLhs);
        thr = log((1.0 - u) / u);
        beta_excl = (float)(z * (double)sqrtf(var_));       // :54
    }
    // RULE D (single-trait BayesA/B/C under a uniform prior pi = 0: RR-BLUP, BayesA, BayesL, the reference's own benchmark
    // setting): every marker is included whatever its rhs, so the scalar kernel's chain rhs -> gHat -> alpha (:36,:39,:46)
    // is one affine map of the block rhs x.  Its two coefficients depend on the marker, its old effect and its draw only,
    //     c1 = fl32( fl64(1/vare) * fl64(1/lhs) ),   c0 = fl32( fl64(1/vare) * fl64(1/lhs) * (fl64(d) * fl64(alpha_old)) + z sqrt(1/lhs) ),
    // and the new effect is  alpha = fmaf(c1, x, c0)  -- the same conditional mean and draw in one rounding instead of
    // four: the serial chain of a dense block is fma -> subtract -> broadcast -> fma per marker (~45 cycles instead of
    // ~100).  Part of the sampler's definition for such sweeps on every path; the oracle applies it too (orc abc_update).
    __device__ __forceinline__ void rule_d(float a_old, float ie, float& c1D, float& c0D) const
    {
        const double k1 = (double)ie * (double)invLhs;
        c1D = (float)k1;
        c0D = (float)(k1 * ((double)d * (double)a_old) + zs);
    }
    __device__ __forceinline__ void store(double* pd, float* pf, int64_t p, int64_t j) const
    {
        pd[0 * p + j] = lp0; pd[1 * p + j] = lp1; pd[2 * p + j] = thr; pd[3 * p + j] = zs;
        pf[0 * p + j] = invLhs; pf[1 * p + j] = c1; pf[2 * p + j] = beta_excl;
    }
    __device__ __forceinline__ void load(const double* pd, const float* pf, int64_t p, int64_t j, float d_)
    {
        d = d_;
        lp0 = pd[0 * p + j]; lp1 = pd[1 * p + j]; thr = pd[2 * p + j]; zs = pd[3 * p + j];
        invLhs = pf[0 * p + j]; c1 = pf[1 * p + j]; beta_excl = pf[2 * p + j];
    }
    __device__ __forceinline__ bool evaluate(float rhs_b, float a_old, float ie, float& gHat) const
    {
        const float rhs   = (rhs_b + d * a_old) * ie;                       // :36
        gHat              = rhs * invLhs;                                   // :39
        const float inner = c1 - gHat * rhs;                                // fp32 part of :40
        const double l1   = -0.5 * (double)inner + lp1;                     // :40
        return (lp0 - l1) < thr;                                            // :41,:44
    }
    __device__ __forceinline__ float alpha_incl(float gHat) const
    {
        return (float)((double)gHat + zs);                                  // :46
    }
    // INCLUSION THRESHOLDS.  For a fixed alpha_old the decision of evaluate() is a monotone function of |rhs|: every
    // operation between the block rhs x and the comparison is a correctly rounded product / sum with a fixed positive
    // factor (monotone, sign-symmetric), so  {x : included} = {x <= lo} U {x >= hi}  for two floats (lo, hi) that depend
    // on the marker and its draw only.  They are found once per sweep, for all markers in parallel, by bisection over
    // the float number line with evaluate() itself as the oracle (32 steps each); the serial chain then decides with
    // two float compares -- no double-precision arithmetic, no dependency on anything but x.
    // always included: lo = hi = -d*alpha_old;  never: lo = -inf, hi = +inf.
    __device__ __forceinline__ void thresholds(float a_old, float ie, float& lo, float& hi) const
    {
        float gh;
        const float x0 = -(d * a_old);                       // rhs = ((x + d*a_old)*ie) = 0 here: |rhs| minimal
        if (evaluate(x0, a_old, ie, gh)) { lo = x0; hi = x0; return; }
        if (!evaluate(kFltMax, a_old, ie, gh)) hi = INFINITY;
        else {
            uint32_t a = float_key(x0), b = float_key(kFltMax);          // a: excluded, b: included
            while (b - a > 1u) { const uint32_t m = a + ((b - a) >> 1); if (evaluate(key_float(m), a_old, ie, gh)) b = m; else a = m; }
            hi = key_float(b);
        }
        if (!evaluate(-kFltMax, a_old, ie, gh)) lo = -INFINITY;
        else {
            uint32_t a = float_key(-kFltMax), b = float_key(x0);         // a: included, b: excluded
            while (b - a > 1u) { const uint32_t m = a + ((b - a) >> 1); if (evaluate(key_float(m), a_old, ie, gh)) a = m; else b = m; }
            lo = key_float(a);
        }
    }
};

// The committed update of a BayesA/B/C marker from its thresholds: same values as AbcMarker::evaluate + alpha_incl.
__device__ __forceinline__ bool abc_included(float x, float lo, float hi) { return (x >= hi) || (x <= lo); }
__device__ __forceinline__ float abc_alpha_new(float x, float a_old, float d, float ie, float invLhs, double zs, bool incl)
{
    const float rhs  = (x + d * a_old) * ie;                                // :36
    const float gHat = rhs * invLhs;                                        // :39
    return incl ? (float)((double)gHat + zs) : 0.f;                         // :46 / :55
}

// BayesR (BayesR.jl:56-96)
struct BayesRMarker {
    float  d, die;
    double lpi[4], invLhs[4], cA[4], zs[4], u;          // zs[k] = z*sqrt(1/lhs_k): the normal draw scaled for class k (:93)
    // Class thresholds.  With s = rhs^2 every class log-weight is LINEAR in s (lp_k = 0.5*invLhs_k*s + const_k, slopes
    // increasing with the class variance), so the family has a monotone likelihood ratio in s and every cumulative
    // probability cum_k(s) = P(class <= k | s) DECREASES in s.  For the marker's uniform u the rule "class = number of
    // k in {0,1,2} with cum_k <= u" (:73-79) is therefore "number of k with s >= T[k]", T[k] the root of cum_k(s) = u.
    // The roots depend on the marker and its draw only, not on the running rhs: they are found once per sweep, for
    // all markers in parallel (k_prepare), and the serial chain decides a class with one multiply and three compares
    // instead of seven double-precision exp/log.  T = -1: boundary always passed; T = +inf: never.
    double T[3];
    __device__ __forceinline__ double cum(int kk, double s) const          // cum_kk(s), same formulas as evaluate()
    {
        double lp[4];
        lp[0] = lpi[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) lp[k] = 0.5 * (cA[k] + invLhs[k] * s) + lpi[k];
        double mx = lp[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) if (lp[k] > mx) mx = lp[k];
        double e[4], se = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { e[k] = exp(lp[k] - mx); se += e[k]; }
        double c = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) if (k <= kk) c += e[k];
        return c / se;
    }
    __device__ __forceinline__ void find_thresholds()
    {
#pragma unroll 1
        for (int kk = 0; kk < 3; ++kk) {
            double t;
            if (!(cum(kk, 0.0) > u)) t = -1.0;                              // already <= u at s = 0 (or NaN priors)
            else {
                double lo = 0.0, hi = 1e-4;
                int it = 0;
                while (cum(kk, hi) > u && it < 80) { lo = hi; hi *= 4.0; ++it; }
                if (it >= 80) t = INFINITY;                                 // never reached (the classes above have zero mass)
                else {
#pragma unroll 1
                    for (int q = 0; q < 56; ++q) {
                        const double mid = 0.5 * (lo + hi);
                        if (cum(kk, mid) > u) lo = mid; else hi = mid;
                    }
                    t = hi;
                }
            }
            // (no dynamic indexing: T must stay in registers for the serial chain)
            if (kk == 0) T[0] = t; else if (kk == 1) T[1] = t; else T[2] = t;
        }
    }
    __device__ __forceinline__ void prepare(float d_, float sigma_sq, const double* pi_j, const double* gamma,
                                            float ie, double u_, double z_)
    {
        d = d_; u = u_;
        die = d_ * ie;
        lpi[0] = log(pi_j[0]); invLhs[0] = 0.0; cA[0] = 0.0; zs[0] = 0.0;
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const double varE = gamma[k] * (double)sigma_sq;
            const double lhs  = (double)die + 1.0 / varE;
            invLhs[k] = 1.0 / lhs;
            cA[k]  = log(invLhs[k]) - log(varE);
            lpi[k] = log(pi_j[k]);
            zs[k]  = z_ * sqrt(invLhs[k]);
        }
        find_thresholds();
    }
    __device__ __forceinline__ void store(double* pd, float* pf, int64_t p, int64_t j) const
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) pd[k * p + j] = lpi[k];
#pragma unroll
        for (int k = 1; k < 4; ++k) { pd[(3 + k) * p + j] = invLhs[k]; pd[(6 + k) * p + j] = cA[k]; pd[(9 + k) * p + j] = zs[k]; }
        pd[13 * p + j] = u;
#pragma unroll
        for (int k = 0; k < 3; ++k) pd[(14 + k) * p + j] = T[k];
        (void)pf;
    }
    __device__ __forceinline__ void load(const double* pd, const float* pf, int64_t p, int64_t j, float d_, float ie)
    {
        d = d_; die = d_ * ie;
        invLhs[0] = 0.0; cA[0] = 0.0; zs[0] = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) lpi[k] = pd[k * p + j];
#pragma unroll
        for (int k = 1; k < 4; ++k) { invLhs[k] = pd[(3 + k) * p + j]; cA[k] = pd[(6 + k) * p + j]; zs[k] = pd[(9 + k) * p + j]; }
        u = pd[13 * p + j];
#pragma unroll
        for (int k = 0; k < 3; ++k) T[k] = pd[(14 + k) * p + j];
        (void)pf;
    }
    // The subset the serial wave needs (parked in LDS, kFastD doubles per marker): 1/lhs_k, z*sqrt(1/lhs_k), T_k.
    static constexpr int kFastD = 9;
    __device__ __forceinline__ void store_fast(double* pd, int64_t p, int64_t j) const
    {
#pragma unroll
        for (int k = 0; k < 3; ++k) { pd[k * p + j] = invLhs[k + 1]; pd[(3 + k) * p + j] = zs[k + 1]; pd[(6 + k) * p + j] = T[k]; }
    }
    __device__ __forceinline__ void load_fast(const double* pd, int64_t p, int64_t j, float d_, float ie)
    {
        d = d_; die = d_ * ie;
        invLhs[0] = 0.0; zs[0] = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { invLhs[k + 1] = pd[k * p + j]; zs[k + 1] = pd[(3 + k) * p + j]; T[k] = pd[(6 + k) * p + j]; }
    }
    // the same subset straight from k_prepare's global arrays (rows as written by store())
    __device__ __forceinline__ void load_fast_global(const double* pd, int64_t p, int64_t j, float d_, float ie)
    {
        d = d_; die = d_ * ie;
        invLhs[0] = 0.0; zs[0] = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { invLhs[k + 1] = pd[(4 + k) * p + j]; zs[k + 1] = pd[(10 + k) * p + j]; T[k] = pd[(14 + k) * p + j]; }
    }
    // returns class 0..3 and the candidate alpha for that class
    __device__ __forceinline__ int evaluate(float rhs_b, float a_old, float ie, float& a_new) const
    {
        const float rhs = (rhs_b + d * a_old) * ie;                         // :60
        double lp[4], bh[4];
        lp[0] = lpi[0]; bh[0] = 0.0;
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            bh[k] = invLhs[k] * (double)rhs;
            lp[k] = 0.5 * (cA[k] + bh[k] * (double)rhs) + lpi[k];           // :71
        }
        double mx = lp[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) if (lp[k] > mx) mx = lp[k];
        double se = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) se += exp(lp[k] - mx);
        const double log_norm = mx + log(se);
        int cls = 0;
        double cp = exp(lp[0] - log_norm);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const double pk = exp(lp[k] - log_norm);
            if (cls == k - 1 && cp <= u) { cls = k; cp += pk; }
        }
        double an = 0.0;
#pragma unroll
        for (int k = 1; k < 4; ++k) if (cls == k) an = bh[k] + zs[k];                  // :93
        a_new = (float)an;
        return cls;
    }
    // The same decision from the precomputed thresholds.  `sure` is false when s lies within a relative 1e-9 of a
    // threshold (the bisection resolves the root to ~1e-16; the exact formulas round at ~1e-15): the caller then
    // re-evaluates with evaluate().  The effect of the chosen class is computed exactly as in evaluate().
    __device__ __forceinline__ int evaluate_thr(float rhs_b, float a_old, float ie, float& a_new, bool& sure) const
    {
        const float rhs = (rhs_b + d * a_old) * ie;                         // :60
        const double rd = (double)rhs, s = rd * rd;
        const int cls = (s >= T[0] ? 1 : 0) + (s >= T[1] ? 1 : 0) + (s >= T[2] ? 1 : 0);
        // "close to a threshold" in single precision: |s - T| <= 1e-6*T is a superset of the 1e-9 band that matters
        // (T = -1: never close; T = +inf: inf - inf = NaN, the comparison is false)
        const float sf = (float)s;
        sure = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) { const float tf = (float)T[k]; sure = sure && !(fabsf(sf - tf) <= 1e-6f * tf); }
        const double il = cls == 1 ? invLhs[1] : (cls == 2 ? invLhs[2] : invLhs[3]);
        const double zz = cls == 1 ? zs[1] : (cls == 2 ? zs[2] : zs[3]);
        a_new = cls == 0 ? 0.f : (float)(il * rd + zz);                     // bh_k + z*sqrt(1/lhs_k)  :93
        return cls;
    }
};

// BayesR candidacy threshold for a marker that is OUT of the model (alpha_old = 0): the smallest |x| at which its class
// leaves 1.  The class is monotone in s = rhs^2 (see BayesRMarker) and rhs = x*ie is monotone in |x|, so bisection over
// the non-negative floats with the serial chain's own decision (thresholds in s, exact formulas next to one) finds it.
// +inf: never.  A marker that is in the model is always a candidate (its effect is redrawn or removed).
__device__ __forceinline__ float bayesr_candidate_threshold(const BayesRMarker& bm, float ie)
{
    auto leaves = [&](float x) {
        float an; bool sure;
        int c = bm.evaluate_thr(x, 0.f, ie, an, sure);
        if (!sure) c = bm.evaluate(x, 0.f, ie, an);
        return c != 0;
    };
    if (leaves(0.f)) return 0.f;
    if (!leaves(kFltMax)) return INFINITY;
    uint32_t a = float_key(0.f), b = float_key(kFltMax);
    while (b - a > 1u) { const uint32_t m = a + ((b - a) >> 1); if (leaves(key_float(m))) b = m; else a = m; }
    return key_float(b);
}

// The serial wave's evaluation of a BayesR marker from plain scalars passed BY VALUE (an object with the constants --
// even one with scalar members only -- is kept in scratch memory by hipcc once any sibling object is, and the
// class-dependent selects then become scratch loads with a full memory wait inside every round).
// Thresholds are ordered T0 <= T1 <= T2 (the classes nest); same arithmetic as BayesRMarker::evaluate_thr.
__device__ __forceinline__ int bayesr_eval_thr(float rhs_b, float a_old, float ie, float d,
                                               double il1, double il2, double il3, double zs1, double zs2, double zs3,
                                               double T0, double T1, double T2, float& a_new, bool& sure)
{
    const float rhs = (rhs_b + d * a_old) * ie;                         // :60
    const double rd = (double)rhs, s = rd * rd;
    const bool g1 = s >= T0, g2 = s >= T1, g3 = s >= T2;
    const int cls = (g1 ? 1 : 0) + (g2 ? 1 : 0) + (g3 ? 1 : 0);
    const float sf = (float)s, t0 = (float)T0, t1 = (float)T1, t2 = (float)T2;
    sure = !(fabsf(sf - t0) <= 1e-6f * t0) && !(fabsf(sf - t1) <= 1e-6f * t1) && !(fabsf(sf - t2) <= 1e-6f * t2);
    const double il = g3 ? il3 : (g2 ? il2 : il1);
    const double zz = g3 ? zs3 : (g2 ? zs2 : zs1);
    a_new = g1 ? (float)(il * rd + zz) : 0.f;                           // bh_k + z*sqrt(1/lhs_k)  :93
    return cls;
}

// t x t inverse on the device: the host's inv_small (double Gauss-Jordan with partial pivoting, rounded to float), the same
// operation sequence -- multi-trait BayesA/B inverts one covariance matrix per marker (Ginv = inv.(varEffects),
// MTBayesABC.jl:66).  A singular matrix yields NaNs.
template <int NT>
__device__ __forceinline__ void inv_small_dev(const float* A, float* Ainv)
{
    double M[NT][2 * NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) { M[i][j] = A[i * NT + j]; M[i][NT + j] = (i == j) ? 1.0 : 0.0; }
    bool bad = false;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        int piv = c;
        double best = fabs(M[c][c]);                                 // (= fabs(M[piv][c]): no indexing by a run-time row)
#pragma unroll
        for (int i = c + 1; i < NT; ++i) { const double v = fabs(M[i][c]); if (v > best) { best = v; piv = i; } }
#pragma unroll
        for (int i = c + 1; i < NT; ++i)
            if (i == piv) {
#pragma unroll
                for (int j = 0; j < 2 * NT; ++j) { const double tmp = M[c][j]; M[c][j] = M[i][j]; M[i][j] = tmp; }
            }
        if (M[c][c] == 0.0) bad = true;
        const double d = M[c][c];
#pragma unroll
        for (int j = 0; j < 2 * NT; ++j) M[c][j] /= d;
#pragma unroll
        for (int i = 0; i < NT; ++i) if (i != c) {
            const double f = M[i][c];
            if (f != 0.0) {
#pragma unroll
                for (int j = 0; j < 2 * NT; ++j) M[i][j] -= f * M[c][j];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) Ainv[i * NT + j] = bad ? NAN : (float)M[i][NT + j];
}

// K_P: per-sweep marker constants for repetition 0.  grid = ceil(p/256), block = 256.
template <int METHOD, int NT>
__global__ __launch_bounds__(256) void k_prepare(const DevParams* __restrict__ P, int64_t p,
                                                 const float* __restrict__ xpx, const float* __restrict__ alpha,
                                                 double* __restrict__ prep_d, float* __restrict__ prep_f)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= p) return;
    const RngKey key{P->seed_lo, P->seed_hi, P->iter, 0u};
    const uint32_t marker = P->marker0 + (uint32_t)j;
    if constexpr (is_mt_method(METHOD)) {
        float Gi[NT * NT];
        if constexpr (has_marker_cov(METHOD)) {                     // this marker's own covariance: invert it once per sweep
            float Gj[NT * NT];
#pragma unroll
            for (int i = 0; i < NT * NT; ++i) Gj[i] = P->var_mat[j * (NT * NT) + i];
            if constexpr (is_mega(METHOD)) {                        // constraint = true: the sampler reads the diagonal variances themselves
#pragma unroll
                for (int i = 0; i < NT * NT; ++i) Gi[i] = Gj[i];
            } else
            inv_small_dev<NT>(Gj, Gi);
#pragma unroll
            for (int i = 0; i < NT * NT; ++i) P->ginv_mat[j * (NT * NT) + i] = Gi[i];
        } else {
#pragma unroll
            for (int i = 0; i < NT * NT; ++i) Gi[i] = P->Ginv[i];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const double u = draw_uniform(key, marker, (uint32_t)t);
            prep_d[(int64_t)t * p + j] = is_sampler2(METHOD) ? u : log((1.0 - u) / u);   // sampler II keeps the raw uniform
            prep_d[(int64_t)(NT + t) * p + j] = draw_normal(key, marker, (uint32_t)t);
            // log of the marker's "in the model" left-hand side (depends on x'x and this sweep's variances only): the
            // double-precision log is taken here, for all markers in parallel, not inside the serial sampler
            const float dj = xpx[j];
            if constexpr (is_mega(METHOD)) {
                const float var = (METHOD == kMegaBayesB) ? Gi[t * NT + t] : P->var_effect[t * NT + t];
                const float lhs = dj * (1.0f / P->vare[t * NT + t]) + 1.0f / var;                     // BayesABC.jl:37
                prep_f[(int64_t)t * p + j] = logf_via_double(lhs) + logf_via_double(var);
            } else {
                prep_f[(int64_t)t * p + j] = logf_via_double(Gi[t * NT + t] + P->Rinv[t * NT + t] * dj);         // MTBayesABC.jl:89
            }
        }
    } else {
        const float ie = 1.0f / P->vare[0];
        const double u = draw_uniform(key, marker, 0u);
        const double z = draw_normal(key, marker, 0u);
        if constexpr (METHOD == kBayesR) {
            double pj[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) pj[k] = P->pi_mat ? P->pi_mat[4 * j + k] : P->pi4[k];
            BayesRMarker bm;
            bm.prepare(xpx[j], P->var_effect[0], pj, P->gamma, ie, u, z);
            bm.store(prep_d, prep_f, p, j);
            prep_f[j] = (alpha[j] != 0.f) ? 0.f : bayesr_candidate_threshold(bm, ie);     // |x| >= this: a candidate
        } else {
            float var_j = P->var_effect[0];
            if constexpr (METHOD == kBayesB) var_j = P->var_vec[j];
            double pi_j = P->pi;
            if (P->pi_vec) pi_j = P->pi_vec[j];
            AbcMarker am;
            am.prepare(xpx[j], var_j, pi_j, ie, u, z);
            am.store(prep_d, prep_f, p, j);
            float lo, hi;
            am.thresholds(alpha[j], ie, lo, hi);             // alpha at the start of the sweep = alpha_old of repetition 0
            prep_f[3 * p + j] = lo; prep_f[4 * p + j] = hi;
            if (P->pi == 0.0 && P->pi_vec == nullptr) {      // Rule D (uniform pi = 0)
                float c1D, c0D;
                am.rule_d(alpha[j], ie, c1D, c0D);
                prep_f[5 * p + j] = c1D; prep_f[6 * p + j] = c0D;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// post-sweep reductions over markers (K11).  out[blockIdx.x][kNStat]
// ---------------------------------------------------------------------------------------------
constexpr int kNCounters = 32;            // diagnostics counters of a sweep (events, phase cycles, compact-chain blocks)
constexpr int kNStat = 4 + 16 + 16 + 4 + 2 + 16;   // sum_delta[4] alpha_ss[16] beta_ss[16] class[4] ssq,nnz state[16]

template <int NT>
__global__ __launch_bounds__(256) void k_marker_stats(int method, int64_t p, const float* __restrict__ alpha,
                                                      const float* __restrict__ beta, const void* __restrict__ delta_v,
                                                      const double* __restrict__ gamma_dev /*P->gamma*/,
                                                      double* __restrict__ out)
{
    __shared__ double red[4 * kNStat];
    double v[kNStat];
#pragma unroll
    for (int i = 0; i < kNStat; ++i) v[i] = 0.0;
    const float* delta_f = reinterpret_cast<const float*>(delta_v);
    const int32_t* delta_i = reinterpret_cast<const int32_t*>(delta_v);
    const int64_t chunk = (p + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = (lo + chunk < p) ? lo + chunk : p;
    for (int64_t j = lo + threadIdx.x; j < hi; j += 256) {
        if (method == kBayesR) {
            const int cls = delta_i[j];
            const double a = alpha[j];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (cls == k + 1) v[36 + k] += 1.0;
            if (cls > 1) { v[40] += (a * a) / gamma_dev[cls - 1]; v[41] += 1.0; }
            v[4] += a * a;
        } else {
            unsigned st = 0u;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const double dl = delta_f[(int64_t)t * p + j];
                v[t] += dl;
                if (dl != 0.0) st |= 1u << t;
            }
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int c = 0; c < NT; ++c) {
                    v[4 + a * NT + c]  += (double)alpha[(int64_t)a * p + j] * (double)alpha[(int64_t)c * p + j];
                    v[20 + a * NT + c] += (double)beta[(int64_t)a * p + j] * (double)beta[(int64_t)c * p + j];
                }
#pragma unroll
            for (int q = 0; q < (1 << NT); ++q) if (st == (unsigned)q) v[42 + q] += 1.0;
        }
    }
    block_sum<kNStat>(v, red, 4);
    if (threadIdx.x == 0)
#pragma unroll
        for (int i = 0; i < kNStat; ++i) out[(int64_t)blockIdx.x * kNStat + i] = v[i];
}

// running posterior means (output.jl:568-577)
JW_PLAIN_KERNEL __global__ __launch_bounds__(256) void k_accumulate(int64_t count, int delta_is_class, double k,
                                                    const float* __restrict__ alpha, const void* __restrict__ delta_v,
                                                    float* __restrict__ mean_a, float* __restrict__ mean_a2,
                                                    float* __restrict__ mean_d)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= count) return;
    const float a = alpha[j];
    mean_a[j]  = (float)((double)mean_a[j] + ((double)(a - mean_a[j])) / k);
    const float a2 = a * a;
    mean_a2[j] = (float)((double)mean_a2[j] + ((double)(a2 - mean_a2[j])) / k);
    const float ind = delta_is_class ? (reinterpret_cast<const int32_t*>(delta_v)[j] > 1 ? 1.f : 0.f)
                                     : reinterpret_cast<const float*>(delta_v)[j];
    mean_d[j]  = (float)((double)mean_d[j] + ((double)(ind - mean_d[j])) / k);
}

// ---------------------------------------------------------------------------------------------
// storage-layer precompute
// ---------------------------------------------------------------------------------------------
// x'x per column, fp64 accumulated (getXpRinvX, tools4genotypes.jl:33-36).  grid = p, block 256.
template <class CX>
__global__ __launch_bounds__(256) void k_xpx(CX cx, float* __restrict__ xpx)
{
    __shared__ double red[4];
    const int64_t ld = cx.ld;
    double v[1] = {0.0};
    for (int64_t i = (int64_t)threadIdx.x * 4; i < ld; i += 1024) {
        const float4 q = cx.load4(blockIdx.x, i);
        const float4 wv = *reinterpret_cast<const float4*>(cx.w + i);          // x'R^-1 x (getXpRinvX, tools4genotypes.jl:28-31)
        v[0] = fma((double)q.x, (double)(q.x * wv.x), v[0]);
        v[0] = fma((double)q.y, (double)(q.y * wv.y), v[0]);
        v[0] = fma((double)q.z, (double)(q.z * wv.z), v[0]);
        v[0] = fma((double)q.w, (double)(q.w * wv.w), v[0]);
    }
    block_sum<1>(v, red, 4);
    if (threadIdx.x == 0) xpx[blockIdx.x] = (float)v[0];
}

// Exact (fp64-accumulated) Gram of one block: grid = (b, nblocks), block = 256; workgroup (a, blk)
// writes row a: G[a][c], c <= a, and mirrors.  Test-size path; O(p*b*n/2) VALU work.
template <class CX>
__global__ __launch_bounds__(256) void k_gram_f64(CX cx, int64_t p, int bsize,
                                                  float* __restrict__ gram, const int64_t* __restrict__ starts = nullptr)
{
    // starts (nblocks + 1 entries, or NULL): explicit, possibly non-uniform block starts (fast_blocks = a vector of starts,
    // JWAS.jl:298-304); blocks stay stored at stride bsize^2, bsize >= every block
    const int64_t ld = cx.ld;
    const int64_t blk = blockIdx.y;
    const int64_t j0 = starts ? starts[blk] : blk * bsize;
    const int b = starts ? (int)(starts[blk + 1] - j0) : (int)((j0 + bsize <= p) ? bsize : (p - j0));
    const int a = blockIdx.x;
    if (a >= b) return;
    float* G = gram + blk * (int64_t)bsize * bsize;   // blocks are stored at stride bsize^2, each b x b packed
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c <= a; c += 4) {
        double s = 0.0;
        for (int64_t i = (int64_t)lane * 4; i < ld; i += 256) {
            const float4 qa = cx.load4(j0 + a, i);
            const float4 qc = cx.load4(j0 + c, i);
            const float4 wv = *reinterpret_cast<const float4*>(cx.w + i);      // X_b' R^-1 X_b (tools4genotypes.jl:263-266)
            s = fma((double)qa.x, (double)(qc.x * wv.x), s);
            s = fma((double)qa.y, (double)(qc.y * wv.y), s);
            s = fma((double)qa.z, (double)(qc.z * wv.z), s);
            s = fma((double)qa.w, (double)(qc.w * wv.w), s);
        }
        s = wave_sum(s);
        if (lane == 0) { G[(int64_t)a * b + c] = (float)s; G[(int64_t)c * b + a] = (float)s; }
    }
}

// fp32 MFMA Gram: one workgroup (4 waves) per 64x64 output tile of one block, K = ld rows.
// v_mfma_f32_32x32x2_f32; K is consumed in 8-row groups with lane (m, h) holding rows 8q+4h..+3 of
// its column (ds_read_b128 from a [64][36]-float LDS image, conflict-free for 16-lane groups);
// A and B use the same K permutation so the contraction is unchanged.  Accumulators are folded
// into fp64 every kGramChunk rows to bound fp32 accumulation error.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kGramKT = 32;          // rows per LDS tile
constexpr int kGramLd = 36;          // LDS row stride (floats)
constexpr int kGramChunk = 64;       // rows per fp32 accumulation chunk (then folded into fp64)

// cross != 0: the cross-Gram of consecutive blocks C = X_{blk-1}' X_blk for blk = blockIdx.y + 1 (all
// nt x nt tiles, rows = markers of the previous block, row stride = size of block blk).
template <class CX>
__global__ __launch_bounds__(256) void k_gram_mfma(CX cx, int64_t p, int bsize,
                                                   float* __restrict__ gram, int cross, const int64_t* __restrict__ starts = nullptr)
{
    const int64_t ld = cx.ld;
    __shared__ __attribute__((aligned(16))) float As[64 * kGramLd];
    __shared__ __attribute__((aligned(16))) float Bs[64 * kGramLd];
    const int64_t blk = cross == 2 ? 2 * (int64_t)blockIdx.y + 1 : (cross ? (int64_t)blockIdx.y + 1 : (int64_t)blockIdx.y);   // (cross = 2: the odd blocks only)
    const int64_t j0 = starts ? starts[blk] : blk * bsize;                        // block of the B operand (columns of the output)
    const int b = starts ? (int)(starts[blk + 1] - j0) : (int)((j0 + bsize <= p) ? bsize : (p - j0));
    const int64_t jA = cross ? (starts ? starts[blk - 1] : j0 - bsize) : j0;      // block of the A operand (rows of the output)
    const int bA = cross ? (int)(j0 - jA) : b;
    int ti = 0, tj = 0;
    if (cross) { const int nt = bsize / 64; ti = blockIdx.x / nt; tj = blockIdx.x % nt; }
    else {   // tile index -> (ti, tj), ti >= tj, over nt = ceil(bsize/64) tiles per side
        int rem = blockIdx.x;
        while (rem > ti) { rem -= ti + 1; ++ti; }
        tj = rem;
    }
    if (ti * 64 >= bA || tj * 64 >= b) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;              // wave's 32x32 sub-tile
    const bool same = !cross && (ti == tj);
    const bool diag = same && !cx.weighted;           // unweighted diagonal tile: B operand = A operand
    float* G = gram + blk * (int64_t)bsize * bsize;

    // staging: thread -> (marker m = tid/8 [+32], float4 q = tid%8)
    const int sm = tid >> 3, sq = tid & 7;
    const int ma0 = ti * 64 + sm, ma1 = ma0 + 32, mb0 = tj * 64 + sm, mb1 = mb0 + 32;
    const int64_t ja0 = jA + (ma0 < bA ? ma0 : 0), ja1 = jA + (ma1 < bA ? ma1 : 0);
    const int64_t jb0 = j0 + (mb0 < b ? mb0 : 0), jb1 = j0 + (mb1 < b ? mb1 : 0);
    const float4 zero4{0.f, 0.f, 0.f, 0.f};

    f32x16 acc;
    double accd[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; accd[i] = 0.0; }

    const int am = lane & 31, ah = lane >> 5;
    const float* a_rd = As + (wm * 32 + am) * kGramLd + 4 * ah;
    const float* b_rd = (diag ? As : Bs) + (wn * 32 + am) * kGramLd + 4 * ah;

    int chunk_rows = 0;
    for (int64_t k0 = 0; k0 < ld; k0 += kGramKT) {
        // unconditional loads from clamped addresses; out-of-block markers are zeroed by value
        float4 va0 = cx.load4(ja0, k0 + sq * 4);
        float4 va1 = cx.load4(ja1, k0 + sq * 4);
        float4 vb0 = zero4, vb1 = zero4;
        if (!diag) {
            vb0 = cx.load4(jb0, k0 + sq * 4);
            vb1 = cx.load4(jb1, k0 + sq * 4);
        }
        if (ma0 >= bA) va0 = zero4;
        if (ma1 >= bA) va1 = zero4;
        if (mb0 >= b) vb0 = zero4;
        if (mb1 >= b) vb1 = zero4;
        if (!diag) {                                           // the B operand carries R^-1
            const float4 wv = *reinterpret_cast<const float4*>(cx.w + k0 + sq * 4);
            vb0.x *= wv.x; vb0.y *= wv.y; vb0.z *= wv.z; vb0.w *= wv.w;
            vb1.x *= wv.x; vb1.y *= wv.y; vb1.z *= wv.z; vb1.w *= wv.w;
        }
        __syncthreads();      // previous tile fully consumed
        *reinterpret_cast<float4*>(&As[sm * kGramLd + sq * 4]) = va0;
        *reinterpret_cast<float4*>(&As[(sm + 32) * kGramLd + sq * 4]) = va1;
        if (!diag) {
            *reinterpret_cast<float4*>(&Bs[sm * kGramLd + sq * 4]) = vb0;
            *reinterpret_cast<float4*>(&Bs[(sm + 32) * kGramLd + sq * 4]) = vb1;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kGramKT / 8; ++q) {
            const float4 fa = *reinterpret_cast<const float4*>(a_rd + 8 * q);
            const float4 fb = *reinterpret_cast<const float4*>(b_rd + 8 * q);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc, 0, 0, 0);
        }
        chunk_rows += kGramKT;
        if (chunk_rows >= kGramChunk) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { accd[i] += (double)acc[i]; acc[i] = 0.f; }
            chunk_rows = 0;
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) accd[i] += (double)acc[i];

    // C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    // here "row" indexes the A operand (markers of tile ti), "col" the B operand (tile tj).
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int rr = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        const int ga = ti * 64 + wm * 32 + rr;
        const int gc = tj * 64 + wn * 32 + (lane & 31);
        if (ga < bA && gc < b) {
            const float v = (float)accd[i];
            G[(int64_t)ga * b + gc] = v;
            if (!same && !cross) G[(int64_t)gc * b + ga] = v;
        }
    }
}

// The cross-Grams of grouped launches (sweep.hpp k_group_step: bsize = 2 or 4 marker blocks, 2048 / 4096 markers): the same
// contraction as k_gram_mfma's cross mode -- same K order, same fp32 chunks folded into fp64, so the same numbers -- on 128 x 128
// output tiles: each wave owns a 64 x 64 quarter (2 x 2 MFMA tiles), every LDS fragment feeds two MFMAs and every global load
// half as many tiles' worth of re-reads (a 4096-marker group is 1024 tiles instead of 4096).  cross = 1: blocks 1, 2, ...;
// cross = 2: the odd blocks only.  Uniform partitions, bsize a multiple of 128.
template <class CX>
__global__ __launch_bounds__(256) void k_cross_mfma128(CX cx, int64_t p, int bsize, float* __restrict__ gram, int cross)
{
    const int64_t ld = cx.ld;
    __shared__ __attribute__((aligned(16))) float As[128 * kGramLd];
    __shared__ __attribute__((aligned(16))) float Bs[128 * kGramLd];
    const int64_t blk = cross == 2 ? 2 * (int64_t)blockIdx.y + 1 : (int64_t)blockIdx.y + 1;
    const int64_t j0 = blk * bsize;                                  // block of the B operand (columns of the output)
    const int b = (int)((j0 + bsize <= p) ? bsize : (p - j0));
    const int64_t jA = j0 - bsize;                                   // block of the A operand (rows of the output): always full
    const int bA = bsize;
    const int nt = bsize / 128;
    const int ti = blockIdx.x / nt, tj = blockIdx.x % nt;
    if (tj * 128 >= b) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                         // wave's 64 x 64 quarter
    float* G = gram + (cross == 2 ? (blk >> 1) : blk) * (int64_t)bsize * bsize;      // (odd blocks only: stored compactly, block 2q + 1 at q)

    // staging: thread -> (marker m = tid/8 + 32 u, float4 q = tid%8), u = 0..3
    const int sm = tid >> 3, sq = tid & 7;
    int64_t ja[4], jb[4];
    bool okb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        ja[u] = jA + ti * 128 + sm + 32 * u;
        const int mb = tj * 128 + sm + 32 * u;
        okb[u] = mb < b;
        jb[u] = j0 + (okb[u] ? mb : 0);
    }
    f32x16 acc[2][2];
    double accd[2][2][16];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[x][y][i] = 0.f; accd[x][y][i] = 0.0; }
    const int am = lane & 31, ah = lane >> 5;
    const float* a_rd = As + (wm * 64 + am) * kGramLd + 4 * ah;
    const float* b_rd = Bs + (wn * 64 + am) * kGramLd + 4 * ah;
    const float4 zero4{0.f, 0.f, 0.f, 0.f};
    int chunk_rows = 0;
    for (int64_t k0 = 0; k0 < ld; k0 += kGramKT) {
        float4 va[4], vb[4];
        const float4 wv = *reinterpret_cast<const float4*>(cx.w + k0 + sq * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) va[u] = cx.load4(ja[u], k0 + sq * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            vb[u] = cx.load4(jb[u], k0 + sq * 4);
            if (!okb[u]) vb[u] = zero4;
            vb[u].x *= wv.x; vb[u].y *= wv.y; vb[u].z *= wv.z; vb[u].w *= wv.w;       // the B operand carries R^-1
        }
        __syncthreads();      // previous tile fully consumed
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            *reinterpret_cast<float4*>(&As[(sm + 32 * u) * kGramLd + sq * 4]) = va[u];
            *reinterpret_cast<float4*>(&Bs[(sm + 32 * u) * kGramLd + sq * 4]) = vb[u];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kGramKT / 8; ++q) {
            const float4 fa0 = *reinterpret_cast<const float4*>(a_rd + 8 * q);
            const float4 fa1 = *reinterpret_cast<const float4*>(a_rd + 32 * kGramLd + 8 * q);
            const float4 fb0 = *reinterpret_cast<const float4*>(b_rd + 8 * q);
            const float4 fb1 = *reinterpret_cast<const float4*>(b_rd + 32 * kGramLd + 8 * q);
#define JW_MFMA4(A_, B_, ACC_)                                                      \
            ACC_ = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.x, B_.x, ACC_, 0, 0, 0); \
            ACC_ = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.y, B_.y, ACC_, 0, 0, 0); \
            ACC_ = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.z, B_.z, ACC_, 0, 0, 0); \
            ACC_ = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.w, B_.w, ACC_, 0, 0, 0);
            JW_MFMA4(fa0, fb0, acc[0][0])
            JW_MFMA4(fa0, fb1, acc[0][1])
            JW_MFMA4(fa1, fb0, acc[1][0])
            JW_MFMA4(fa1, fb1, acc[1][1])
#undef JW_MFMA4
        }
        chunk_rows += kGramKT;
        if (chunk_rows >= kGramChunk) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int i = 0; i < 16; ++i) { accd[x][y][i] += (double)acc[x][y][i]; acc[x][y][i] = 0.f; }
            chunk_rows = 0;
        }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const double v = accd[x][y][i] + (double)acc[x][y][i];
                const int rr = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                const int ga = ti * 128 + wm * 64 + 32 * x + rr;
                const int gc = tj * 128 + wn * 64 + 32 * y + (lane & 31);
                if (ga < bA && gc < b) G[(int64_t)ga * b + gc] = (float)v;
            }
}

// ---------------------------------------------------------------------------------------------
// X * alpha (EBV, output.jl:302) and r -= X*alpha0 (MCMC_BayesianAlphabet.jl:142)
// grid = nslices, block = 256: one row per thread, columns in marker order.
// ---------------------------------------------------------------------------------------------
template <class CX>
__global__ __launch_bounds__(256) void k_mul_alpha(CX cx, int64_t p,
                                                   const float* __restrict__ alpha, float* __restrict__ out)
{
    const int64_t row = (int64_t)blockIdx.x * kSliceRows + threadIdx.x;
    double s = 0.0;
    // dense effects (the sparse case goes through k_mul_alpha_list): 16 independent column loads in flight per thread;
    // a zero effect contributes fma(0, x, s) = s exactly, so nothing is skipped and nothing branches
    constexpr int kUn = 16;
    int64_t j = 0;
    for (; j + kUn <= p; j += kUn) {
        float x[kUn];
#pragma unroll
        for (int u = 0; u < kUn; ++u) x[u] = cx.load1(j + u, row);
#pragma unroll
        for (int u = 0; u < kUn; ++u) s = fma((double)alpha[j + u], (double)x[u], s);
    }
    for (; j < p; ++j) s = fma((double)alpha[j], (double)cx.load1(j, row), s);
    out[row] = (float)s;
}

// ---------------------------------------------------------------------------------------------
// Window genomic values of one marker-effect sample (GWAS.jl:152-165): window w holds the nonzero effects
// idx[wptr[w] .. wptr[w+1]) with values val[..]; BV_w[i] = sum_j X[i, idx_j] * val_j (fp64-accumulated, fixed order).
// grid = nslices, block = 256 (one individual per thread); partial[(w * nslices + slice) * 2 + {0,1}] = the slice's
// sum of BV_w and of BV_w^2.  k_window_reduce then adds the slices in fixed order (deterministic, no atomics).
// ---------------------------------------------------------------------------------------------
// NV = 2: (sum, sum of squares) of one effect vector;  NV = 5: two effect vectors over the same markers (two traits'
// samples): (sum1, ss1, sum2, ss2, sum of products) -- the window genetic covariance / correlation of GWAS.jl:199-217.
template <class CX, int NV>
__global__ __launch_bounds__(256) void k_window_partial(CX cx, int nwin, const int32_t* __restrict__ wptr,
                                                        const int32_t* __restrict__ idx, const float* __restrict__ val,
                                                        const float* __restrict__ val2, double* __restrict__ partial)
{
    __shared__ double red[NV][4];
    const int64_t row = (int64_t)blockIdx.x * kSliceRows + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int w = 0; w < nwin; ++w) {
        double bv = 0.0, bv2 = 0.0;
        for (int e = wptr[w]; e < wptr[w + 1]; ++e) {
            const double x = (double)cx.load1(idx[e], row);
            bv = fma((double)val[e], x, bv);
            if constexpr (NV == 5) bv2 = fma((double)val2[e], x, bv2);
        }
        double v[NV];
        v[0] = bv; v[1] = bv * bv;
        if constexpr (NV == 5) { v[2] = bv2; v[3] = bv2 * bv2; v[4] = bv * bv2; }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
            if (lane == 0) red[k][wave] = v[k];
        }
        __syncthreads();
        if (threadIdx.x < NV)
            partial[((int64_t)w * gridDim.x + blockIdx.x) * NV + threadIdx.x] =
                ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3];
        __syncthreads();
    }
}
// out[v * nwin + w] = sum over the slices (fixed order) of value v of window w
JW_PLAIN_KERNEL __global__ __launch_bounds__(256) void k_window_reduce(int nwin, int nslices, int nv, const double* __restrict__ partial,
                                                       double* __restrict__ out)
{
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= nwin) return;
    for (int v = 0; v < nv; ++v) {
        double s = 0.0;
        for (int k = 0; k < nslices; ++k) s += partial[((int64_t)w * nslices + k) * nv + v];
        out[(int64_t)v * nwin + w] = s;
    }
}

// X * alpha over the nonzero effects only (marker order, the same fp64 accumulation as k_mul_alpha -> identical
// results): with a sparse prior a saved sample has a few hundred nonzero effects among 600 000, and the loop over all
// markers (one scalar load + branch each) costs 39 ms where the 600 useful columns cost 30 us.
template <class CX>
__global__ __launch_bounds__(256) void k_mul_alpha_list(CX cx, int nnz, const int32_t* __restrict__ idx,
                                                        const float* __restrict__ val, float* __restrict__ out)
{
    const int64_t row = (int64_t)blockIdx.x * kSliceRows + threadIdx.x;
    double s = 0.0;
    for (int e = 0; e < nnz; ++e) s = fma((double)val[e], (double)cx.load1(idx[e], row), s);
    out[row] = (float)s;
}

// The nonzero effects of one trait as (index, value) lists in marker order.  grid = 1, block = 1024: the workgroup walks
// the p effects 1024 at a time with a running offset (ballot + wave prefix), ~30 us at p = 600 000.
JW_PLAIN_KERNEL __global__ __launch_bounds__(1024) void k_compact_alpha(int64_t p, const float* __restrict__ alpha, int32_t* __restrict__ idx,
                                                        float* __restrict__ val, int32_t* __restrict__ count)
{
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int64_t j0 = 0; j0 < p; j0 += 1024) {
        const int64_t j = j0 + tid;
        const float a = j < p ? alpha[j] : 0.f;
        const bool nz = a != 0.f;
        const unsigned long long m = __ballot(nz);
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int pre = base_s, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) pre += wsum[w]; tot += wsum[w]; }
        if (nz) {
            const int pos = pre + __popcll(m & ((1ull << lane) - 1ull));
            idx[pos] = (int32_t)j; val[pos] = a;
        }
        __syncthreads();
        if (tid == 0) base_s += tot;
        __syncthreads();
    }
    if (tid == 0) *count = base_s;
}

template <class CX>
__global__ __launch_bounds__(256) void k_sub_xalpha(CX cx, int64_t p,
                                                    const float* __restrict__ alpha, float* __restrict__ r)
{
    const int64_t row = (int64_t)blockIdx.x * kSliceRows + threadIdx.x;
    float rv = r[row];
    for (int64_t j = 0; j < p; ++j) {
        const float a = alpha[j];
        if (a != 0.f) rv = fmaf(-a, cx.load1(j, row), rv);
    }
    r[row] = rv;
}

// ---------------------------------------------------------------------------------------------
// synthetic genotypes (bench / tests).  grid = p, block = 256.
// ---------------------------------------------------------------------------------------------
// kind 2 = single-step shaped input (the dense real-valued matrix impute_genotypes hands to the sweep, SSBR.jl:83-142):
// rows < n_int are 0/1/2 genotypes, rows >= n_int are "imputed": the average of two genotyped rows (a, b) drawn per
// ROW (the same linear map for every marker, as A_ng A_gg^-1 M_g is).
JW_PLAIN_KERNEL __global__ __launch_bounds__(256) void k_synth(float* X, int64_t n, int64_t ld, uint32_t seed_lo,
                                               uint32_t seed_hi, int kind, int center, uint32_t marker0, int64_t n_int = 0)
{
    __shared__ double red[4];
    const uint32_t j = marker0 + blockIdx.x;          // global marker index keys the generator
    float* x = X + (int64_t)blockIdx.x * ld;
    const u32x4 wf = philox4x32_10(j, 0xFFFFFFFFu, 0u, 0u, seed_lo, seed_hi);
    const float f = 0.1f + 0.3f * ((float)(wf.x >> 8) * 0x1.0p-24f);          // U(0.1, 0.4)
    double v[1] = {0.0};
    const int64_t n_code = (kind == 2) ? n_int : n;                             // rows holding generated values
    for (int64_t i = threadIdx.x; i < ld; i += 256) {
        float val = 0.f;
        if (i < n_code) {
            const u32x4 w = philox4x32_10(j, (uint32_t)i, 1u, 0u, seed_lo, seed_hi);
            if (kind == 1) val = (float)(w.x >> 8) * 0x1.0p-24f;                // U[0,1)
            else {
                const float u1 = (float)(w.x >> 8) * 0x1.0p-24f, u2 = (float)(w.y >> 8) * 0x1.0p-24f;
                val = (u1 < f ? 1.f : 0.f) + (u2 < f ? 1.f : 0.f);
            }
            v[0] += (double)val;
        }
        x[i] = val;
    }
    if (kind == 2) {
        __syncthreads();                                                        // the genotyped rows of this column are written
        for (int64_t i = n_int + threadIdx.x; i < n; i += 256) {
            const u32x4 w = philox4x32_10(0xFFFFFFFEu, (uint32_t)i, 2u, 0u, seed_lo, seed_hi);   // keyed by the row only
            const int64_t a = (int64_t)(((uint64_t)w.x * (uint64_t)n_int) >> 32), b = (int64_t)(((uint64_t)w.y * (uint64_t)n_int) >> 32);
            const float val = 0.5f * (x[a] + x[b]);
            v[0] += (double)val;
            x[i] = val;
        }
    }
    if (!center) return;
    block_sum<1>(v, red, 4);
    const float mean = (float)(v[0] / (double)n);
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += 256) x[i] = x[i] - mean;
}

// Same generator, written as the reference's 2-bit codes + per-marker mean (kind 0 only: 0/1/2 genotypes).
// Q: [p][sb] bytes, sb = ld/4.  The decoded matrix equals k_synth's output bit for bit.
JW_PLAIN_KERNEL __global__ __launch_bounds__(256) void k_synth_packed(uint8_t* __restrict__ Q, float* __restrict__ mean, int64_t n, int64_t ld,
                                                      uint32_t seed_lo, uint32_t seed_hi, uint32_t marker0)
{
    __shared__ double red[4];
    const uint32_t j = marker0 + blockIdx.x;
    uint8_t* q = Q + (int64_t)blockIdx.x * (ld >> 2);
    const u32x4 wf = philox4x32_10(j, 0xFFFFFFFFu, 0u, 0u, seed_lo, seed_hi);
    const float f = 0.1f + 0.3f * ((float)(wf.x >> 8) * 0x1.0p-24f);
    double v[1] = {0.0};
    for (int64_t i4 = threadIdx.x; i4 < (ld >> 2); i4 += 256) {
        unsigned byte = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = i4 * 4 + k;
            if (i < n) {
                const u32x4 w = philox4x32_10(j, (uint32_t)i, 1u, 0u, seed_lo, seed_hi);
                const float u1 = (float)(w.x >> 8) * 0x1.0p-24f, u2 = (float)(w.y >> 8) * 0x1.0p-24f;
                const unsigned code = (u1 < f ? 1u : 0u) + (u2 < f ? 1u : 0u);
                byte |= code << (2 * k);
                v[0] += (double)code;
            }
        }
        q[i4] = (uint8_t)byte;
    }
    block_sum<1>(v, red, 4);
    if (threadIdx.x == 0) mean[blockIdx.x] = (float)(v[0] / (double)n);
}

// Decode columns [j0, j0+count) into a dense column-major n x count host-visible buffer (ld_out = n).
template <class CX>
__global__ __launch_bounds__(256) void k_get_columns(CX cx, int64_t j0, int64_t n, float* __restrict__ out)
{
    const int64_t j = j0 + blockIdx.x;
    for (int64_t i = threadIdx.x; i < n; i += 256) out[(int64_t)blockIdx.x * n + i] = cx.load1(j, i);
}

// empty kernel: calibrates what a HIP-event pair around ONE launch measures beyond the kernel itself
JW_PLAIN_KERNEL __global__ void k_null() {}

}  // namespace jw
