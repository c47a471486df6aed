# JWASHip.jl -- ccall binding of libjwas_hip.so (include/jwas_hip.h) for reworkhow/JWAS.jl.
#
# Drop this file next to src/1.JWAS/src/markers/streaming_genotypes.jl and `include` it after that file; the call
# sites to patch are listed in INTEGRATION.md section 2.  Julia is not installed in the image this repository is built
# in, so the file has not been executed there.  What IS checked mechanically (tests/test_abi.py::test_julia_struct_layout):
# the field order, field types and NTuple lengths of HipSweepParams / HipSweepStats below are parsed and their C layout
# (natural alignment, which is what an isbits Julia struct uses) is compared with gcc's offsetof / sizeof of
# jwas_sweep_params / jwas_sweep_stats.
module JWASHip

export HipBackend, HipSweepParams, HipSweepStats, hip_sweep!, hip_sweep_sharded!, hip_comm_unique_id, hip_comm_init!, hip_comm_info, hip_residual_add_scalar!,
       hip_setup_blocks!, hip_set_residual!, hip_get_residual!, hip_accumulate!, hip_posterior, hip_mul_alpha, JWAS_HIP_BAYESC,
       JWAS_HIP_BAYESB, JWAS_HIP_BAYESR, JWAS_HIP_MTBAYESC1, JWAS_HIP_MTBAYESC2, JWAS_HIP_MEGABAYESC, JWAS_HIP_MTBAYESB1,
       JWAS_HIP_MTBAYESB2, JWAS_HIP_MEGABAYESB

const LIBJWAS_HIP = get(ENV, "JWAS_HIP_LIB", "libjwas_hip.so")

const JWAS_HIP_BAYESC, JWAS_HIP_BAYESB, JWAS_HIP_BAYESR = Int32(0), Int32(1), Int32(2)
const JWAS_HIP_MTBAYESC1, JWAS_HIP_MTBAYESC2, JWAS_HIP_MEGABAYESC, JWAS_HIP_MTBAYESB1 = Int32(3), Int32(4), Int32(5), Int32(6)
const JWAS_HIP_MTBAYESB2, JWAS_HIP_MEGABAYESB = Int32(7), Int32(8)     # BayesA/B under sampler II / constraint = true
const JWAS_HIP_GRAM_F64, JWAS_HIP_GRAM_MFMA = Int32(0), Int32(1)

# mirrors of struct jwas_sweep_params / jwas_sweep_stats (isbits, same field order; JWAS_HIP_MAX_TRAITS = 4)
struct HipSweepParams
    method::Int32
    ntraits::Int32
    nreps::Int32
    iteration::UInt32
    seed::UInt64
    marker_offset::UInt32
    independent_blocks::UInt32
    vare::NTuple{16,Float32}
    var_effect::NTuple{16,Float32}
    pi::Float64
    pi_classes::NTuple{4,Float64}
    gamma::NTuple{4,Float64}
    log_prior_states::NTuple{16,Float64}
    var_effect_vec::Ptr{Float32}
    pi_vec::Ptr{Float64}
    pi_matrix::Ptr{Float64}
    log_prior_states_matrix::Ptr{Float64}
    var_effect_matrix::Ptr{Float32}
    vare_f64::NTuple{16,Float64}
    var_effect_f64::NTuple{16,Float64}
    var_effect_vec_f64::Ptr{Float64}
    section_solve::Int32
    group_launch::Int32
end

struct HipSweepStats
    sum_delta::NTuple{4,Float64}
    alpha_ss::NTuple{16,Float64}
    beta_ss::NTuple{16,Float64}
    resid_ss::NTuple{16,Float64}
    resid_sum::NTuple{4,Float64}
    class_counts::NTuple{4,Float64}
    bayesr_ssq::Float64
    bayesr_nnz::Float64
    state_counts::NTuple{16,Float64}
    n_events::Float64
    sweep_ms::Float64
    update_kernel_ms::Float64
    update_kernel_samples::Float64
    update_kernel_bytes::Float64
    event_overhead_ms::Float64
end

mutable struct HipBackend            # the analogue of Packed2BitBackend (streaming_genotypes.jl:7-25)
    ctx::Ptr{Cvoid}
    nObs::Int
    nMarkers::Int
    block_size::Int
end

hip_error(ctx) = unsafe_string(ccall((:jwas_hip_last_error, LIBJWAS_HIP), Cstring, (Ptr{Cvoid},), ctx))
hip_check(ctx, rc) = rc == 0 || error("libjwas_hip: " * hip_error(ctx))      # error(...) like every JWAS check

"Upload a genotype matrix (after align_genotypes, tools4genotypes.jl:310-321) and precompute x'x and the block Grams."
function HipBackend(X::Matrix{Float32}; device::Integer=0, block_size::Integer=512, invweights=nothing,
                    method::Integer=JWAS_HIP_BAYESC, ntraits::Integer=1)
    ctxref = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:jwas_hip_create, LIBJWAS_HIP), Cint, (Cint, Ref{Ptr{Cvoid}}), device, ctxref)
    rc == 0 || error("libjwas_hip: " * hip_error(C_NULL))
    n, p = size(X)          # Julia matrices are column-major == the device's marker-major layout: no re-layout
    b = HipBackend(ctxref[], n, p, block_size)
    finalizer(x -> ccall((:jwas_hip_destroy, LIBJWAS_HIP), Cvoid, (Ptr{Cvoid},), x.ctx), b)
    hip_check(b.ctx, ccall((:jwas_hip_load_dense_f32, LIBJWAS_HIP), Cint,
                           (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Int64), b.ctx, X, n, p, n))
    if invweights !== nothing          # GibbsMats with Rinv (tools4genotypes.jl:253-266)
        w = Vector{Float32}(invweights)
        hip_check(b.ctx, ccall((:jwas_hip_set_weights, LIBJWAS_HIP), Cint, (Ptr{Cvoid}, Ptr{Float32}), b.ctx, w))
    end
    hip_check(b.ctx, ccall((:jwas_hip_setup_blocks, LIBJWAS_HIP), Cint, (Ptr{Cvoid}, Int32, Int32),
                           b.ctx, block_size, JWAS_HIP_GRAM_MFMA))
    hip_check(b.ctx, ccall((:jwas_hip_init_state, LIBJWAS_HIP), Cint, (Ptr{Cvoid}, Int32, Int32), b.ctx, method, ntraits))
    return b
end

"fast_blocks = a vector of 1-based block starts, possibly non-uniform (JWAS.jl:298-304): the device runs exactly that partition."
function hip_setup_blocks!(b::HipBackend, starts::AbstractVector{<:Integer})
    s0 = Int64.(starts) .- 1
    hip_check(b.ctx, ccall((:jwas_hip_setup_blocks_explicit, LIBJWAS_HIP), Cint, (Ptr{Cvoid}, Ptr{Int64}, Int64, Int32),
                           b.ctx, s0, length(s0), JWAS_HIP_GRAM_MFMA))
    return b
end

hip_set_residual!(b::HipBackend, trait::Integer, r::Vector{Float32}) =
    hip_check(b.ctx, ccall((:jwas_hip_set_residual, LIBJWAS_HIP), Cint, (Ptr{Cvoid}, Int32, Ptr{Float32}), b.ctx, trait, r))
hip_get_residual!(b::HipBackend, trait::Integer, r::Vector{Float32}) =
    hip_check(b.ctx, ccall((:jwas_hip_get_residual, LIBJWAS_HIP), Cint, (Ptr{Cvoid}, Int32, Ptr{Float32}), b.ctx, trait, r))

"Grouped launches for the selected block size: 2 or 4 consecutive blocks per launch of the step kernel (0 frees the buffers); the sweeps
whose HipSweepParams carry group_launch = 1 then use them.  Worth its set-up time from a few thousand iterations on (INTEGRATION.md)."
hip_setup_groups!(b::HipBackend, blocks_per_launch::Integer) =
    hip_check(b.ctx, ccall((:jwas_hip_setup_groups, LIBJWAS_HIP), Cint, (Ptr{Cvoid}, Int32, Int32), b.ctx, blocks_per_launch, JWAS_HIP_GRAM_MFMA))

"Float64 context (runMCMC(double_precision=true), JWAS.jl:349-366): call before loading genotypes; the data entry points are then
jwas_hip_load_dense_f64 / _set_state_f64 / _get_state_f64 / _set_residual_f64 / _get_residual_f64 / _get_posterior_f64 (Ptr{Float64})."
hip_set_precision!(ctx::Ptr{Cvoid}, bits::Integer) =
    hip_check(ctx, ccall((:jwas_hip_set_precision, LIBJWAS_HIP), Cint, (Ptr{Cvoid}, Int32), ctx, bits))

"ycorr .+= shift on the device: the residual correction of an all-ones design column (intercept step, solver.jl:143-162)."
hip_residual_add_scalar!(b::HipBackend, trait::Integer, shift::Real) =
    hip_check(b.ctx, ccall((:jwas_hip_residual_add_scalar, LIBJWAS_HIP), Cint, (Ptr{Cvoid}, Int32, Cdouble), b.ctx, trait, shift))

_z16(::Type{T}) where {T} = ntuple(_ -> zero(T), 16)

"Parameters of a single-trait BayesC / BayesR sweep (the multi-trait samplers fill vare / var_effect / log_prior_states row-major)."
function HipSweepParams(method::Integer, iter::Integer, seed::Integer; vare::Real, var_effect::Real, pi::Real=0.0,
                        pi_classes=(0.0, 0.0, 0.0, 0.0), gamma=(0.0, 0.01, 0.1, 1.0), nreps::Integer=1,
                        marker_offset::Integer=0, independent_blocks::Bool=false,
                        pi_vec::Ptr{Float64}=Ptr{Float64}(C_NULL), pi_matrix::Ptr{Float64}=Ptr{Float64}(C_NULL),
                        var_effect_vec::Ptr{Float32}=Ptr{Float32}(C_NULL), section_solve::Bool=false, group_launch::Bool=false)
    HipSweepParams(Int32(method), Int32(1), Int32(nreps), UInt32(iter), UInt64(seed), UInt32(marker_offset),
                   UInt32(independent_blocks), Base.setindex(_z16(Float32), Float32(vare), 1),
                   Base.setindex(_z16(Float32), Float32(var_effect), 1), Float64(pi), NTuple{4,Float64}(pi_classes),
                   NTuple{4,Float64}(gamma), _z16(Float64), var_effect_vec, pi_vec, pi_matrix, Ptr{Float64}(C_NULL),
                   Ptr{Float32}(C_NULL), Base.setindex(_z16(Float64), Float64(vare), 1),
                   Base.setindex(_z16(Float64), Float64(var_effect), 1), Ptr{Float64}(C_NULL),      # (the Float64 context's copies)
                   Int32(section_solve), Int32(group_launch))                                        # Rule T (multi-trait sampler I, dense prior); grouped launches
end

"One marker sweep = one call of BayesABC! / BayesR! / MTBayesABC! (BayesABC.jl:60-80, BayesR.jl:45-97, MTBayesABC.jl:57-127)."
function hip_sweep!(b::HipBackend, P::HipSweepParams)
    S = Ref{HipSweepStats}()
    hip_check(b.ctx, ccall((:jwas_hip_sweep, LIBJWAS_HIP), Cint, (Ptr{Cvoid}, Ref{HipSweepParams}, Ref{HipSweepStats}), b.ctx, Ref(P), S))
    return S[]
end

# ---- marker shards over the GPUs of a node (one Julia process / task per GPU) -----------------------------------------
"128-byte RCCL id: rank 0 creates it and hands it to the other ranks (a file, a socket, MPI.bcast ...)."
function hip_comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    rc = ccall((:jwas_hip_comm_unique_id, LIBJWAS_HIP), Cint, (Ptr{UInt8},), id)
    rc == 0 || error("libjwas_hip: " * hip_error(C_NULL))
    return id
end
hip_comm_init!(b::HipBackend, id::Vector{UInt8}, rank::Integer, world::Integer) =
    hip_check(b.ctx, ccall((:jwas_hip_comm_init, LIBJWAS_HIP), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32), b.ctx, id, rank, world))

"(rank, world) of the attached communicator as RCCL reports them (ncclCommUserRank / ncclCommCount)."
function hip_comm_info(b::HipBackend)
    r, w = Ref{Int32}(0), Ref{Int32}(1)
    hip_check(b.ctx, ccall((:jwas_hip_comm_info, LIBJWAS_HIP), Cint, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}), b.ctx, r, w))
    return Int(r[]), Int(w[])
end

"Sweep of this rank's markers + on-device reconcile (one ncclAllReduce of delta r and the packed statistics, BayesABC.jl:205-253)."
function hip_sweep_sharded!(b::HipBackend, P::HipSweepParams)
    S = Ref{HipSweepStats}()
    hip_check(b.ctx, ccall((:jwas_hip_sweep_sharded, LIBJWAS_HIP), Cint, (Ptr{Cvoid}, Ref{HipSweepParams}, Ref{HipSweepStats}), b.ctx, Ref(P), S))
    return S[]
end

# ---- posterior accumulators and EBV ------------------------------------------------------------------------------------
hip_accumulate!(b::HipBackend, nsamples::Real) =
    hip_check(b.ctx, ccall((:jwas_hip_accumulate, LIBJWAS_HIP), Cint, (Ptr{Cvoid}, Cdouble), b.ctx, nsamples))

function hip_posterior(b::HipBackend, trait::Integer)
    m, m2, f = (Vector{Float32}(undef, b.nMarkers) for _ in 1:3)
    hip_check(b.ctx, ccall((:jwas_hip_get_posterior, LIBJWAS_HIP), Cint, (Ptr{Cvoid}, Int32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                           b.ctx, trait, m, m2, f))
    return m, m2, f          # output_posterior_mean_variance (output.jl:568-577): mean(alpha), mean(alpha^2), model frequency
end

function hip_mul_alpha(b::HipBackend, trait::Integer)
    out = Vector{Float32}(undef, b.nObs)
    hip_check(b.ctx, ccall((:jwas_hip_mul_alpha, LIBJWAS_HIP), Cint, (Ptr{Cvoid}, Int32, Ptr{Float32}), b.ctx, trait, out))
    return out               # getEBV's X*alpha (output.jl:281-306)
end

"Drop-in for BayesABC!(Mi, ycorr, vare, locus_effect_variances) (BayesABC.jl:10-14) when Mi.storage_mode == :hip."
function BayesABC_hip!(Mi, ycorr::Vector{Float32}, vare::Float32, varEffect::Float32, iter::Integer, seed::Integer)
    b = Mi.hip_backend
    hip_set_residual!(b, 0, ycorr)
    pvec = Mi.π isa AbstractVector ? Vector{Float64}(Mi.π) : Float64[]
    length(pvec) in (0, b.nMarkers) ||                       # bayesabc_pi_vector (BayesABC.jl:16-22)
        error("BayesABC pi vector length $(length(pvec)) must match the number of markers ($(b.nMarkers)).")
    S = GC.@preserve pvec hip_sweep!(b, HipSweepParams(JWAS_HIP_BAYESC, iter, seed; vare=vare, var_effect=varEffect,
                                                       pi=Mi.π isa Number ? Mi.π : 0.0,
                                                       pi_vec=isempty(pvec) ? Ptr{Float64}(C_NULL) : pointer(pvec)))
    hip_get_residual!(b, 0, ycorr)
    return S          # sum_delta -> samplePi (Pi.jl:7-9); alpha_ss -> sample_variance (variance_components.jl:160-162);
end                   # resid_ss -> residual variance (:60-66): no O(p) host pass is needed

end # module
