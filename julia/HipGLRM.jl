# HipGLRM.jl -- the reference-side binding: a new `AbstractParams` subtype plus one `fit!` method that `ccall`s libglrm_hip.so
# (include/glrm_hip.h).  `include("HipGLRM.jl")` next to LowRankModels.jl; every driver that forwards `params=` (fit!,
# cross_validate, cv_by_iter, regularization_path, precision_at_k, the ScikitLearn wrappers -- src/fit.jl:8-12,
# src/cross_validate.jl:10,142,184,243) then runs on the MI355X engine.  Host code stays in Julia; nothing here computes.
#
#   fit!(glrm, HipProxGradParams())                       one GPU            glrm_hip_create + glrm_hip_fit
#   fit!(glrm, HipProxGradParams(ngpus = 8))              eight GPUs, ONE Julia process: glrm_hip_multi_create + glrm_hip_multi_fit
#                                                         (the library shards rows / columns, replicates X, Y and exchanges the
#                                                         updated blocks over xGMI after every half-step)
# * a fully observed single-QuadLoss model hands `glrm.A` over as the dense matrix it is (column-major, `dense_colmajor = 1`):
#   the half-steps then run on the fp64 matrix cores and no index list is built (BASELINE config 3: 80 GB instead of 2 x 120 GB);
# * the engine handle (Omega views and A on the device) is cached per model, so warm starts, `cv_by_iter`'s `max_iter = 1` loop
#   (src/cross_validate.jl:164-175) and `regularization_path` do not re-upload; new regularizers only replace descriptors;
# * loss / regularizer types outside include/glrm_hip.h fall back to the reference solver.
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: no `julia` binary exists in the build image or on the GPU box (SURVEY.md F2).  It is pure
# marshalling; the same C entry points are exercised from C (examples/c_abi_example.c, examples/c_abi_multi.c) and Python/ctypes.
module HipGLRM

using LowRankModels
import LowRankModels: fit!, GLRM, AbstractParams, ConvergenceHistory, update_ch!,
                      Loss, Regularizer, QuadLoss, L1Loss, HuberLoss, QuantileLoss, PeriodicLoss, PoissonLoss,
                      OrdinalHingeLoss, LogisticLoss, WeightedHingeLoss,
                      MultinomialLoss, OvALoss, BvSLoss, OrdisticLoss, MultinomialOrdinalLoss, embedding_dim,
                      ZeroReg, QuadReg, OneReg, NonNegConstraint, UnitOneSparseConstraint,
                      lastentry1, lastentry_unpenalized, OrdinalReg, MNLOrdinalReg, ProxGradParams

export HipProxGradParams, hip_release!

const LIB = get(ENV, "GLRM_HIP_LIB", "libglrm_hip.so")
const ABI_VERSION = 2                         # GLRM_HIP_ABI_VERSION of the include/glrm_hip.h these struct mirrors were written against

# mirrors of the C structs (include/glrm_hip.h)
struct CLoss; kind::Int32; dim::Int32; scale::Float64; p0::Float64; p1::Float64; end   # dim = embedding_dim (0/1: scalar)
struct CReg;  kind::Int32; wrap::Int32; scale::Float64; end                             # wrap = GLRM_WRAP_* flag
struct CProblem
    m::Int64; n::Int64; k::Int32; flags::Int32
    row_begin::Int64; row_end::Int64; col_begin::Int64; col_end::Int64
    rowptr::Ptr{Int64}; colidx::Ptr{Int32}; rowvals::Ptr{Float64}
    colptr::Ptr{Int64}; rowidx::Ptr{Int32}; colvals::Ptr{Float64}
    losses::Ptr{CLoss}; n_losses::Int64; rx::Ptr{CReg}; n_rx::Int64; ry::Ptr{CReg}; n_ry::Int64
    dense_A::Ptr{Float64}; dense_ld::Int64; dense_colmajor::Int32; dense_reserved::Int32
end
struct CParams
    stepsize::Float64; max_iter::Int64; inner_iter_X::Int64; inner_iter_Y::Int64
    abs_tol::Float64; rel_tol::Float64; min_stepsize::Float64
end
struct COptions; device_id::Int32; profile::Int32; waves_row::Int32; waves_col::Int32; stream::Ptr{Cvoid}; caller_stream::Int32; tiled::Int32; quad_gram::Int32; reserved::Int32; end
struct CMultiOptions; n_shards::Int32; exchange::Int32; device_ids::Ptr{Int32}; x_chunks::Int32; reserved::Int32; end
# glrm_signature: only hosts that shard a problem THEMSELVES (one handle per shard, GLRM_PROBLEM_DEFER_SETUP) need it; this shim hands
# the whole problem to glrm_hip_multi_create, which does that internally.  Mirrored so that tests/test_julia_shim.py checks it too.
struct CSignature; nnz_rows::Int64; nnz_cols::Int64; max_row_len::Int64; max_col_len::Int64; rows_unordered::Int32; cols_unordered::Int32; end
# glrm_sum_order: the order in which a handle's kernels add the terms of a row's / column's sums (diagnostic, see hip_sum_order below)
struct CSumOrder
    family::Int32; lanes::Int32; comps::Int32; waves::Int32
    waves4_from::Int64; waves8_from::Int64; cached_maxlen::Int64
    cached_waves::Int32; batch::Int32; batch_one_wave_only::Int32; rotate::Int32
    window::Int64; windows_per_sup::Int64
    private_order::Int32; reserved::Int32
end

"The 7 ProxGradParams fields (src/algorithms/proxgrad.jl:4-12) + where to run: `device_id` (one GPU) or `ngpus` / `device_ids`."
mutable struct HipProxGradParams <: AbstractParams
    stepsize::Float64; max_iter::Int; inner_iter_X::Int; inner_iter_Y::Int
    abs_tol::Float64; rel_tol::Float64; min_stepsize::Float64
    device_id::Int; ngpus::Int; device_ids::Vector{Int32}; exchange::Symbol; x_chunks::Int; dense::Bool; quad_gram::Bool
end
function HipProxGradParams(stepsize::Number=1.0; max_iter::Int=100, inner_iter_X::Int=1, inner_iter_Y::Int=1,
                           inner_iter::Int=1, abs_tol::Number=0.00001, rel_tol::Number=0.0001,
                           min_stepsize::Number=0.01*stepsize, device_id::Int=-1, ngpus::Int=1,
                           device_ids=Int32.(0:ngpus-1), exchange::Symbol=:direct, x_chunks::Int=4, dense::Bool=true, quad_gram::Bool=false)
    length(device_ids) == ngpus || error("device_ids must list one device per shard")
    exchange in (:direct, :rccl) || error("exchange must be :direct or :rccl")
    HipProxGradParams(Float64(stepsize), max_iter, max(inner_iter_X, inner_iter), max(inner_iter_Y, inner_iter),
                      Float64(abs_tol), Float64(rel_tol), Float64(min_stepsize), device_id, ngpus, Vector{Int32}(device_ids),
                      exchange, x_chunks, dense, quad_gram)
end

closs(l::QuadLoss) = CLoss(0, 0, l.scale, 0, 0)
closs(l::L1Loss) = CLoss(1, 0, l.scale, 0, 0)
closs(l::HuberLoss) = CLoss(2, 0, l.scale, l.crossover, 0)
closs(l::QuantileLoss) = CLoss(3, 0, l.scale, l.quantile, 0)
closs(l::PeriodicLoss) = CLoss(4, 0, l.scale, l.T, 0)
closs(l::PoissonLoss) = CLoss(5, 0, l.scale, 0, 0)
closs(l::OrdinalHingeLoss) = CLoss(6, 0, l.scale, l.min, l.max)
closs(l::LogisticLoss) = CLoss(7, 0, l.scale, 0, 0)
closs(l::WeightedHingeLoss) = CLoss(8, 0, l.scale, l.case_weight_ratio, 0)
# multi-dimensional losses (src/losses.jl:360-620): dim columns of Y per column of A; bin_loss must be Logistic / Hinge
binkind(b::LogisticLoss) = 7.0
binkind(b::WeightedHingeLoss) = b.case_weight_ratio == 1 ? 8.0 : NaN
binkind(b) = NaN
closs(l::MultinomialLoss) = CLoss(9, l.max, l.scale, 0, 0)
closs(l::OvALoss) = isnan(binkind(l.bin_loss)) ? nothing : CLoss(10, l.max, l.scale, l.bin_loss.scale, binkind(l.bin_loss))
closs(l::BvSLoss) = isnan(binkind(l.bin_loss)) ? nothing : CLoss(11, l.max - 1, l.scale, l.bin_loss.scale, binkind(l.bin_loss))
closs(l::OrdisticLoss) = CLoss(12, l.max, l.scale, 0, 0)
closs(l::MultinomialOrdinalLoss) = CLoss(13, l.max - 1, l.scale, 0, 0)
closs(l::Loss) = nothing                     # anything else: reference path
creg(r::ZeroReg) = CReg(0, 0, 1.0)
creg(r::QuadReg) = CReg(1, 0, r.scale)
creg(r::OneReg) = CReg(2, 0, r.scale)
creg(r::NonNegConstraint) = CReg(3, 0, 1.0)
creg(r::UnitOneSparseConstraint) = CReg(4, 0, 1.0)
# wrappers around one of the five base regularizers (src/regularizers.jl:163-189,356-411)
wrapped(r, flag) = (b = creg(r.r); (b === nothing || b.wrap != 0) ? nothing : CReg(b.kind, flag, b.scale))
creg(r::lastentry1) = wrapped(r, 1)
creg(r::lastentry_unpenalized) = wrapped(r, 2)
creg(r::OrdinalReg) = wrapped(r, 4)
creg(r::MNLOrdinalReg) = wrapped(r, 8)
creg(r::Regularizer) = nothing

isclass(l) = l isa LogisticLoss || l isa WeightedHingeLoss
value(l, a) = isclass(l) ? (a isa Bool ? Float64(a) : Float64(LowRankModels.myBool(Int(a)))) : Float64(a)   # src/losses.jl:104-106
collapse(v) = all(==(v[1]), v) ? v[1:1] : v          # one descriptor when every column / row carries the same one

# observed_features / observed_examples -> 0-based CSR / CSC, each built from ITS OWN list (order and duplicates kept)
function flatten(lists, getval)
    ptr = Vector{Int64}(undef, length(lists) + 1); ptr[1] = 0
    for (s, l) in enumerate(lists); ptr[s + 1] = ptr[s] + length(l); end
    idx = Vector{Int32}(undef, ptr[end]); vals = Vector{Float64}(undef, ptr[end])
    t = 1
    for (s, l) in enumerate(lists), i in l
        idx[t] = Int32(i - 1); vals[t] = getval(s, i); t += 1
    end
    ptr, idx, vals
end

lasterr() = unsafe_string(ccall((:glrm_hip_last_error, LIB), Cstring, ()))
check(rc) = rc == 0 ? nothing : error("glrm_hip [$rc]: " * lasterr())
# a library built from another header revision would read these structs with a different layout: refuse it up front
function check_abi()
    v = ccall((:glrm_hip_version, LIB), Cint, ())
    v == ABI_VERSION || error("libglrm_hip.so speaks ABI $v, HipGLRM.jl was written against ABI $ABI_VERSION (include/glrm_hip.h)")
end

fallback(glrm, p; kw...) = fit!(glrm, ProxGradParams(p.stepsize; max_iter=p.max_iter, inner_iter_X=p.inner_iter_X, inner_iter_Y=p.inner_iter_Y,
                                             abs_tol=p.abs_tol, rel_tol=p.rel_tol, min_stepsize=p.min_stepsize); kw...)

# descriptors of a model, or nothing if some loss / regularizer type is outside include/glrm_hip.h
function descriptors(glrm::GLRM)
    cl = map(closs, glrm.losses); crx = map(creg, glrm.rx); cry = map(creg, glrm.ry)
    (any(isnothing, cl) || any(isnothing, crx) || any(isnothing, cry)) && return nothing
    n = size(glrm.A, 2)
    general = embedding_dim(glrm.losses) != n || any(c -> c.wrap != 0, crx) || any(c -> c.wrap != 0, cry)
    (general && glrm.k > 64) && return nothing
    collapse(Vector{CLoss}(cl)), collapse(Vector{CReg}(crx)), collapse(Vector{CReg}(cry))
end

# the dense hand-over applies when every entry is observed (the constructor's default UnitRanges) under one QuadLoss
fully_observed(glrm) = (s = size(glrm.A); all(==(1:s[2]), glrm.observed_features) && all(==(1:s[1]), glrm.observed_examples))
dense_ok(glrm, desc, p) = p.dense && glrm.A isa Matrix{Float64} && length(desc[1]) == 1 && desc[1][1].kind == 0 &&
                          9 <= glrm.k <= 64 && fully_observed(glrm)

# ---- engine handles, cached per model ---------------------------------------------------------------------------------
mutable struct Entry; h::Ptr{Cvoid}; multi::Bool; hard::UInt64; soft::UInt64; end
const CACHE = IdDict{Any,Entry}()
destroy(e::Entry) = (e.h == C_NULL || ccall(e.multi ? (:glrm_hip_multi_destroy, LIB) : (:glrm_hip_destroy, LIB), Cvoid, (Ptr{Cvoid},), e.h); e.h = C_NULL)
"Drop the device copy of a model's data (also done by the model's finalizer).  Call it after mutating `glrm.A` in place."
hip_release!(glrm::GLRM) = (haskey(CACHE, glrm) && (destroy(CACHE[glrm]); delete!(CACHE, glrm)); glrm)
# what the device copy depends on (data, Omega, losses, placement) / what set_regularizers can replace
hardkey(glrm, desc, p, dense) = hash((objectid(glrm.A), size(glrm.A), glrm.k, objectid(glrm.observed_features), objectid(glrm.observed_examples),
                                      sum(length, glrm.observed_features), sum(length, glrm.observed_examples), desc[1],
                                      length(desc[2]), length(desc[3]), p.device_id, p.ngpus, p.device_ids, p.exchange, p.x_chunks, dense, p.quad_gram))
softkey(desc) = hash((desc[2], desc[3]))

function handle(glrm::GLRM, desc, p::HipProxGradParams)
    losses, rx, ry = desc
    dense = dense_ok(glrm, desc, p); multi = p.ngpus > 1
    hard, soft = hardkey(glrm, desc, p, dense), softkey(desc)
    e = get(CACHE, glrm, nothing)
    if e !== nothing && e.hard == hard
        if e.soft != soft                   # scale_regularizer! / regularization_path: Omega and A stay on the device
            check(ccall(multi ? (:glrm_hip_multi_set_regularizers, LIB) : (:glrm_hip_set_regularizers, LIB), Cint,
                        (Ptr{Cvoid}, Ptr{CReg}, Int64, Ptr{CReg}, Int64), e.h, rx, length(rx), ry, length(ry)))
            e.soft = soft
        end
        return e.h
    end
    e === nothing ? finalizer(hip_release!, glrm) : destroy(e)
    check_abi()
    A = glrm.A; m, n = size(A); h = Ref{Ptr{Cvoid}}(C_NULL)
    rowptr, colidx, rowvals = dense ? (Int64[], Int32[], Float64[]) : flatten(glrm.observed_features, (e, j) -> value(glrm.losses[j], A[e, j]))
    colptr, rowidx, colvals = dense ? (Int64[], Int32[], Float64[]) : flatten(glrm.observed_examples, (j, e) -> value(glrm.losses[j], A[e, j]))
    nul(v) = dense ? Ptr{eltype(v)}(C_NULL) : pointer(v)
    GC.@preserve losses rx ry rowptr colidx rowvals colptr rowidx colvals A p begin
        prob = CProblem(m, n, glrm.k, 0, 0, m, 0, n, nul(rowptr), nul(colidx), nul(rowvals), nul(colptr), nul(rowidx), nul(colvals),
                        pointer(losses), length(losses), pointer(rx), length(rx), pointer(ry), length(ry),
                        dense ? pointer(A) : Ptr{Float64}(C_NULL), dense ? m : 0, dense ? 1 : 0, 0)   # Julia's A is column-major
        opt = COptions(p.device_id, 0, 0, 0, C_NULL, 0, 0, p.quad_gram ? 1 : 0, 0)
        if multi
            mo = CMultiOptions(p.ngpus, p.exchange == :rccl ? 1 : 0, pointer(p.device_ids), p.x_chunks, 0)
            check(ccall((:glrm_hip_multi_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Ref{CProblem}, Ref{COptions}, Ref{CMultiOptions}), h, prob, opt, mo))
        else
            check(ccall((:glrm_hip_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Ref{CProblem}, Ref{COptions}), h, prob, opt))
        end
    end                                     # create copied everything: the host arrays may go
    CACHE[glrm] = Entry(h[], multi, hard, soft)
    h[]
end

function fit!(glrm::GLRM, p::HipProxGradParams; ch::ConvergenceHistory=ConvergenceHistory("HipProxGradGLRM"),
              verbose=true, kwargs...)
    desc = descriptors(glrm)
    desc === nothing && return fallback(glrm, p; ch=ch, verbose=verbose, kwargs...)   # outside the engine: reference path
    X = glrm.X isa Matrix{Float64} ? glrm.X : Matrix{Float64}(glrm.X); Y = glrm.Y    # Y is k x embedding_dim(glrm.losses)
    cap = p.max_iter + 1
    obj = zeros(cap); sec = zeros(cap); nrec = Ref{Int64}(0)
    h = handle(glrm, desc, p)
    prm = CParams(p.stepsize, p.max_iter, p.inner_iter_X, p.inner_iter_Y, p.abs_tol, p.rel_tol, p.min_stepsize)
    verbose && println("Fitting GLRM")
    check(ccall(p.ngpus > 1 ? (:glrm_hip_multi_fit, LIB) : (:glrm_hip_fit, LIB), Cint,
                (Ptr{Cvoid}, Ref{CParams}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ref{Int64}),
                h, prm, X, Y, obj, sec, cap, nrec))
    X === glrm.X || copyto!(glrm.X, X)
    scaled_abs_tol = p.abs_tol * sum(length, glrm.observed_features)                  # src/algorithms/proxgrad.jl:72
    for i in 1:nrec[]
        update_ch!(ch, i == 1 ? 0.0 : sec[i] - sec[i-1], obj[i])
        # the reference prints every 10th iteration it did NOT stop at (:210-216) -- including iteration max_iter when the run ends there
        it = i - 1
        if verbose && it >= 1 && it % 10 == 0
            dec = obj[i-1] - obj[i]
            stopped = it > 10 && (dec < scaled_abs_tol || dec / obj[i] < p.rel_tol)
            stopped || println("Iteration $it: objective value = $(obj[i])")
        end
    end
    return glrm.X, glrm.Y, ch
end

# a single-device handle for the other entry points (julia/HipGLRMExtras.jl); nothing if the model is outside the engine
function with_handle(f, glrm::GLRM, device_id::Int=-1)
    desc = descriptors(glrm)
    desc === nothing && return nothing
    p = HipProxGradParams(device_id=device_id, dense=false)      # list handle: init_svd / impute / subset work on the Omega views
    Some(f(handle(glrm, desc, p)))
end

end # module
