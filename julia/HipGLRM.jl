# HipGLRM.jl -- the reference-side binding: a new `AbstractParams` subtype plus one `fit!` method that
# `ccall`s libglrm_hip.so (include/glrm_hip.h).  Drop this file next to LowRankModels.jl and
# `include("HipGLRM.jl")`; every driver that forwards `params=` (fit!, cross_validate, cv_by_iter,
# regularization_path, precision_at_k, the ScikitLearn wrappers) then runs on the MI355X engine.
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: no `julia` binary exists in the build image or on the GPU box
# (SURVEY.md F2).  It is pure marshalling; the same C entry points are exercised from Python/ctypes.
module HipGLRM

using LowRankModels
import LowRankModels: fit!, GLRM, AbstractParams, ConvergenceHistory, update_ch!,
                      Loss, Regularizer, QuadLoss, L1Loss, HuberLoss, QuantileLoss, PeriodicLoss, PoissonLoss,
                      OrdinalHingeLoss, LogisticLoss, WeightedHingeLoss,
                      MultinomialLoss, OvALoss, BvSLoss, OrdisticLoss, MultinomialOrdinalLoss, embedding_dim,
                      ZeroReg, QuadReg, OneReg, NonNegConstraint, UnitOneSparseConstraint,
                      lastentry1, lastentry_unpenalized, OrdinalReg, MNLOrdinalReg, ProxGradParams

export HipProxGradParams, hip_init_svd!, hip_error_metric, hip_impute

const LIB = get(ENV, "GLRM_HIP_LIB", "libglrm_hip.so")

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
struct COptions; device_id::Int32; profile::Int32; waves_row::Int32; waves_col::Int32; stream::Ptr{Cvoid}; caller_stream::Int32; tiled::Int32; end

"The 7 ProxGradParams fields (src/algorithms/proxgrad.jl:4-12) + the device ordinal."
mutable struct HipProxGradParams <: AbstractParams
    stepsize::Float64; max_iter::Int; inner_iter_X::Int; inner_iter_Y::Int
    abs_tol::Float64; rel_tol::Float64; min_stepsize::Float64; device_id::Int
end
function HipProxGradParams(stepsize::Number=1.0; max_iter::Int=100, inner_iter_X::Int=1, inner_iter_Y::Int=1,
                           inner_iter::Int=1, abs_tol::Number=0.00001, rel_tol::Number=0.0001,
                           min_stepsize::Number=0.01*stepsize, device_id::Int=-1)
    HipProxGradParams(Float64(stepsize), max_iter, max(inner_iter_X, inner_iter), max(inner_iter_Y, inner_iter),
                      Float64(abs_tol), Float64(rel_tol), Float64(min_stepsize), device_id)
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
value(l, a) = isclass(l) ? (a isa Bool ? Float64(a) : Float64(LowRankModels.myBool(a))) : Float64(a)

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

fallback(glrm, p; kw...) = fit!(glrm, ProxGradParams(p.stepsize; max_iter=p.max_iter, inner_iter_X=p.inner_iter_X, inner_iter_Y=p.inner_iter_Y,
                                             abs_tol=p.abs_tol, rel_tol=p.rel_tol, min_stepsize=p.min_stepsize); kw...)

# descriptors of a model, or nothing if some loss / regularizer type is outside include/glrm_hip.h
function descriptors(glrm::GLRM)
    cl = map(closs, glrm.losses); crx = map(creg, glrm.rx); cry = map(creg, glrm.ry)
    (any(isnothing, cl) || any(isnothing, crx) || any(isnothing, cry)) && return nothing
    n = size(glrm.A, 2)
    general = embedding_dim(glrm.losses) != n || any(c -> c.wrap != 0, crx) || any(c -> c.wrap != 0, cry)
    (general && glrm.k > 64) && return nothing
    Vector{CLoss}(cl), Vector{CReg}(crx), Vector{CReg}(cry)
end

# f(handle) on an engine handle holding the model's Omega views and values; the handle lives for the call.
# (A host that calls several entry points in a row -- init_svd!, fit!, error_metric -- keeps it instead.)
function with_handle(f, glrm::GLRM, desc, device_id::Int=-1)
    losses, rx, ry = desc
    A = glrm.A; m, n = size(A)
    rowptr, colidx, rowvals = flatten(glrm.observed_features, (e, j) -> value(glrm.losses[j], A[e, j]))
    colptr, rowidx, colvals = flatten(glrm.observed_examples, (j, e) -> value(glrm.losses[j], A[e, j]))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve losses rx ry rowptr colidx rowvals colptr rowidx colvals begin
        prob = CProblem(m, n, glrm.k, 0, 0, m, 0, n, pointer(rowptr), pointer(colidx), pointer(rowvals),
                        pointer(colptr), pointer(rowidx), pointer(colvals), pointer(losses), n,
                        pointer(rx), m, pointer(ry), n, C_NULL, 0, 0, 0)
        opt = COptions(device_id, 0, 0, 0, C_NULL, 0, 0)
        check(ccall((:glrm_hip_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Ref{CProblem}, Ref{COptions}), h, prob, opt))
    end                                     # create copied everything: the host arrays may go
    try
        return f(h[])
    finally
        ccall((:glrm_hip_destroy, LIB), Cvoid, (Ptr{Cvoid},), h[])
    end
end

function fit!(glrm::GLRM, p::HipProxGradParams; ch::ConvergenceHistory=ConvergenceHistory("HipProxGradGLRM"),
              verbose=true, kwargs...)
    desc = descriptors(glrm)
    desc === nothing && return fallback(glrm, p; ch=ch, verbose=verbose, kwargs...)   # outside the engine: reference path
    X = glrm.X isa Matrix{Float64} ? glrm.X : Matrix{Float64}(glrm.X); Y = glrm.Y    # Y is k x embedding_dim(glrm.losses)
    cap = p.max_iter + 1
    obj = zeros(cap); sec = zeros(cap); nrec = Ref{Int64}(0)
    with_handle(glrm, desc, p.device_id) do h
        prm = CParams(p.stepsize, p.max_iter, p.inner_iter_X, p.inner_iter_Y, p.abs_tol, p.rel_tol, p.min_stepsize)
        verbose && println("Fitting GLRM")
        check(ccall((:glrm_hip_fit, LIB), Cint,
                    (Ptr{Cvoid}, Ref{CParams}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ref{Int64}),
                    h, prm, X, Y, obj, sec, cap, nrec))
    end
    X === glrm.X || copyto!(glrm.X, X)
    for i in 1:nrec[]
        update_ch!(ch, i == 1 ? 0.0 : sec[i] - sec[i-1], obj[i])
        (verbose && i > 1 && (i - 1) % 10 == 0 && i < nrec[]) && println("Iteration $(i-1): objective value = $(obj[i])")
    end
    return glrm.X, glrm.Y, ch
end

# ---- the entry points around fit! (src/initialize.jl:35-132, src/evaluate_fit.jl:107-168, src/impute_and_err.jl) ----

struct CDomain; kind::Int32; reserved::Int32; lo::Float64; hi::Float64; end
cdomain(d::LowRankModels.RealDomain) = CDomain(0, 0, 0, 0)
cdomain(d::LowRankModels.BoolDomain) = CDomain(1, 0, 0, 0)
cdomain(d::LowRankModels.OrdinalDomain) = CDomain(2, 0, d.min, d.max)
cdomain(d::LowRankModels.PeriodicDomain) = CDomain(3, 0, d.T, 0)
cdomain(d::LowRankModels.CountDomain) = CDomain(4, 0, 0, d.max_count)
cdomain(d::LowRankModels.CategoricalDomain) = CDomain(5, 0, d.min, d.max)

"init_svd!(glrm) on the device (the engine's subspace iteration in place of Arpack's svds); falls back to the reference."
function hip_init_svd!(glrm::GLRM; device_id::Int=-1, tol=1e-10, max_iter=0, seed=1)
    desc = descriptors(glrm)
    desc === nothing && return LowRankModels.init_svd!(glrm)
    X = Matrix{Float64}(undef, size(glrm.X)); Y = Matrix{Float64}(undef, size(glrm.Y))
    with_handle(glrm, desc, device_id) do h
        check(ccall((:glrm_hip_init_svd, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32, Float64, UInt64, Ptr{Float64}, Ptr{Int32}),
                    h, X, Y, max_iter, tol, seed, C_NULL, C_NULL))
    end
    copyto!(glrm.X, X); copyto!(glrm.Y, Y)
    glrm
end

"error_metric(glrm, X, Y, domains; standardize) evaluated on the device; usable as `error_fn` of cross_validate."
function hip_error_metric(glrm::GLRM, X::Matrix{Float64}, Y::Matrix{Float64},
                          domains=[l.domain for l in glrm.losses]; standardize=false, device_id::Int=-1)
    desc = descriptors(glrm)
    desc === nothing && return LowRankModels.error_metric(glrm, X, Y, domains; standardize=standardize)
    doms = CDomain[cdomain(d) for d in domains]; out = Ref{Float64}(0.0)
    with_handle(glrm, desc, device_id) do h
        check(ccall((:glrm_hip_error_metric, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{CDomain}, Int32, Ref{Float64}),
                    h, X, Y, doms, standardize ? 1 : 0, out))
    end
    out[]
end

"impute(glrm): the m x n matrix of imputed values (Bool columns as 1.0 / 0.0)."
function hip_impute(glrm::GLRM; device_id::Int=-1)
    desc = descriptors(glrm)
    desc === nothing && return LowRankModels.impute(glrm)
    m, n = size(glrm.A); Ahat = Matrix{Float64}(undef, m, n)
    doms = CDomain[cdomain(l.domain) for l in glrm.losses]
    with_handle(glrm, desc, device_id) do h
        check(ccall((:glrm_hip_impute, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{CDomain}, Ptr{Float64}),
                    h, Matrix{Float64}(glrm.X), glrm.Y, doms, Ahat))
    end
    Ahat
end

# Train / test split of a fold on the device: `tags` labels the entries of observed_features in flatten_observations order,
# `ctags` the entries of observed_examples; see cross_validate in lowrankmodels.jl_amd/crossval.py for the complete driver.
function hip_subset(parent::Ptr{Cvoid}, tags::Vector{UInt8}, ctags::Vector{UInt8}, fold::Integer; invert::Bool)
    child = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:glrm_hip_subset, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Ptr{UInt8}, Int32, Int32, Ref{Ptr{Cvoid}}),
                parent, tags, ctags, fold, invert ? 1 : 0, child))
    child[]
end

end # module
