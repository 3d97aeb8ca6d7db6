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

export HipProxGradParams

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

function fit!(glrm::GLRM, p::HipProxGradParams; ch::ConvergenceHistory=ConvergenceHistory("HipProxGradGLRM"),
              verbose=true, kwargs...)
    cl = map(closs, glrm.losses); crx = map(creg, glrm.rx); cry = map(creg, glrm.ry)
    if any(isnothing, cl) || any(isnothing, crx) || any(isnothing, cry)   # outside the engine: reference path
        return fit!(glrm, ProxGradParams(p.stepsize; max_iter=p.max_iter, inner_iter_X=p.inner_iter_X,
                    inner_iter_Y=p.inner_iter_Y, abs_tol=p.abs_tol, rel_tol=p.rel_tol, min_stepsize=p.min_stepsize);
                    ch=ch, verbose=verbose, kwargs...)
    end
    A = glrm.A; m, n = size(A); k = glrm.k          # glrm.Y is k x embedding_dim(glrm.losses): passed through as is
    (k > 64 && (embedding_dim(glrm.losses) != n || any(c -> c.wrap != 0, crx) || any(c -> c.wrap != 0, cry))) &&
        return fit!(glrm, ProxGradParams(p.stepsize; max_iter=p.max_iter, inner_iter_X=p.inner_iter_X, inner_iter_Y=p.inner_iter_Y,
                    abs_tol=p.abs_tol, rel_tol=p.rel_tol, min_stepsize=p.min_stepsize); ch=ch, verbose=verbose, kwargs...)
    losses = Vector{CLoss}(cl); rx = Vector{CReg}(crx); ry = Vector{CReg}(cry)
    rowptr, colidx, rowvals = flatten(glrm.observed_features, (e, f) -> value(glrm.losses[f], A[e, f]))
    colptr, rowidx, colvals = flatten(glrm.observed_examples, (f, e) -> value(glrm.losses[f], A[e, f]))
    X = glrm.X isa Matrix{Float64} ? glrm.X : Matrix{Float64}(glrm.X); Y = glrm.Y
    cap = p.max_iter + 1
    obj = zeros(cap); sec = zeros(cap); nrec = Ref{Int64}(0); h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve losses rx ry rowptr colidx rowvals colptr rowidx colvals X Y obj sec begin
        prob = CProblem(m, n, k, 0, 0, m, 0, n, pointer(rowptr), pointer(colidx), pointer(rowvals),
                        pointer(colptr), pointer(rowidx), pointer(colvals), pointer(losses), n,
                        pointer(rx), m, pointer(ry), n, C_NULL, 0, 0, 0)
        opt = COptions(p.device_id, 0, 0, 0, C_NULL, 0, 0)
        check(ccall((:glrm_hip_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Ref{CProblem}, Ref{COptions}), h, prob, opt))
        try
            prm = CParams(p.stepsize, p.max_iter, p.inner_iter_X, p.inner_iter_Y, p.abs_tol, p.rel_tol, p.min_stepsize)
            verbose && println("Fitting GLRM")
            check(ccall((:glrm_hip_fit, LIB), Cint,
                        (Ptr{Cvoid}, Ref{CParams}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ref{Int64}),
                        h[], prm, X, Y, obj, sec, cap, nrec))
        finally
            ccall((:glrm_hip_destroy, LIB), Cvoid, (Ptr{Cvoid},), h[])
        end
    end
    X === glrm.X || copyto!(glrm.X, X)
    for i in 1:nrec[]
        update_ch!(ch, i == 1 ? 0.0 : sec[i] - sec[i-1], obj[i])
        (verbose && i > 1 && (i - 1) % 10 == 0 && i < nrec[]) && println("Iteration $(i-1): objective value = $(obj[i])")
    end
    return glrm.X, glrm.Y, ch
end

end # module
