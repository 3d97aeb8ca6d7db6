# HipGLRM.jl -- the reference-side binding: a new `AbstractParams` subtype plus one `fit!` method that `ccall`s libglrm_hip.so
# (include/glrm_hip.h).  `include("HipGLRM.jl")` next to LowRankModels.jl; every driver that forwards `params=` (fit!,
# cross_validate, cv_by_iter, regularization_path, precision_at_k, the ScikitLearn wrappers -- src/fit.jl:8-12,
# src/cross_validate.jl:10,142,184,243) then runs on the MI355X engine.  Host code stays in Julia; nothing here computes.
#
#   fit!(glrm, HipProxGradParams())              one GPU:  glrm_hip_create + glrm_hip_fit
#   fit!(glrm, HipProxGradParams(ngpus = 8))     eight GPUs, ONE Julia process: glrm_hip_multi_create + glrm_hip_multi_fit
#
# This file is the marshalling core (<= 150 code lines, SURVEY.md section 8(b); tests/test_julia_shim.py counts them):
#   HipGLRMDescriptors.jl  loss / regularizer types -> (kind, dim, scale, p0, p1) / (kind, wrap, scale); which models the engine takes
#   HipGLRMExtras.jl       init_svd! / error_metric / impute / subset / sum_order on the same cached handle
# Omega at north-star scale (1e9 observations): a `SparseMatrixCSC` whose lists are the constructor's (`findall(!iszero, A)`,
# src/glrm.jl:46-48) IS the column view -- colptr / rowval / nzval are handed over after one index shift and NOTHING ELSE: the row view
# (ascending columns per row = the order sort_observations pushes them in, src/modify_glrm.jl:8-12) is derived by the engine on the
# device (GLRM_PROBLEM_ROWS_FROM_COLS), and no entry is looked up through `A[e, j]` (a binary search per observation on a CSC matrix).
# Lists the caller built (obs tuples, duplicates, two different views) are flattened list by list, each view from ITS OWN list.
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: no `julia` binary exists in the build image or on the GPU box (SURVEY.md F2).  The same C entry
# points are exercised from C (examples/c_abi_example.c, examples/c_abi_multi.c) and from Python/ctypes (lowrankmodels.jl_amd/_capi.py,
# whose `flatten` twin is timed at 1e9 observations by bench.py: setup_s.create_from_host).
module HipGLRM

using LowRankModels, SparseArrays
import LowRankModels: fit!, GLRM, AbstractParams, ConvergenceHistory, update_ch!, ProxGradParams

export HipProxGradParams, hip_release!

const LIB = get(ENV, "GLRM_HIP_LIB", "libglrm_hip.so")
const ABI_VERSION = 3                         # GLRM_HIP_ABI_VERSION of the include/glrm_hip.h these struct mirrors were written against

# mirrors of the C structs (include/glrm_hip.h); tests/test_julia_shim.py compares every field offset with the C compiler's
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
struct COptions
    device_id::Int32; profile::Int32; waves_row::Int32; waves_col::Int32; stream::Ptr{Cvoid}
    caller_stream::Int32; tiled::Int32; quad_gram::Int32; sum_order::Int32; reserved0::Int32; reserved::Int32
end
struct CMultiOptions; n_shards::Int32; exchange::Int32; device_ids::Ptr{Int32}; x_chunks::Int32; arrival::Int32; end

include("HipGLRMDescriptors.jl")              # closs / creg / descriptors / dense_ok / value

"""The 7 ProxGradParams fields (src/algorithms/proxgrad.jl:4-12) + where and how to run (include/glrm_hip.h: glrm_options, glrm_multi_options).
Tolerance: `mode = :fast` (production) keeps ch.objective within 1e-5 of the reference's on every committed fixture and equals the CPU oracle bit for
bit in the engine's own summation order; factor ENTRIES of recipes that amplify rounding (NNMF of BASELINE config 4: 7e-3 after 100 iterations, the
oracle against itself in two orders) meet 1e-5 only with `mode = :reference_order` (every sum added as the reference adds it; ~7x slower)."""
mutable struct HipProxGradParams <: AbstractParams
    stepsize::Float64; max_iter::Int; inner_iter_X::Int; inner_iter_Y::Int
    abs_tol::Float64; rel_tol::Float64; min_stepsize::Float64
    device_id::Int; ngpus::Int; device_ids::Vector{Int32}; exchange::Symbol; x_chunks::Int; dense::Bool; quad_gram::Bool
    mode::Symbol                               # :fast (the engine's summation orders) | :reference_order (validation, glrm_options.sum_order = 1)
end
function HipProxGradParams(stepsize::Number=1.0; max_iter::Int=100, inner_iter_X::Int=1, inner_iter_Y::Int=1,
                           inner_iter::Int=1, abs_tol::Number=0.00001, rel_tol::Number=0.0001,
                           min_stepsize::Number=0.01*stepsize, device_id::Int=-1, ngpus::Int=1,
                           device_ids=Int32.(0:ngpus-1), exchange::Symbol=:direct, x_chunks::Int=4, dense::Bool=true,
                           quad_gram::Bool=false, mode::Symbol=:fast)
    length(device_ids) == ngpus || error("device_ids must list one device per shard")
    exchange in (:direct, :rccl) || error("exchange must be :direct or :rccl")
    mode in (:fast, :reference_order) || error("mode must be :fast or :reference_order")
    HipProxGradParams(Float64(stepsize), max_iter, max(inner_iter_X, inner_iter), max(inner_iter_Y, inner_iter),
                      Float64(abs_tol), Float64(rel_tol), Float64(min_stepsize), device_id, ngpus, Vector{Int32}(device_ids),
                      exchange, x_chunks, dense, quad_gram, mode)
end

# ---- Omega -> 0-based CSR / CSC (include/glrm_hip.h: glrm_problem) ------------------------------------------------------
# one view from ITS OWN lists (order and duplicates kept); `at(s, i)` is A[s, i] for the row view, A[i, s] for the column view
function flatten(lists, losses, at, bycol::Bool)
    ptr = Vector{Int64}(undef, length(lists) + 1); ptr[1] = 0
    @inbounds for s in eachindex(lists); ptr[s + 1] = ptr[s] + length(lists[s]); end
    idx = Vector{Int32}(undef, ptr[end]); vals = Vector{Float64}(undef, ptr[end]); t = 0
    @inbounds for s in eachindex(lists), i in lists[s]
        t += 1; idx[t] = Int32(i - 1); vals[t] = value(losses[bycol ? s : i], at(s, i))
    end
    ptr, idx, vals
end
# do the lists say exactly what the CSC arrays say?  (true for GLRM(A::SparseMatrixCSC, ...) without obs, src/glrm.jl:46-48)
function csc_is_omega(A::SparseMatrixCSC, oe)
    cp, rv = A.colptr, A.rowval
    @inbounds for j in eachindex(oe)
        l = oe[j]; length(l) == cp[j + 1] - cp[j] || return false
        for (t, i) in enumerate(l); i == rv[cp[j] + t - 1] || return false; end
    end
    true
end
# ... and are the row lists its transpose (every row's entries by ascending column)?  One walk over the columns with a cursor per row;
# nothing is built.  (They are for the sparse constructor; a caller may have replaced them: the two views are independent.)
function rows_are_transpose(A::SparseMatrixCSC, of)
    cp, rv = A.colptr, A.rowval; cur = ones(Int, size(A, 1))
    @inbounds for j in 1:size(A, 2), t in cp[j]:cp[j+1]-1
        i = rv[t]; c = cur[i]
        (c <= length(of[i]) && of[i][c] == j) || return false
        cur[i] = c + 1
    end
    all(i -> cur[i] == length(of[i]) + 1, eachindex(of))
end
# the column view of a SparseMatrixCSC IS (colptr, rowval, nzval): one index shift, no lookup
function cols_from_csc(A::SparseMatrixCSC, losses)
    cp, rv, nz = A.colptr, A.rowval, A.nzval; n = size(A, 2)
    colptr = Vector{Int64}(undef, n + 1); @inbounds for j in 1:n+1; colptr[j] = cp[j] - 1; end
    rowidx = Vector{Int32}(undef, length(rv)); colvals = Vector{Float64}(undef, length(rv))
    @inbounds for j in 1:n, t in cp[j]:cp[j+1]-1; rowidx[t] = Int32(rv[t] - 1); colvals[t] = value(losses[j], nz[t]); end
    colptr, rowidx, colvals
end
# (rowptr, colidx, rowvals, colptr, rowidx, colvals, flags).  A sparse matrix's pattern goes over as its column view alone
# (GLRM_PROBLEM_ROWS_FROM_COLS = 8: the engine derives the row view on the device, the multi-GPU create on the host); anything else is
# flattened list by list, each view from ITS OWN list.
function omega_views(glrm::GLRM)
    A = glrm.A
    rows() = flatten(glrm.observed_features, glrm.losses, (e, j) -> A[e, j], false)
    if A isa SparseMatrixCSC && csc_is_omega(A, glrm.observed_examples)
        cols = cols_from_csc(A, glrm.losses)
        rows_are_transpose(A, glrm.observed_features) && return (Int64[], Int32[], Float64[], cols..., Int32(8))
        return (rows()..., cols..., Int32(0))
    end
    (rows()..., flatten(glrm.observed_examples, glrm.losses, (j, e) -> A[e, j], true)..., Int32(0))
end

lasterr() = unsafe_string(ccall((:glrm_hip_last_error, LIB), Cstring, ()))
check(rc) = rc == 0 ? nothing : error("glrm_hip [$rc]: " * lasterr())
# a library built from another header revision would read these structs with a different layout: refuse it up front
function check_abi()
    v = ccall((:glrm_hip_version, LIB), Cint, ())
    v == ABI_VERSION || error("libglrm_hip.so speaks ABI $v, HipGLRM.jl was written against ABI $ABI_VERSION (include/glrm_hip.h)")
end

include("HipGLRMHandle.jl")                   # handle(glrm, desc, p): the engine handle cached per model; hip_release!; with_handle

# upload: glrm_problem from the model (Omega views or the dense matrix), glrm_options / glrm_multi_options from the params
function create_handle(glrm::GLRM, desc, p::HipProxGradParams, dense::Bool)
    losses, rx, ry = desc
    check_abi()
    A = glrm.A; m, n = size(A); h = Ref{Ptr{Cvoid}}(C_NULL)
    rowptr, colidx, rowvals, colptr, rowidx, colvals, flags = dense ? (Int64[], Int32[], Float64[], Int64[], Int32[], Float64[], Int32(0)) : omega_views(glrm)
    nul(v) = (dense || (flags == 8 && v !== colptr && v !== rowidx && v !== colvals)) ? Ptr{eltype(v)}(C_NULL) : pointer(v)
    GC.@preserve losses rx ry rowptr colidx rowvals colptr rowidx colvals A p begin
        prob = CProblem(m, n, glrm.k, flags, 0, m, 0, n, nul(rowptr), nul(colidx), nul(rowvals), nul(colptr), nul(rowidx), nul(colvals),
                        pointer(losses), length(losses), pointer(rx), length(rx), pointer(ry), length(ry),
                        dense ? pointer(A) : Ptr{Float64}(C_NULL), dense ? m : 0, dense ? 1 : 0, 0)   # Julia's A is column-major
        opt = COptions(p.device_id, 0, 0, 0, C_NULL, 0, 0, p.quad_gram ? 1 : 0, p.mode == :reference_order ? 1 : 0, 0, 0)
        if p.ngpus > 1
            mo = CMultiOptions(p.ngpus, p.exchange == :rccl ? 1 : 0, pointer(p.device_ids), p.x_chunks, 0)
            check(ccall((:glrm_hip_multi_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Ref{CProblem}, Ref{COptions}, Ref{CMultiOptions}), h, prob, opt, mo))
        else
            check(ccall((:glrm_hip_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Ref{CProblem}, Ref{COptions}), h, prob, opt))
        end
    end                                     # create copied everything: the host arrays may go
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
    check(p.ngpus > 1 ?                      # (a `ccall` target is a constant expression: one literal call per entry point)
          ccall((:glrm_hip_multi_fit, LIB), Cint, (Ptr{Cvoid}, Ref{CParams}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ref{Int64}),
                h, prm, X, Y, obj, sec, cap, nrec) :
          ccall((:glrm_hip_fit, LIB), Cint, (Ptr{Cvoid}, Ref{CParams}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ref{Int64}),
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

end # module
