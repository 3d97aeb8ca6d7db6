# HipGLRMExtras.jl -- the entry points AROUND fit! on the same cached engine handle (include after HipGLRM.jl):
# init_svd! (src/initialize.jl:35-132), error_metric / impute (src/evaluate_fit.jl:107-168, src/impute_and_err.jl) and the
# device-side train / test split of the cross-validation drivers (src/cross_validate.jl:54-105).  Not executed here (no julia).
module HipGLRMExtras

using LowRankModels
using ..HipGLRM
import ..HipGLRM: LIB, check, with_handle

export hip_init_svd!, hip_error_metric, hip_impute, hip_subset


# glrm_signature: only hosts that shard a problem THEMSELVES (one handle per shard, GLRM_PROBLEM_DEFER_SETUP) need it; the fit! shim hands
# the whole problem to glrm_hip_multi_create, which does that internally.  Mirrored so that tests/test_julia_shim.py checks it too.
struct CSignature; nnz_rows::Int64; nnz_cols::Int64; max_row_len::Int64; max_col_len::Int64; rows_unordered::Int32; cols_unordered::Int32; end
# glrm_sum_order: the order in which a handle's kernels add the terms of a row's / column's sums (diagnostic, see hip_sum_order below)
struct CSumOrder
    family::Int32; lanes::Int32; comps::Int32; waves::Int32
    waves4_from::Int64; waves8_from::Int64; cached_maxlen::Int64
    cached_waves::Int32; batch::Int32; batch_one_wave_only::Int32; rotate::Int32
    window::Int64; windows_per_sup::Int64
    private_order::Int32; long_from::Int32
end
struct CDomain; kind::Int32; reserved::Int32; lo::Float64; hi::Float64; end
cdomain(d::LowRankModels.RealDomain) = CDomain(0, 0, 0, 0)
cdomain(d::LowRankModels.BoolDomain) = CDomain(1, 0, 0, 0)
cdomain(d::LowRankModels.OrdinalDomain) = CDomain(2, 0, d.min, d.max)
cdomain(d::LowRankModels.PeriodicDomain) = CDomain(3, 0, d.T, 0)
cdomain(d::LowRankModels.CountDomain) = CDomain(4, 0, 0, d.max_count)
cdomain(d::LowRankModels.CategoricalDomain) = CDomain(5, 0, d.min, d.max)

"init_svd!(glrm) on the device (the engine's subspace iteration in place of Arpack's svds); falls back to the reference."
function hip_init_svd!(glrm::GLRM; device_id::Int=-1, tol=1e-10, max_iter=0, seed=1)
    X = Matrix{Float64}(undef, size(glrm.X)); Y = Matrix{Float64}(undef, size(glrm.Y))
    r = with_handle(glrm, device_id) do h
        check(ccall((:glrm_hip_init_svd, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32, Float64, UInt64, Ptr{Float64}, Ptr{Int32}),
                    h, X, Y, max_iter, tol, seed, C_NULL, C_NULL))
    end
    r === nothing && return LowRankModels.init_svd!(glrm)
    copyto!(glrm.X, X); copyto!(glrm.Y, Y)
    glrm
end

"error_metric(glrm, X, Y, domains; standardize) evaluated on the device; usable as `error_fn` of cross_validate."
function hip_error_metric(glrm::GLRM, X::Matrix{Float64}, Y::Matrix{Float64},
                          domains=[l.domain for l in glrm.losses]; standardize=false, device_id::Int=-1)
    doms = CDomain[cdomain(d) for d in domains]; out = Ref{Float64}(0.0)
    r = with_handle(glrm, device_id) do h
        check(ccall((:glrm_hip_error_metric, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{CDomain}, Int32, Ref{Float64}),
                    h, X, Y, doms, standardize ? 1 : 0, out))
    end
    r === nothing ? LowRankModels.error_metric(glrm, X, Y, domains; standardize=standardize) : out[]
end

"impute(glrm): the m x n matrix of imputed values (Bool columns as 1.0 / 0.0)."
function hip_impute(glrm::GLRM; device_id::Int=-1)
    m, n = size(glrm.A); Ahat = Matrix{Float64}(undef, m, n)
    doms = CDomain[cdomain(l.domain) for l in glrm.losses]
    r = with_handle(glrm, device_id) do h
        check(ccall((:glrm_hip_impute, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{CDomain}, Ptr{Float64}),
                    h, Matrix{Float64}(glrm.X), glrm.Y, doms, Ahat))
    end
    r === nothing ? LowRankModels.impute(glrm) : Ahat
end

# Train / test split of a fold on the device: `tags` labels the entries of observed_features in flatten_observations order,
# `ctags` the entries of observed_examples; see cross_validate in lowrankmodels.jl_amd/crossval.py for the complete driver.
function hip_subset(parent::Ptr{Cvoid}, tags::Vector{UInt8}, ctags::Vector{UInt8}, fold::Integer; invert::Bool)
    child = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:glrm_hip_subset, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Ptr{UInt8}, Int32, Int32, Ref{Ptr{Cvoid}}),
                parent, tags, ctags, fold, invert ? 1 : 0, child))
    child[]
end

# The order in which the engine adds a row's (which = 0) or a column's (which = 1) loss and gradient terms for THIS model -- the
# attribution tool for trajectories that leave the CPU solver's on the last bit of a line-search sum (src/algorithms/proxgrad.jl:143,187
# compare two long sums with a strict `<`; include/glrm_hip.h: glrm_sum_order).  `h` is the engine handle fit! keeps per model.
function hip_sum_order(h::Ptr{Cvoid}, which::Integer)
    o = Ref(CSumOrder(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0))
    check(ccall((:glrm_hip_sum_order, LIB), Cint, (Ptr{Cvoid}, Int32, Ref{CSumOrder}), h, Int32(which), o))
    o[]
end

end # module
