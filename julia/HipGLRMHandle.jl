# HipGLRMHandle.jl -- included by HipGLRM.jl: the engine handle (Omega views and A on the device) is cached per model, so warm
# starts, `cv_by_iter`'s `max_iter = 1` loop (src/cross_validate.jl:164-175) and `regularization_path` do not re-upload; new
# regularizers only replace descriptors (glrm_hip_set_regularizers).
mutable struct Entry; h::Ptr{Cvoid}; multi::Bool; hard::UInt64; soft::UInt64; end
const CACHE = IdDict{Any,Entry}()
# (a `ccall` target is a constant expression: one literal call per entry point, the choice is made around it)
function destroy(e::Entry)
    e.h == C_NULL && return
    e.multi ? ccall((:glrm_hip_multi_destroy, LIB), Cvoid, (Ptr{Cvoid},), e.h) : ccall((:glrm_hip_destroy, LIB), Cvoid, (Ptr{Cvoid},), e.h)
    e.h = C_NULL
end
"Drop the device copy of a model's data (also done by the model's finalizer).  Call it after mutating `glrm.A` in place."
hip_release!(glrm::GLRM) = (haskey(CACHE, glrm) && (destroy(CACHE[glrm]); delete!(CACHE, glrm)); glrm)
# what the device copy depends on (data, Omega, losses, placement, options) / what set_regularizers can replace
hardkey(glrm, desc, p, dense) = hash((objectid(glrm.A), size(glrm.A), glrm.k, objectid(glrm.observed_features), objectid(glrm.observed_examples),
                                      sum(length, glrm.observed_features), sum(length, glrm.observed_examples), desc[1], length(desc[2]), length(desc[3]),
                                      p.device_id, p.ngpus, p.device_ids, p.exchange, p.x_chunks, dense, p.quad_gram, p.mode))
softkey(desc) = hash((desc[2], desc[3]))

function handle(glrm::GLRM, desc, p)
    dense = dense_ok(glrm, desc, p); multi = p.ngpus > 1
    hard, soft = hardkey(glrm, desc, p, dense), softkey(desc)
    e = get(CACHE, glrm, nothing)
    if e !== nothing && e.hard == hard
        if e.soft != soft                   # scale_regularizer! / regularization_path: Omega and A stay on the device
            rx, ry = desc[2], desc[3]
            check(multi ? ccall((:glrm_hip_multi_set_regularizers, LIB), Cint, (Ptr{Cvoid}, Ptr{CReg}, Int64, Ptr{CReg}, Int64), e.h, rx, length(rx), ry, length(ry)) :
                          ccall((:glrm_hip_set_regularizers, LIB), Cint, (Ptr{Cvoid}, Ptr{CReg}, Int64, Ptr{CReg}, Int64), e.h, rx, length(rx), ry, length(ry)))
            e.soft = soft
        end
        return e.h
    end
    e === nothing ? finalizer(hip_release!, glrm) : destroy(e)
    h = create_handle(glrm, desc, p, dense)
    CACHE[glrm] = Entry(h, multi, hard, soft)
    h
end

# a single-device handle for the other entry points (julia/HipGLRMExtras.jl); nothing if the model is outside the engine
function with_handle(f, glrm::GLRM, device_id::Int=-1)
    desc = descriptors(glrm)
    desc === nothing && return nothing
    Some(f(handle(glrm, desc, HipProxGradParams(device_id=device_id, dense=false))))   # list handle: init_svd / impute / subset work on the Omega views
end
