# TVLearningFunctionHIP.jl -- the glue of INTEGRATION.md section 1 as a file: include it from
# src/BPLDenoising.jl:31 instead of TVLearningFunctionVec.jl.  It binds libbpltv's C ABI (include/bpltv.h)
# behind the reference's own names; bilevel_learn (src/TRBox.jl) is untouched.
# NOT EXECUTED HERE: Julia is not available in the build image; the same entry points are exercised through
# bpldenoising_amd/_lib.py (ctypes) by the test suite.
# Same exports as src/TVLearningFunctionVec.jl:6
export tv_op_learning_function, denoise

const libbpltv = "libbpltv"            # on LD_LIBRARY_PATH, or an absolute path

# struct bpltv_params (include/bpltv.h) -- field order and types must match
struct BpltvParams
    rho::Cdouble; tau0::Cdouble; sigma0::Cdouble
    accel::Cint; maxiter::Cint
    delta_t::Cdouble
    check_every::Cint
    gap_tol::Cdouble
    tile_iters::Cint; use_graph::Cint
    kappa_cap::Cdouble
    refine::Cint
    reserved::NTuple{5,Cint}
end

mutable struct BpltvHandle
    ptr::Ptr{Cvoid}
    M::Int; N::Int; O::Int
    data_id::UInt                      # objectid of the dataset currently resident on the GPU
end

function bpltv_check(h::BpltvHandle, rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:bpltv_last_error, libbpltv), Cstring, (Ptr{Cvoid},), h.ptr))
    error("libbpltv error $rc: $msg")  # reference behaviour: exceptions propagate
end

function BpltvHandle(M, N, O; device = -1)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:bpltv_create, libbpltv), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Cint, Cint, Cint),
               p, M, N, O, device, 64)
    h = BpltvHandle(p[], M, N, O, 0)
    bpltv_check(h, rc)
    finalizer(x -> ccall((:bpltv_destroy, libbpltv), Cint, (Ptr{Cvoid},), x.ptr), h)
    return h
end

function default_params(; kwargs...)
    r = Ref{BpltvParams}()
    ccall((:bpltv_default_params, libbpltv), Cint, (Ref{BpltvParams},), r)
    p = r[]
    # the reference's NamedTuple keys (src/TVLearningFunctionVec.jl:33-43); others are ignored
    get_(k, d) = haskey(kwargs, k) ? kwargs[k] : d
    return BpltvParams(get_(:ρ, p.rho), get_(:τ₀, p.tau0), get_(:σ₀, p.sigma0),
                       get_(:accel, p.accel != 0) ? 1 : 0, get_(:maxiter, p.maxiter),
                       get_(:Δt, p.delta_t), p.check_every, p.gap_tol, p.tile_iters, p.use_graph,
                       p.kappa_cap, p.refine, p.reserved)
end

const _handle = Ref{Union{Nothing,BpltvHandle}}(nothing)

# One handle per dataset: bilevel_learn passes the same `ds` to every evaluation
# (src/TRBox.jl:210,227), so the images are uploaded once.
function handle_for(ū::Array{Float64,3}, f::Array{Float64,3})
    M, N, O = size(f)
    h = _handle[]
    if h === nothing || (h.M, h.N, h.O) != (M, N, O)
        h = BpltvHandle(M, N, O); _handle[] = h
    end
    id = hash((objectid(ū), objectid(f)))
    if h.data_id != id
        GC.@preserve ū f bpltv_check(h, ccall((:bpltv_set_data, libbpltv), Cint,
            (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), h.ptr, ū, f))
        h.data_id = id
    end
    return h
end

alpha_arg(x::Real) = (Float64[x], 1, 1)
alpha_arg(x::AbstractVector) = (Vector{Float64}(x), length(x), 1)
alpha_arg(x::AbstractMatrix) = (Matrix{Float64}(x), size(x, 1), size(x, 2))   # column major m x n

# src/TVLearningFunctionVec.jl:14-27
function tv_op_learning_function(x, data, Δ; Δt = 1e-6, kwargs...)
    ū, f = data[1], data[2]
    h = handle_for(ū, f)
    a, am, an = alpha_arg(x)
    u = similar(f); cost = Ref{Cdouble}(0); grad = zeros(am, an)
    p = Ref(default_params(; Δt = Δt, kwargs...))
    GC.@preserve a u grad bpltv_check(h, ccall((:bpltv_evaluate, libbpltv), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Cdouble, Ref{BpltvParams}, Ptr{Cdouble}, Ref{Cdouble}, Ptr{Cdouble}),
        h.ptr, a, am, an, Δ, p, u, cost, grad))
    # grad has the type of x: Float64 for scalar x (src/TRBox.jl:37-39,167,237)
    return u, cost[], x isa Real ? grad[1] : reshape(grad, size(x))
end

# src/TVLearningFunctionVec.jl:45-70 (op must be FwdGradientOp(); it is the only operator on this path)
function denoise(data::Array{Float64,3}, x, op::LinOp; kwargs...)
    h = handle_for(data, data)
    a, am, an = alpha_arg(x)
    u = similar(data)
    p = Ref(default_params(; kwargs...))
    GC.@preserve a u bpltv_check(h, ccall((:bpltv_denoise, libbpltv), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Ref{BpltvParams}, Ptr{Cdouble}), h.ptr, a, am, an, p, u))
    return u
end

# src/BPLDenoising.jl:41-82
TVDenoise(data, parameter; visualize = false) = denoise(data, parameter, FwdGradientOp(); maxiter = 10000)
