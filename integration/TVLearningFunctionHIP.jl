# TVLearningFunctionHIP.jl -- the glue of INTEGRATION.md section 1 as a file: include it from
# src/BPLDenoising.jl:31 instead of TVLearningFunctionVec.jl.  It binds libbpltv's C ABI (include/bpltv.h)
# behind the reference's own names; bilevel_learn (src/TRBox.jl) is untouched.
# NOT EXECUTED HERE: Julia is not available in the build image; the same entry points are exercised through
# bpldenoising_amd/_lib.py (ctypes) by the test suite.
# Same exports as src/TVLearningFunctionVec.jl:6
export tv_op_learning_function, denoise, sumregs_learning_function

const libbpltv = "libbpltv"            # on LD_LIBRARY_PATH, or an absolute path

# Devices behind the handle: 1 = a single GPU (default); 0 = every visible MI355X (bpltv_create_multi shards the
# images over them and all-reduces [cost, grad...] with RCCL inside the library).  One Julia task drives all of
# them -- the structure of src/TRBox.jl:192-273 does not change.  The default stays 1 until the n > 1 RCCL path
# has run on a multi-GPU node (tests/test_gpu_multi.py holds the device-count-gated checks for that first run).
const BPLTV_NGPUS = parse(Int, get(ENV, "BPLTV_NGPUS", "1"))
# 1: totals bitwise independent of the number of GPUs (all-gather of per-image rows, added in image order)
const BPLTV_DETERMINISTIC = parse(Int, get(ENV, "BPLTV_DETERMINISTIC", "0"))
# The choices of the PDHG loop the reference does not pin (VariationalImaging.op_denoise_pdps is not in the repo):
# set these if your installed VariationalImaging starts from x0 = 0, takes the dual step first, or estimates the
# operator norm differently (include/bpltv.h: bpltv_params.init / order / opnorm; 0 = the restatement).
const BPLTV_INIT = parse(Int, get(ENV, "BPLTV_INIT", "0"))
const BPLTV_ORDER = parse(Int, get(ENV, "BPLTV_ORDER", "0"))
const BPLTV_OPNORM = parse(Float64, get(ENV, "BPLTV_OPNORM", "0"))

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
    deterministic::Cint
    reserved::NTuple{5,Cint}
    init::Cint            # 0: x0 = f; 1: x0 = 0                  (unpinned choices of op_denoise_pdps,
    order::Cint           # 0: primal step first; 1: dual first    DESIGN.md 2.3; 0 = the restatement)
    opnorm::Cdouble       # operator-norm estimate L; 0 = sqrt(8)
end

mutable struct BpltvHandle
    ptr::Ptr{Cvoid}
    M::Int; N::Int; O::Int
    # The dataset currently resident on the GPUs.  The arrays themselves are kept (not their objectid): a
    # reference held here keeps them alive, so a later dataset cannot be allocated at the same address and
    # pass for this one after a GC; identity is compared with `===`.
    ū::Union{Nothing,Array{Float64,3}}
    f::Union{Nothing,Array{Float64,3}}
    fingerprint::Tuple{Float64,Float64}   # (sum(ū), sum(f)) at upload time: catches in-place edits
end

function bpltv_check(h::BpltvHandle, rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:bpltv_last_error, libbpltv), Cstring, (Ptr{Cvoid},), h.ptr))
    error("libbpltv error $rc: $msg")  # reference behaviour: exceptions propagate
end

function BpltvHandle(M, N, O; ngpus = BPLTV_NGPUS)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:bpltv_create_multi, libbpltv), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Cint, Cint, Cint),
               p, M, N, O, ngpus, 64)
    h = BpltvHandle(p[], M, N, O, nothing, nothing, (NaN, NaN))
    finalizer(x -> ccall((:bpltv_destroy, libbpltv), Cint, (Ptr{Cvoid},), x.ptr), h)
    bpltv_check(h, rc)
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
                       p.kappa_cap, p.refine, BPLTV_DETERMINISTIC, p.reserved, BPLTV_INIT, BPLTV_ORDER, BPLTV_OPNORM)
end

const _handle = Ref{Union{Nothing,BpltvHandle}}(nothing)

# One handle per dataset: bilevel_learn passes the same `ds` to every evaluation
# (src/TRBox.jl:210,227), so the images are uploaded once.  A different array object, or the same object
# with edited content, is uploaded again.
function handle_for(ū::Array{Float64,3}, f::Array{Float64,3})
    M, N, O = size(f)
    h = _handle[]
    if h === nothing || (h.M, h.N, h.O) != (M, N, O)
        h = BpltvHandle(M, N, O); _handle[] = h
    end
    fp = (sum(ū), sum(f))
    if !(h.ū === ū && h.f === f && h.fingerprint == fp)
        GC.@preserve ū f bpltv_check(h, ccall((:bpltv_set_data, libbpltv), Cint,
            (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), h.ptr, ū, f))
        h.ū = ū; h.f = f; h.fingerprint = fp
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

# src/SumRegsLearningFunction.jl:8-36 -- x::Vector (3 weights: forward, backward, centred TV) or x::Array{T,3} (m x n x 3).
# The C ABI takes the three slices one after the other, which is exactly Julia's column-major memory of x.
function sumregs_learning_function(x::Union{AbstractVector{Float64},AbstractArray{Float64,3}}, data, Δ; Δt = 1e-3)
    ū, f = data[1], data[2]
    h = handle_for(ū, f)
    a = Array{Float64}(x)
    am, an = x isa AbstractVector ? (1, 1) : (size(x, 1), size(x, 2))
    u = similar(f); cost = Ref{Cdouble}(0); grad = zeros(size(a))
    r = Ref{BpltvParams}()
    ccall((:bpltv_sumregs_default_params, libbpltv), Cint, (Ref{BpltvParams},), r)
    d = r[]
    p = Ref(BpltvParams(d.rho, d.tau0, d.sigma0, d.accel, d.maxiter, Δt, d.check_every, d.gap_tol, d.tile_iters,
                        d.use_graph, d.kappa_cap, d.refine, BPLTV_DETERMINISTIC, d.reserved, BPLTV_INIT, BPLTV_ORDER, BPLTV_OPNORM))
    GC.@preserve a u grad bpltv_check(h, ccall((:bpltv_sumregs_evaluate, libbpltv), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Cdouble, Ref{BpltvParams}, Ptr{Cdouble}, Ref{Cdouble}, Ptr{Cdouble}),
        h.ptr, a, am, an, Δ, p, u, cost, grad))
    return u, cost[], grad
end

# generate_cost / generate_2d_cost, src/BPLDenoising.jl:92-111,136-158: cost(alpha_k) = L2CostFunction(TVDenoise(f, alpha_k), true)
# for K parameters (a Vector of scalars, or an m x n x K array of parameter matrices) as ONE bpltv_sweep call.  Over several
# GPUs (BPLTV_NGPUS) the library splits the images or -- a one-image dataset, the default num_samples = 1 of :313 -- the K
# parameter blocks over replicas of the dataset; the costs are bitwise those of one GPU.
function generate_cost_sweep(data, parameters; maxiter = 10000)
    ū, f = data[1], data[2]
    h = handle_for(ū, f)
    a = Array{Float64}(parameters)
    am, an, K = ndims(a) == 1 ? (1, 1, length(a)) : (size(a, 1), size(a, 2), size(a, 3))
    costs = zeros(K)
    p = Ref(default_params(; maxiter = maxiter))
    GC.@preserve a costs bpltv_check(h, ccall((:bpltv_sweep, libbpltv), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Cint, Ref{BpltvParams}, Ptr{Cdouble}, Ptr{Cdouble}),
        h.ptr, a, K, am, an, p, costs, C_NULL))
    return costs
end

# test / measurement aids of a handle (include/bpltv.h, bpltv_set_option), e.g. set_option(h, "sweep_split", 2)
set_option(h::BpltvHandle, name::String, value::Real) =
    bpltv_check(h, ccall((:bpltv_set_option, libbpltv), Cint, (Ptr{Cvoid}, Cstring, Cdouble), h.ptr, name, value))
