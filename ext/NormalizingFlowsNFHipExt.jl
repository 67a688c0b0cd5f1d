# NormalizingFlowsNFHipExt -- package extension binding libnfhip.so (MI355X / gfx950, C ABI of include/nfhip.h)
# into NormalizingFlows.jl.  Modelled on ext/NormalizingFlowsCUDAExt.jl:1-76 of the reference: an extension module
# that adds methods to the generic functions the ELBO / log-likelihood hot path goes through, selected by the device
# RNG type (there: CUDA.RNG; here: NFHipRNG) and by the array type (there: CuArray; here: AMDGPU.ROCArray).
# AMDGPU.jl is used for device memory and the task's HIP stream only -- no CUDA.jl, no compatibility shim.
#
# NOT EXECUTED in the build image (no Julia there).  The same ABI is exercised call for call from Python
# (normalizingflows.jl_amd/, tests/test_gpu_parity.py); every ccall signature below is the ctypes signature in
# normalizingflows.jl_amd/_lib.py:SYMBOLS.
#
# Project.toml stanza: ext/Project.toml.stanza
module NormalizingFlowsNFHipExt

using AMDGPU
using NormalizingFlows
using NormalizingFlows: Bijectors, Distributions, Optimisers, ADTypes, Random, LinearAlgebra
using NormalizingFlows: AffineCoupling, NeuralSplineCoupling
import ChainRulesCore

const libnfhip = get(ENV, "NFHIP_LIB", "libnfhip.so")

# ------------------------------------------------------------------------------------------------------------
# C structs of include/nfhip.h (isbits, same field order and padding: five Int32, NTuple{4,Int32}, Int32, Float32, Ptr)
# ------------------------------------------------------------------------------------------------------------
struct NFDesc
    kind::Int32
    dtype::Int32
    d::Int32
    nlayers::Int32
    n_hidden::Int32
    hdims::NTuple{4,Int32}
    K::Int32
    B::Float32
    score::Ptr{Cvoid}   # NF_KIND_HAMILTONIAN: pointer to the NFTarget behind LeapFrog's score, else C_NULL
    base::Ptr{Cvoid}    # pointer to an NFBase for a general MvNormal(mu, Sigma) q0, C_NULL = MvNormal(zeros, I)
    nsegments::Int32    # NF_KIND_COMPOSITE: mixed bijector families, see segments_of
    segments::Ptr{Cvoid}
end
struct NFBase            # nf_base: kind 1 = Diagonal(sigma^2) (scale = sigma), 2 = dense (scale = L, column-major)
    kind::Int32
    mu::Ptr{Cvoid}
    scale::Ptr{Cvoid}
    logdet::Float64
end
struct NFTarget
    kind::Int32
    p0::Ptr{Cvoid}
    p1::Ptr{Cvoid}
    s0::Float64
    s1::Float64
end

const NF_KIND_PLANAR, NF_KIND_RADIAL, NF_KIND_REALNVP, NF_KIND_NSF, NF_KIND_MEANFIELD = Int32(0), Int32(1), Int32(2), Int32(3), Int32(4)
const NF_TARGET_DIAGGAUSS, NF_TARGET_BANANA = Int32(0), Int32(1)
dtype_code(::Type{Float32}) = Int32(0)
dtype_code(::Type{Float64}) = Int32(1)

function check(code::Integer)
    code == 0 && return nothing
    return error(unsafe_string(ccall((:nf_strerror, libnfhip), Cstring, (Cint,), code)))
end

# ------------------------------------------------------------------------------------------------------------
# context: one nf_ctx per device, enqueueing on the current task's HIP stream
# ------------------------------------------------------------------------------------------------------------
const CONTEXTS = Dict{Int,Ptr{Cvoid}}()

function context()
    dev = AMDGPU.device_id(AMDGPU.device()) - 1          # AMDGPU.jl ids are 1-based, HIP ordinals 0-based
    stream = Ptr{Cvoid}(UInt(AMDGPU.stream().stream.handle))
    ctx = get(CONTEXTS, dev, C_NULL)
    if ctx == C_NULL
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:nf_ctx_create, libnfhip), Cint, (Cint, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), dev, stream, ref))
        ctx = CONTEXTS[dev] = ref[]
    else
        check(ccall((:nf_ctx_set_stream, libnfhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx, stream))
    end
    return ctx
end

"""
    stash_budget!(bytes)

Bound (bytes ≥ 0; 0 = off) or reset (negative) the activation stash of the RealNVP training steps on the current device:
what Zygote's tape is to `src/optimize.jl:12-14`, kept by the forward kernels for the reverse pass.  Larger batches run in
chunks through the buffer; with 0 the reverse pass recomputes (slower, no extra memory).
"""
stash_budget!(bytes::Integer) =
    check(ccall((:nf_ctx_set_stash_budget, libnfhip), Cint, (Ptr{Cvoid}, Int64), context(), Int64(bytes)))

devptr(A::ROCArray) = Ptr{Cvoid}(UInt(pointer(A)))
devptr(::Nothing) = C_NULL

# ------------------------------------------------------------------------------------------------------------
# the device RNG handle: what CUDA.RNG is to the CUDA extension.  Philox4x32-10 keyed by `seed`; every draw call
# consumes one stream id; `offset` places a shard inside a global batch (multi-GPU).
# ------------------------------------------------------------------------------------------------------------
mutable struct NFHipRNG <: Random.AbstractRNG
    seed::UInt64
    stream::UInt32
    offset::UInt64
end
NFHipRNG(seed::Integer=0; offset::Integer=0) = NFHipRNG(UInt64(seed), UInt32(0), UInt64(offset))
next_stream!(rng::NFHipRNG) = (s = rng.stream; rng.stream += UInt32(1); s)

# ------------------------------------------------------------------------------------------------------------
# desc_of(flow): the static description the library needs, read off flow.transform.
# create_flow composes reduce(∘, Ls) (src/flows/utils.jl:23-26): a left-nested ComposedFunction tree whose leaves,
# left to right, are L1 … Ln -- the order Optimisers.destructure walks (outer before inner), which is the order of
# theta the library expects.
# ------------------------------------------------------------------------------------------------------------
leaves(f::ComposedFunction) = (leaves(f.outer)..., leaves(f.inner)...)
leaves(f) = (f,)

hidden_dims(chain) = Int32[size(l.weight, 1) for l in chain.layers[1:(end - 1)]]

function pad4(h::Vector{Int32})
    length(h) <= 4 || error("nfhip: at most 4 hidden layers")
    return ntuple(i -> i <= length(h) ? h[i] : Int32(0), 4)
end

function desc_of(flow::Bijectors.TransformedDistribution)
    T = eltype(flow.dist)
    d = Int32(length(flow.dist))
    Ls = leaves(flow.transform)
    L1 = first(Ls)
    nohid = ntuple(_ -> Int32(0), 4)
    if all(l -> l isa Bijectors.PlanarLayer, Ls)
        return NFDesc(NF_KIND_PLANAR, dtype_code(T), d, Int32(length(Ls)), 0, nohid, 0, 0.0f0, C_NULL, C_NULL, Int32(0), C_NULL)
    elseif all(l -> l isa Bijectors.RadialLayer, Ls)
        return NFDesc(NF_KIND_RADIAL, dtype_code(T), d, Int32(length(Ls)), 0, nohid, 0, 0.0f0, C_NULL, C_NULL, Int32(0), C_NULL)
    elseif length(Ls) == 2 && Ls[1] isa Bijectors.Shift && Ls[2] isa Bijectors.Scale
        return NFDesc(NF_KIND_MEANFIELD, dtype_code(T), d, Int32(1), 0, nohid, 0, 0.0f0, C_NULL, C_NULL, Int32(0), C_NULL)
    elseif all(l -> l isa AffineCoupling, Ls)
        iseven(length(Ls)) || error("nfhip: realnvp flows are built from RealNVP_layer pairs (src/flows/realnvp.jl:132-145)")
        h = hidden_dims(L1.s)
        return NFDesc(NF_KIND_REALNVP, dtype_code(T), d, Int32(length(Ls) ÷ 2), Int32(length(h)), pad4(h), 0, 0.0f0, C_NULL, C_NULL, Int32(0), C_NULL)
    elseif all(l -> l isa NeuralSplineCoupling, Ls)
        iseven(length(Ls)) || error("nfhip: nsf flows are built from NSF_layer pairs (src/flows/neuralspline.jl:169-184)")
        h = hidden_dims(L1.nn)
        return NFDesc(NF_KIND_NSF, dtype_code(T), d, Int32(length(Ls) ÷ 2), Int32(length(h)), pad4(h), Int32(L1.K), Float32(L1.B), C_NULL, C_NULL, Int32(0), C_NULL)
    end
    return composite_desc(Ls, T, d)   # mixed families: create_flow((L1, …, Ln), q0), src/flows/utils.jl:23-26
end

# Mixed bijector families: maximal runs of one family become the segments of an NF_KIND_COMPOSITE descriptor (flat
# order, first = outermost).  The segment array must outlive the descriptor: SEGMENT_ROOTS owns ONE array per distinct
# segment list (keyed by the segments' own bits -- desc_of runs on every rand / nfhip(flow) call, and a push-only list
# would grow without bound in a sampling loop).
const NF_KIND_COMPOSITE = Int32(6)
const SEGMENT_ROOTS = Dict{Vector{NFDesc},Vector{NFDesc}}()
family(l) = l isa Bijectors.PlanarLayer ? :planar : l isa Bijectors.RadialLayer ? :radial : l isa AffineCoupling ? :realnvp :
            l isa NeuralSplineCoupling ? :nsf : error("nfhip: no device kernels for a $(typeof(l)) layer")
function composite_desc(Ls, ::Type{T}, d::Int32) where {T}
    segs = NFDesc[]
    i = 1
    while i <= length(Ls)
        j = i
        while j < length(Ls) && family(Ls[j + 1]) == family(Ls[i]); j += 1; end
        run = Ls[i:j]
        fake = Bijectors.transformed(Distributions.MvNormal(zeros(T, d), LinearAlgebra.I), reduce(∘, run))
        push!(segs, desc_of(fake))          # a single-family run: one of the branches above
        i = j + 1
    end
    segs = get!(SEGMENT_ROOTS, segs, segs)   # the array whose pointer goes into the descriptor stays rooted here
    nohid = ntuple(_ -> Int32(0), 4)
    return NFDesc(NF_KIND_COMPOSITE, dtype_code(T), d, Int32(1), 0, nohid, 0, 0.0f0, C_NULL, C_NULL,
                  Int32(length(segs)), Ptr{Cvoid}(pointer(segs)))
end

# theta: Optimisers.destructure(flow) as is, on the device (src/NormalizingFlows.jl:67)
function theta_of(flow)
    θ, re = Optimisers.destructure(flow)
    θd = θ isa ROCArray ? θ : ROCArray(θ)
    desc = desc_of(flow)
    P = ccall((:nf_param_count, libnfhip), Int64, (Ref{NFDesc},), desc)
    P == length(θd) || error("nfhip: flow has $(length(θd)) parameters, descriptor says $P")
    return θd, re, desc
end

# ------------------------------------------------------------------------------------------------------------
# built-in targets: callable like any `logp`, and recognised by the fused ELBO entry points
# ------------------------------------------------------------------------------------------------------------
abstract type NFHipTarget end
struct DiagGaussTarget{V<:ROCVector} <: NFHipTarget   # logpdf(MvNormal(μ, Diagonal(σ²)), z), test/flow.jl:43-46
    μ::V
    σ²::V
end
struct BananaTarget <: NFHipTarget                    # Banana(d, b, var), example/targets/banana.jl:58-83
    d::Int
    b::Float64
    var::Float64
end
c_target(t::DiagGaussTarget) = NFTarget(NF_TARGET_DIAGGAUSS, devptr(t.μ), devptr(t.σ²), 0.0, 0.0)
c_target(t::BananaTarget) = NFTarget(NF_TARGET_BANANA, C_NULL, C_NULL, t.b, t.var)
function check_target(t::DiagGaussTarget, ::Type{T}, d) where {T}
    (eltype(t.μ) === T && length(t.μ) == d) || error("nfhip: target must be a length-$d ROCVector{$T} pair")
end
check_target(t::BananaTarget, ::Type, d) = t.d == d || error("nfhip: Banana dimension mismatch")

function (t::NFHipTarget)(ys::ROCMatrix{T}) where {T}
    d, N = size(ys)
    check_target(t, T, d)
    out = ROCVector{T}(undef, N)
    check(ccall((:nf_target_logp, libnfhip), Cint,
                (Ptr{Cvoid}, Int32, Ref{NFTarget}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                context(), dtype_code(T), c_target(t), d, N, devptr(ys), devptr(out), C_NULL))
    return out
end

# ------------------------------------------------------------------------------------------------------------
# a4 + a5: _device_specific_rand  (src/NormalizingFlows.jl:94-127; CUDA version ext/NormalizingFlowsCUDAExt.jl:7-48)
# ------------------------------------------------------------------------------------------------------------
function is_standard_normal(s::Distributions.MvNormal)
    return all(iszero, s.μ) && s.Σ isa Union{Distributions.PDMats.ScalMat,Distributions.PDMats.PDiagMat} &&
           all(isone, LinearAlgebra.diag(s.Σ))
end

function NormalizingFlows._device_specific_rand(rng::NFHipRNG, s::Distributions.MvNormal, n::Int)
    T = float(eltype(s))
    d = length(s)
    x = ROCMatrix{T}(undef, d, n)
    bptr, keep = base_of(s, T)     # general MvNormal: μ + L ε on the device, as ext/NormalizingFlowsCUDAExt.jl:43-48 does
    GC.@preserve keep check(ccall((:nf_base_rand, libnfhip), Cint,
                (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int32, Int64, UInt64, UInt64, UInt32, Ptr{Cvoid}, Ptr{Cvoid}),
                context(), dtype_code(T), bptr, d, n, rng.seed, rng.offset, next_stream!(rng), devptr(x), C_NULL))
    return x
end
NormalizingFlows._device_specific_rand(rng::NFHipRNG, s::Distributions.MvNormal) = vec(NormalizingFlows._device_specific_rand(rng, s, 1))

# rand(rng, flow, n): batched and fused (the CUDA extension maps the transform over columns, :61-74)
function NormalizingFlows._device_specific_rand(rng::NFHipRNG, td::Bijectors.TransformedDistribution, n::Int)
    θ, _, desc = theta_of(td)
    T = eltype(θ)
    y = ROCMatrix{T}(undef, Int(desc.d), n)
    check(ccall((:nf_flow_rand, libnfhip), Cint,
                (Ptr{Cvoid}, Ref{NFDesc}, Ptr{Cvoid}, Int64, UInt64, UInt64, UInt32, Ptr{Cvoid}),
                context(), desc, devptr(θ), n, rng.seed, rng.offset, next_stream!(rng), devptr(y)))
    return y
end
NormalizingFlows._device_specific_rand(rng::NFHipRNG, td::Bijectors.TransformedDistribution) = vec(NormalizingFlows._device_specific_rand(rng, td, 1))

# ------------------------------------------------------------------------------------------------------------
# a6 / a7 / a10 / a12 / a13: with_logabsdet_jacobian on device arrays.
# NFHipTransform carries (theta, desc) so the composed transform is ONE library call instead of a recursion over
# ComposedFunction.  `nfhip(flow)` builds the device flow; flow.transform then is an NFHipTransform.
# ------------------------------------------------------------------------------------------------------------
struct NFHipTransform{V<:ROCVector,R} <: Bijectors.Bijector
    θ::V
    desc::NFDesc
    re::R            # Optimisers restructure: re(θ) gives back the reference-side flow
    inverted::Bool
    keep::Any        # roots what desc points into (the NFBase Ref and its device arrays), or nothing
end
Bijectors.inverse(t::NFHipTransform) = NFHipTransform(t.θ, t.desc, t.re, !t.inverted, t.keep)

# a general MvNormal(μ, Σ) q0 as an nf_base: Σ = L L', scale = σ (diagonal) or L (dense, column-major as Julia stores it)
function base_of(q0::Distributions.MvNormal, ::Type{T}) where {T}
    is_standard_normal(q0) && return C_NULL, nothing
    μ = ROCVector{T}(q0.μ)
    if q0.Σ isa Union{Distributions.PDMats.ScalMat,Distributions.PDMats.PDiagMat}
        σ = ROCVector{T}(sqrt.(LinearAlgebra.diag(q0.Σ)))
        ref = Ref(NFBase(Int32(1), devptr(μ), devptr(σ), Float64(sum(log, Array(σ)))))
        return Base.unsafe_convert(Ptr{Cvoid}, ref), (ref, μ, σ)
    end
    Lh = Matrix{T}(LinearAlgebra.cholesky(Matrix(q0.Σ)).L)
    L = ROCMatrix{T}(Lh)
    ref = Ref(NFBase(Int32(2), devptr(μ), devptr(L), Float64(sum(log, LinearAlgebra.diag(Lh)))))
    return Base.unsafe_convert(Ptr{Cvoid}, ref), (ref, μ, L)
end

with_base(d::NFDesc, base::Ptr{Cvoid}) =
    NFDesc(d.kind, d.dtype, d.d, d.nlayers, d.n_hidden, d.hdims, d.K, d.B, d.score, base, d.nsegments, d.segments)

"nfhip(flow): the same flow with parameters on the device and a library-backed transform"
function nfhip(flow::Bijectors.TransformedDistribution)
    θ, re, desc = theta_of(flow)
    bptr, keep = base_of(flow.dist, eltype(θ))
    return Bijectors.transformed(flow.dist, NFHipTransform(θ, with_base(desc, bptr), re, false, keep))
end

function apply(t::NFHipTransform, x::ROCMatrix{T}) where {T}
    d, N = size(x)
    d == t.desc.d || throw(DimensionMismatch("flow has d=$(t.desc.d), input has $d"))
    y = similar(x)
    ladj = ROCVector{T}(undef, N)
    code = t.inverted ?
        ccall((:nf_flow_inv, libnfhip), Cint, (Ptr{Cvoid}, Ref{NFDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
              context(), t.desc, devptr(t.θ), devptr(x), N, devptr(y), devptr(ladj)) :
        ccall((:nf_flow_fwd, libnfhip), Cint, (Ptr{Cvoid}, Ref{NFDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
              context(), t.desc, devptr(t.θ), devptr(x), N, devptr(y), devptr(ladj))
    check(code)
    return y, ladj
end

Bijectors.with_logabsdet_jacobian(t::NFHipTransform, x::ROCMatrix) = apply(t, x)
function Bijectors.with_logabsdet_jacobian(t::NFHipTransform, x::ROCVector)      # vector = one column; scalar logdet
    y, l = apply(t, reshape(x, :, 1))                                             # (src/flows/realnvp.jl:69-75)
    return vec(y), AMDGPU.@allowscalar l[1]
end
Bijectors.transform(t::NFHipTransform, x::ROCVecOrMat) = first(Bijectors.with_logabsdet_jacobian(t, x))
(t::NFHipTransform)(x::ROCVecOrMat) = Bijectors.transform(t, x)

# reverse-mode rule: the mechanism MonotonicSplines uses for its kernels (test/ad.jl:126-127), so AutoZygote works
# for an ARBITRARY logp closure: library forward that keeps its tape (nf_flow_fwd_keep), the closure's own pullback,
# library pullback from that tape (nf_flow_bwd_kept).  The tape is a ROCArray owned by the pullback closure -- the
# device form of the Zygote tape the reference differentiates (src/optimize.jl:12-14 on src/objectives/elbo.jl:65-70);
# the pullback leaves it intact, so calling it twice (jacobians) is fine.
function ChainRulesCore.rrule(::typeof(Bijectors.with_logabsdet_jacobian), t::NFHipTransform, x::ROCMatrix{T}) where {T}
    t.inverted && error("nfhip: the pullback of the inverse chain is exposed through loglikelihood training (AutoNFHip)")
    d, N = size(x)
    d == t.desc.d || throw(DimensionMismatch("flow has d=$(t.desc.d), input has $d"))
    nbytes = ccall((:nf_tape_bytes, libnfhip), Int64, (Ptr{Cvoid}, Ref{NFDesc}, Int64), context(), t.desc, N)
    nbytes < 0 && check(Cint(nbytes))
    tape = ROCVector{UInt8}(undef, nbytes)                       # AMDGPU.jl allocations are 256-byte aligned
    y = similar(x)
    ladj = ROCVector{T}(undef, N)
    check(ccall((:nf_flow_fwd_keep, libnfhip), Cint,
                (Ptr{Cvoid}, Ref{NFDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                context(), t.desc, devptr(t.θ), devptr(x), N, devptr(y), devptr(ladj), devptr(tape), nbytes))
    function pullback(Δ)
        ȳ, l̄ = ChainRulesCore.unthunk(Δ[1]), ChainRulesCore.unthunk(Δ[2])
        ȳ = ȳ isa ChainRulesCore.AbstractZero ? AMDGPU.zeros(T, size(y)) : ROCMatrix{T}(ȳ)
        l̄ = l̄ isa ChainRulesCore.AbstractZero ? AMDGPU.zeros(T, length(ladj)) : ROCVector{T}(l̄)
        x̄ = similar(x)
        ḡ = similar(t.θ)
        check(ccall((:nf_flow_bwd_kept, libnfhip), Cint,
                    (Ptr{Cvoid}, Ref{NFDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                    context(), t.desc, devptr(t.θ), devptr(tape), nbytes, devptr(ȳ), devptr(l̄), N, devptr(x̄), devptr(ḡ)))
        t̄ = ChainRulesCore.Tangent{typeof(t)}(; θ=ḡ)     # tangent_of: the only differentiable field is θ
        return ChainRulesCore.NoTangent(), t̄, x̄
    end
    return (y, ladj), pullback
end

# logpdf(q0, xs) on the device for the standard normal and for a general MvNormal(μ, Σ) (src/objectives/elbo.jl:6,68)
function Distributions.logpdf(s::Distributions.MvNormal, xs::ROCMatrix{T}) where {T}
    out = ROCVector{T}(undef, size(xs, 2))
    bptr, keep = base_of(s, T)
    GC.@preserve keep check(ccall((:nf_base_logpdf_general, libnfhip), Cint,
                (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                context(), dtype_code(T), bptr, size(xs, 1), size(xs, 2), devptr(xs), devptr(out)))
    return out
end

# ------------------------------------------------------------------------------------------------------------
# a1-a3, a16: objectives with a device RNG and a device flow.  Built-in targets take the fused entry points; any
# other `logp` falls through to the reference's own generic code (src/objectives/elbo.jl:65-97), which then runs on
# the methods above.
# ------------------------------------------------------------------------------------------------------------
const DeviceFlow = Bijectors.TransformedDistribution{<:Any,<:NFHipTransform}

function NormalizingFlows.elbo_batch(rng::NFHipRNG, flow::DeviceFlow, logp::NFHipTarget, n::Int)
    t = flow.transform
    check_target(logp, eltype(t.θ), t.desc.d)
    val = Ref{Float64}(0.0)
    check(ccall((:nf_elbo_batch_rng, libnfhip), Cint,
                (Ptr{Cvoid}, Ref{NFDesc}, Ref{NFTarget}, Ptr{Cvoid}, Int64, UInt64, UInt64, UInt32, Ptr{Float64}),
                context(), t.desc, c_target(logp), devptr(t.θ), n, rng.seed, rng.offset, next_stream!(rng), val))
    return eltype(t.θ)(val[])
end
function NormalizingFlows.elbo_batch(flow::DeviceFlow, logp::NFHipTarget, xs::ROCMatrix)
    t = flow.transform
    val = Ref{Float64}(0.0)
    check(ccall((:nf_elbo_batch, libnfhip), Cint,
                (Ptr{Cvoid}, Ref{NFDesc}, Ref{NFTarget}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Float64}),
                context(), t.desc, c_target(logp), devptr(t.θ), devptr(xs), size(xs, 2), C_NULL, val))
    return eltype(t.θ)(val[])
end
NormalizingFlows.elbo(rng::NFHipRNG, flow::DeviceFlow, logp::NFHipTarget, n::Int) = NormalizingFlows.elbo_batch(rng, flow, logp, n)

function NormalizingFlows.loglikelihood(::Random.AbstractRNG, flow::DeviceFlow, xs::ROCMatrix)
    t = flow.transform
    val = Ref{Float64}(0.0)
    check(ccall((:nf_loglikelihood, libnfhip), Cint,
                (Ptr{Cvoid}, Ref{NFDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Float64}),
                context(), t.desc, devptr(t.θ), devptr(xs), size(xs, 2), C_NULL, val))
    return eltype(t.θ)(val[])
end

# ------------------------------------------------------------------------------------------------------------
# a15: the training step.  AutoNFHip is an ADTypes backend whose "differentiation" is the library's hand-derived
# reverse pass; it plugs into the reference's own loop through the two hooks of src/optimize.jl:8-14.
# train_flow dispatches on the device RNG, exactly as the CUDA extension's methods dispatch on CUDA.RNG.
# ------------------------------------------------------------------------------------------------------------
struct AutoNFHip{V,A} <: ADTypes.AbstractADType
    vo::V                   # elbo, elbo_batch or loglikelihood
    desc::NFDesc
    args::A                 # (logp, n) for the ELBO; (xs,) for loglikelihood
    n_global::Int           # samples over all ranks (== the local n on one GPU)
    comm::Bool              # all-reduce [grad ; loss] over the context's RCCL communicator
end

NormalizingFlows._prepare_gradient(loss, ::AutoNFHip, θ, args...) = nothing

function NormalizingFlows._value_and_gradient(loss, prep, ad::AutoNFHip, θ::ROCVector{T}, rng::NFHipRNG, args...) where {T}
    P = length(θ)
    out = ROCVector{T}(undef, P + 1)                         # [grad ; loss] packed: ONE all-reduce when sharded
    ctx = context()
    if ad.vo === NormalizingFlows.loglikelihood
        xs = ad.args[1]::ROCMatrix{T}
        check(ccall((:nf_loglikelihood_value_and_grad, libnfhip), Cint,
                    (Ptr{Cvoid}, Ref{NFDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}),
                    ctx, ad.desc, devptr(θ), devptr(xs), size(xs, 2), ad.n_global, devptr(out)))
    else
        logp, n = ad.args
        logp isa NFHipTarget || return generic_value_and_gradient(loss, θ, rng, args...)
        check(ccall((:nf_elbo_value_and_grad, libnfhip), Cint,
                    (Ptr{Cvoid}, Ref{NFDesc}, Ref{NFTarget}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, UInt64, UInt64, UInt32, Ptr{Cvoid}),
                    ctx, ad.desc, c_target(logp), devptr(θ), C_NULL, n, ad.n_global, rng.seed, rng.offset, next_stream!(rng), devptr(out)))
    end
    if ad.comm                                               # sum over ranks, in place, on the context stream
        check(ccall((:nf_allreduce_grad_loss, libnfhip), Cint, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int64), ctx, dtype_code(T), devptr(out), P + 1))
    end
    ls = AMDGPU.@allowscalar out[P + 1]
    return ls, view(out, 1:P)
end

# arbitrary `logp` closure: reverse-mode AD of the reference's own loss through the rrule above
function generic_value_and_gradient(loss, θ, rng, args...)
    Zygote = Base.require(Base.PkgId(Base.UUID("e88e6eb3-aa80-5325-afca-941959d7151f"), "Zygote"))
    ls, back = Zygote.pullback(th -> loss(th, rng, args...), θ)
    return ls, first(back(one(ls)))
end

function NormalizingFlows.train_flow(rng::NFHipRNG, vo, flow::Bijectors.TransformedDistribution, args...;
                                     max_iters::Int=1000, optimiser::Optimisers.AbstractRule=Optimisers.ADAM(),
                                     ADbackend=nothing, n_global::Int=0, allreduce::Bool=false, kwargs...)
    dflow = flow isa DeviceFlow ? flow : nfhip(flow)
    t = dflow.transform
    θ = copy(t.θ)
    re_dev = th -> Bijectors.transformed(dflow.dist, NFHipTransform(th, t.desc, t.re, false, t.keep))   # t.desc is static
    loss(th, rng_, args_...) = -vo(rng_, re_dev(th), args_...)
    nloc = vo === NormalizingFlows.loglikelihood ? size(args[1], 2) : args[2]
    ad = AutoNFHip(vo, t.desc, args, n_global > 0 ? n_global : nloc, allreduce)
    θ_trained, opt_stats, st = NormalizingFlows.optimize(ad, loss, θ, re_dev, rng, args...;
                                                        max_iters=max_iters, optimiser=optimiser, kwargs...)
    return re_dev(θ_trained), opt_stats, st
end

# ------------------------------------------------------------------------------------------------------------
# (e) multi-GPU: sample shards + ONE all-reduce of [grad ; loss] per step (include/nfhip.h, "multi-GPU").
# One Julia process per GPU (as with MPI.jl launchers): rank 0 creates the id, the launcher's own channel ships the
# 128 bytes, every rank joins; then train_flow(NFHipRNG(seed; offset = rank * n), ...; n_global = nranks * n,
# allreduce = true).  Single process, G devices: nf_comm_init_all over the G contexts + nf_allreduce_grad_loss_all.
# ------------------------------------------------------------------------------------------------------------
function comm_unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:nf_comm_get_unique_id, libnfhip), Cint, (Ptr{UInt8},), id))
    return id
end
comm_init_rank(id::Vector{UInt8}, nranks::Integer, rank::Integer) =
    check(ccall((:nf_comm_init_rank, libnfhip), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32), context(), id, nranks, rank))
comm_destroy() = check(ccall((:nf_comm_destroy, libnfhip), Cint, (Ptr{Cvoid},), context()))
# ABI v4: the step's one logical all-reduce may travel in buckets of whole couplings on a second stream (large gradients only,
# 4 MiB by default); bytes < 0 automatic, 0 one message.  comm_bucket_count(flow): messages nf_elbo_step issues per step.
comm_bucket_bytes!(bytes::Integer) =
    check(ccall((:nf_ctx_set_comm_bucket_bytes, libnfhip), Cint, (Ptr{Cvoid}, Int64), context(), bytes))
comm_bucket_count(flow::DeviceFlow) =
    ccall((:nf_comm_bucket_count, libnfhip), Cint, (Ptr{Cvoid}, Ref{NFDesc}), context(), flow.transform.desc)

# ------------------------------------------------------------------------------------------------------------
# ABI v4: the whole iteration of src/optimize.jl:85-99 in ONE call (nf_elbo_step: draws, forward, reverse pass, [all-reduce,]
# Adam, norm(g)) -- what bench.py times and what the Python mirror's train_flow runs for built-in targets.  The loop owns θ
# between iterations, so it may opt in to the library's packed-weight cache (nf_ctx_set_weight_cache) and must opt out when
# it returns; by default every nf_elbo_step packs from θ, so a θ edited or re-allocated behind the library's back is safe.
# ------------------------------------------------------------------------------------------------------------
weight_cache!(on::Bool) = check(ccall((:nf_ctx_set_weight_cache, libnfhip), Cint, (Ptr{Cvoid}, Int32), context(), on ? 1 : 0))
weights_changed!() = check(ccall((:nf_ctx_weights_changed, libnfhip), Cint, (Ptr{Cvoid},), context()))

function elbo_step!(θ::ROCVector{T}, m::ROCVector{T}, v::ROCVector{T}, desc::NFDesc, logp::NFHipTarget, n::Integer, rng::NFHipRNG,
                    step::Integer, rule::Optimisers.Adam) where {T}
    loss, gnorm = Ref{Cdouble}(0), Ref{Cdouble}(0)
    check(ccall((:nf_elbo_step, libnfhip), Cint,
                (Ptr{Cvoid}, Ref{NFDesc}, Ref{NFTarget}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, UInt64, UInt32,
                 Cdouble, Cdouble, Cdouble, Cdouble, Ref{Cdouble}, Ref{Cdouble}),
                context(), desc, c_target(logp), devptr(θ), devptr(m), devptr(v), n, rng.seed, step,
                rule.eta, rule.beta[1], rule.beta[2], rule.epsilon, loss, gnorm))
    return loss[], gnorm[]
end

# train_flow_fused(rng, flow, logp, n; max_iters, optimiser::Adam, state): the fused loop for a built-in target (same numbers as
# train_flow over AutoNFHip; tests/test_gpu_tape.py checks that equality through the Python mirror of this function,
# objectives._optimize_fused / _fused_steps_apply).
# nf_elbo_step uses ONE index for the Philox stream of the draws and for Adam's step count t - 1, and draws from sample offset
# 0 of a single-rank context.  So the fused loop applies only when the two are in step -- a fresh rng (stream 0) with a fresh
# state, or an rng and a `state` continued together (rng.stream == state.t) -- with rng.offset == 0 and no communicator on the
# context; anything else is refused here and belongs to train_flow over AutoNFHip (ADVICE r4: an rng that had already drawn
# started Adam at t = stream + 1 with m = v = 0, and the offset was dropped).  A non-finite loss is recorded, not thrown, as
# the reference's loop would record it (src/optimize.jl:85-99); the returned state carries the real step count.
const NF_ERR_NONFINITE = Cint(-4)  # include/nfhip.h
comm_size() = ccall((:nf_comm_size, libnfhip), Cint, (Ptr{Cvoid},), context())
function fused_steps_apply(rng::NFHipRNG, state)
    t0 = state === nothing ? 0 : Int(state.t)
    return rng.offset == 0 && Int(rng.stream) == t0 && comm_size() <= 1
end
function train_flow_fused(rng::NFHipRNG, flow::Bijectors.TransformedDistribution, logp::NFHipTarget, n::Integer;
                          max_iters::Int=1000, optimiser::Optimisers.Adam=Optimisers.Adam(), state=nothing)
    fused_steps_apply(rng, state) ||
        error("nfhip: train_flow_fused needs rng.stream == state.t (0 for a fresh run), rng.offset == 0 and a single-rank context; use train_flow(rng, elbo_batch, flow, logp, n; ADbackend = AutoNFHip(...)) otherwise")
    dflow = flow isa DeviceFlow ? flow : nfhip(flow)
    t = dflow.transform
    θ = copy(t.θ)
    m = state === nothing ? zero(θ) : copy(state.m)
    v = state === nothing ? zero(θ) : copy(state.v)
    steps = state === nothing ? 0 : Int(state.t)
    stats = NamedTuple[]
    weight_cache!(true)
    try
        for i in 1:max_iters
            loss, gnorm = Ref{Cdouble}(0), Ref{Cdouble}(0)
            code = ccall((:nf_elbo_step, libnfhip), Cint,
                         (Ptr{Cvoid}, Ref{NFDesc}, Ref{NFTarget}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, UInt64, UInt32,
                          Cdouble, Cdouble, Cdouble, Cdouble, Ref{Cdouble}, Ref{Cdouble}),
                         context(), t.desc, c_target(logp), devptr(θ), devptr(m), devptr(v), n, rng.seed, next_stream!(rng),
                         optimiser.eta, optimiser.beta[1], optimiser.beta[2], optimiser.epsilon, loss, gnorm)
            code == NF_ERR_NONFINITE || check(code)   # a non-finite loss is recorded
            steps += 1
            push!(stats, (iteration=i, loss=loss[], gradient_norm=gnorm[]))
        end
    finally
        weight_cache!(false)
    end
    return Bijectors.transformed(dflow.dist, NFHipTransform(θ, t.desc, t.re, false, t.keep)), stats, (m=m, v=v, t=steps)
end

end # module
