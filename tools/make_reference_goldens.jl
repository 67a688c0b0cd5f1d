# make_reference_goldens.jl -- pins oracle/nf_oracle.py to the REAL reference.
#
# The build container has no Julia, so `oracle/nf_oracle.py` restates Bijectors' PlanarLayer / RadialLayer,
# MonotonicSplines' rqs_params_from_nn / rqs_forward / rqs_inverse, Flux' Dense / leakyrelu and Optimisers' destructure
# order from their published algorithms ("parity unpinned", DESIGN.md section 5).  This script is what a maintainer WITH
# Julia runs once to close that gap: for every committed fixture tests/golden/<name>.npz it rebuilds the same flow with
# the real packages, loads the fixture's theta / xs / target parameters, and writes tests/golden/ref_<name>.npz holding
# what the reference itself computes (ys, ladj, inverse round trip, per-sample ELBO terms, loss, Zygote gradient, theta
# after one Optimisers.Adam step, loglikelihood) plus, for NSF fixtures, a known-answer vector of
# MonotonicSplines.rqs_params_from_nn on the first-applied coupling's raw conditioner output.
# `python -m pytest tests/test_reference_goldens.py` then compares the oracle (and, with -m gpu, the HIP library) with
# those files; without them the test is skipped.
#
# Usage (from the repo root, with the reference checkout's test environment -- test/Project.toml:22-29 pins
# Bijectors 0.16.2, MonotonicSplines 0.3.3, Flux 0.16.10, Optimisers 0.4.7, Distributions 0.25.129, Zygote 0.7.11 --
# plus NPZ.jl for the container format):
#   julia --project=/path/to/NormalizingFlows.jl/test -e 'using Pkg; Pkg.add("NPZ")'
#   julia --project=/path/to/NormalizingFlows.jl/test tools/make_reference_goldens.jl /path/to/NormalizingFlows.jl
#
# Inputs are read ONLY as numbers (NPZ.jl does not decode numpy unicode scalars), so the flow specifications are
# restated in CASES below; tests/test_reference_goldens.py checks that table against tests/golden/make_golden.py:CASES.
using LinearAlgebra, Random, Statistics
using NPZ
using Distributions, Bijectors, Flux, Functors, Optimisers, Zygote
using MonotonicSplines
using NormalizingFlows
using NormalizingFlows: realnvp, nsf, planarflow, radialflow, elbo_batch, _batched_elbos, loglikelihood

const REF = length(ARGS) >= 1 ? ARGS[1] : error("usage: make_reference_goldens.jl <NormalizingFlows.jl checkout>")
for f in ("banana.jl", "neal_funnel.jl", "cross.jl", "warped_gaussian.jl")
    include(joinpath(REF, "example", "targets", f))
end
const GOLDEN = joinpath(@__DIR__, "..", "tests", "golden")

# name => (kind, d, nlayers, hdims, K, B, eltype, target kind)            (tests/golden/make_golden.py:CASES)
const CASES = [
    "realnvp_d5_h32" => (:realnvp, 5, 2, [32, 32], 0, 0.0, Float32, :diaggauss),
    "realnvp_d64_h64" => (:realnvp, 64, 4, [64, 64], 0, 0.0, Float32, :diaggauss),
    "realnvp_d64_h32" => (:realnvp, 64, 1, [32, 32], 0, 0.0, Float32, :diaggauss),
    "planar_d2_banana" => (:planar, 2, 10, Int[], 0, 0.0, Float64, :banana),
    "planar_d5" => (:planar, 5, 10, Int[], 0, 0.0, Float32, :diaggauss),
    "radial_d5" => (:radial, 5, 10, Int[], 0, 0.0, Float32, :diaggauss),
    "meanfield_d4" => (:meanfield, 4, 1, Int[], 0, 0.0, Float64, :diaggauss),
    "nsf_d5_k10" => (:nsf, 5, 2, [32, 32], 10, 5.0, Float32, :diaggauss),
    "nsf_d32_k8" => (:nsf, 32, 1, [32, 32], 8, 5.0, Float32, :diaggauss),
    "nsf_d32_k8_nl4" => (:nsf, 32, 4, [32, 32], 8, 5.0, Float32, :diaggauss),
    "realnvp_d5_h32_f64" => (:realnvp, 5, 2, [32, 32], 0, 0.0, Float64, :diaggauss),
    "nsf_d5_k10_f64" => (:nsf, 5, 2, [32, 32], 10, 5.0, Float64, :diaggauss),
    "realnvp_d70_h65_33" => (:realnvp, 70, 1, [65, 33], 0, 0.0, Float32, :diaggauss),
    "realnvp_d9_3hidden" => (:realnvp, 9, 1, [24, 16, 8], 0, 0.0, Float32, :diaggauss),
    "realnvp_d64_3hidden" => (:realnvp, 64, 2, [64, 64, 64], 0, 0.0, Float32, :diaggauss),
    "realnvp_d64_1hidden" => (:realnvp, 64, 2, [64], 0, 0.0, Float32, :diaggauss),
    "realnvp_d64_h64_f64" => (:realnvp, 64, 2, [64, 64], 0, 0.0, Float64, :diaggauss),
    "nsf_d32_h64_k8" => (:nsf, 32, 1, [64, 64], 8, 3.0, Float32, :diaggauss),
    "planar_d5_funnel" => (:planar, 5, 4, Int[], 0, 0.0, Float64, :funnel),
    "radial_d2_cross" => (:radial, 2, 4, Int[], 0, 0.0, Float32, :cross),
    "planar_d2_warped" => (:planar, 2, 4, Int[], 0, 0.0, Float64, :warped),
]

@leaf MvNormal   # q0 is not trainable (test/flow.jl:10, example/demo_planar_flow.jl:23)

function build_flow(kind, d, nlayers, hdims, K, B, ::Type{T}) where {T}
    q0 = MvNormal(zeros(T, d), I)
    kind == :realnvp && return realnvp(q0, hdims, nlayers; paramtype=T)
    kind == :nsf && return nsf(q0, hdims, K, T(B), nlayers; paramtype=T)
    kind == :planar && return planarflow(q0, nlayers; paramtype=T)
    kind == :radial && return radialflow(q0, nlayers; paramtype=T)
    kind == :meanfield && return Bijectors.transformed(q0, Bijectors.Shift(zeros(T, d)) ∘ Bijectors.Scale(ones(T, d)))
    error("unknown flow kind $kind")
end

function build_target(tk, tp, d, ::Type{T}) where {T}
    tk == :diaggauss && return MvNormal(T.(tp[1, :]), Diagonal(T.(tp[2, :])))
    tk == :banana && return Banana(d, T(tp[1, 1]), T(tp[2, 1]))
    tk == :funnel && return Funnel(d, T(tp[1, 1]), T(tp[2, 1]))
    tk == :cross && return Cross(T(tp[1, 1]), T(tp[2, 1]))
    tk == :warped && return WarpedGauss(T(tp[1, 1]), T(tp[2, 1]))
    error("unknown target $tk")
end

# the first-APPLIED NSF coupling: reduce(∘, Ls) applies the last-listed layer first, and NSF_layer = af1 ∘ af2
# (src/flows/neuralspline.jl:176-183), so walk to the innermost bijector of the composition
innermost(f::ComposedFunction) = innermost(f.inner)
innermost(f) = f

function main()
    for (name, (kind, d, nlayers, hdims, K, B, T, tk)) in CASES
        z = npzread(joinpath(GOLDEN, name * ".npz"), ["theta", "xs", "target_params", "fkl_xs"])
        flow0 = build_flow(kind, d, nlayers, hdims, K, B, T)
        θ0, re = Optimisers.destructure(flow0)
        θ = T.(vec(z["theta"]))
        length(θ) == length(θ0) || error("$name: fixture has $(length(θ)) parameters, destructure gives $(length(θ0))")
        flow = re(θ)
        xs = T.(z["xs"])
        target = build_target(tk, Float64.(z["target_params"]), d, T)
        logp(y) = logpdf(target, y)

        ys, ladj = Bijectors.with_logabsdet_jacobian(flow.transform, xs)
        xr, ladj_inv = Bijectors.with_logabsdet_jacobian(Bijectors.inverse(flow.transform), ys)
        elbos = _batched_elbos(flow, logp, xs)
        loss(θ_) = -elbo_batch(re(θ_), logp, xs)
        ℓ, back = Zygote.pullback(loss, θ)
        g = back(one(ℓ))[1]
        st = Optimisers.setup(Optimisers.Adam(1f-3), θ)
        _, θ1 = Optimisers.update(st, copy(θ), g)
        fkl_xs = T.(z["fkl_xs"])
        fkl(θ_) = -loglikelihood(Random.default_rng(), re(θ_), fkl_xs)
        ℓf, backf = Zygote.pullback(fkl, θ)
        gf = backf(one(ℓf))[1]

        out = Dict{String,Any}(
            "ys" => Float64.(ys), "ladj" => Float64.(vec(ladj)), "x_roundtrip" => Float64.(xr),
            "ladj_inv" => Float64.(vec(ladj_inv)), "elbos" => Float64.(vec(elbos)), "loss" => Float64(ℓ),
            "grad" => Float64.(g), "theta_adam1" => Float64.(θ1), "loglik_of_fkl_xs" => Float64(-ℓf),
            "fkl_loss" => Float64(ℓf), "fkl_grad" => Float64.(gf), "theta_order_check" => Float64.(θ0 .* 0 .+ θ),
        )
        if kind == :nsf
            # known-answer test of MonotonicSplines.rqs_params_from_nn (call site src/flows/neuralspline.jl:65-71):
            # raw conditioner output of the first-applied coupling on its conditioner partition of xs, and the knots
            nsc = innermost(flow.transform)
            x1, x2, x3 = Bijectors.partition(nsc.mask, xs)
            raw = nsc.nn(x2)
            pX, pY, dYdX = MonotonicSplines.rqs_params_from_nn(raw, nsc.n_dims_transformed, nsc.B)
            y1, lj = MonotonicSplines.rqs_forward(x1, pX, pY, dYdX)
            out["rqs_raw"] = Float64.(raw)
            out["rqs_x1"] = Float64.(x1)
            out["rqs_pX"] = Float64.(pX)
            out["rqs_pY"] = Float64.(pY)
            out["rqs_dYdX"] = Float64.(dYdX)
            out["rqs_y1"] = Float64.(y1)
            out["rqs_logjac"] = Float64.(vec(lj))
        end
        npzwrite(joinpath(GOLDEN, "ref_" * name * ".npz"), out)
        println(rpad(name, 24), " P=", length(θ), "  loss=", ℓ, "  |g|inf=", maximum(abs, g))
    end
end

main()
