"""Generates the committed golden fixtures tests/golden/*.npz from the CPU oracle
(oracle/nf_oracle.py, float64).  Run from the repo root:  python tests/golden/make_golden.py

The reference itself (Julia) cannot run in the build container and ships no golden vectors
(SURVEY.md 8c), so these vectors pin the HIP path to the oracle, and the oracle is pinned by
tests/test_oracle.py.  Each fixture holds: flow spec, theta, base draws xs, ys, ladj, per-sample
elbos, loss = -elbo_batch, grad, theta after one Adam(1e-3) step, and the forward-KL training pair
(fkl_loss, fkl_grad) = value and gradient of -loglikelihood(flow, fkl_xs) on the data fkl_xs = ys rounded to
the storage dtype.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import nf_oracle as o  # noqa: E402

CASES = {
    # name: (spec, N, target kind, storage dtype of inputs)
    "realnvp_d5_h32": (o.FlowSpec("realnvp", 5, 2, (32, 32)), 10, "diaggauss", np.float32),  # test/flow.jl:2-38
    "realnvp_d64_h64": (o.FlowSpec("realnvp", 64, 4, (64, 64)), 96, "diaggauss", np.float32),  # cfg 2 shape
    "realnvp_d64_h32": (o.FlowSpec("realnvp", 64, 1, (32, 32)), 33, "diaggauss", np.float32),
    "planar_d2_banana": (o.FlowSpec("planar", 2, 10), 64, "banana", np.float64),  # cfg 1
    "planar_d5": (o.FlowSpec("planar", 5, 10), 10, "diaggauss", np.float32),  # test/flow.jl:137
    "radial_d5": (o.FlowSpec("radial", 5, 10), 10, "diaggauss", np.float32),  # test/flow.jl:203
    "meanfield_d4": (o.FlowSpec("meanfield", 4, 1), 16, "diaggauss", np.float64),
    "nsf_d5_k10": (o.FlowSpec("nsf", 5, 2, (32, 32), K=10, B=5.0), 10, "diaggauss", np.float32),  # test/flow.jl:68
    "nsf_d32_k8": (o.FlowSpec("nsf", 32, 1, (32, 32), K=8, B=5.0), 40, "diaggauss", np.float32),  # cfg 3 shape
    "nsf_d32_k8_nl4": (o.FlowSpec("nsf", 32, 4, (32, 32), K=8, B=5.0), 40, "diaggauss", np.float32),  # cfg 3: all 8 couplings
    # Float64 coupling flows (test/flow.jl:7,72 run both element types) -> general kernels
    "realnvp_d5_h32_f64": (o.FlowSpec("realnvp", 5, 2, (32, 32)), 10, "diaggauss", np.float64),
    "nsf_d5_k10_f64": (o.FlowSpec("nsf", 5, 2, (32, 32), K=10, B=5.0), 10, "diaggauss", np.float64),
    # weight-streaming kernels at a small padded shape (64-128-128-64 geometry), and an odd hidden-layer count
    "realnvp_d70_h65_33": (o.FlowSpec("realnvp", 70, 1, (65, 33)), 33, "diaggauss", np.float32),
    "realnvp_d9_3hidden": (o.FlowSpec("realnvp", 9, 1, (24, 16, 8)), 20, "diaggauss", np.float32),
    # the shapes of the depth-generic resident RealNVP kernels (nf_deep.hip), of the Float64 matrix-instruction path
    # (nf_g64m.h) and the reference's documented nsf(q0, [64, 64], 8, 3.0, ...) example (src/flows/neuralspline.jl:215)
    "realnvp_d64_3hidden": (o.FlowSpec("realnvp", 64, 2, (64, 64, 64)), 40, "diaggauss", np.float32),
    "realnvp_d64_1hidden": (o.FlowSpec("realnvp", 64, 2, (64,)), 40, "diaggauss", np.float32),
    "realnvp_d64_h64_f64": (o.FlowSpec("realnvp", 64, 2, (64, 64)), 40, "diaggauss", np.float64),
    "nsf_d32_h64_k8": (o.FlowSpec("nsf", 32, 1, (64, 64), K=8, B=3.0), 40, "diaggauss", np.float32),
    # demo targets (example/targets/*.jl)
    "planar_d5_funnel": (o.FlowSpec("planar", 5, 4), 32, "funnel", np.float64),
    "radial_d2_cross": (o.FlowSpec("radial", 2, 4), 32, "cross", np.float32),
    "planar_d2_warped": (o.FlowSpec("planar", 2, 4), 32, "warped", np.float64),
}


def main():
    only = sys.argv[1:]  # optional: regenerate only the named fixtures
    for name, (spec, n, tkind, dt) in CASES.items():
        if only and name not in only:
            continue
        rng = np.random.default_rng(abs(hash(name)) % (2**31) if False else sum(map(ord, name)))
        theta = o.init_params(spec, rng)
        if spec.kind in ("realnvp", "nsf"):
            theta = theta + 0.05 * rng.standard_normal(theta.shape)  # non-zero biases
        if spec.kind == "meanfield":
            theta = theta + 0.3 * rng.standard_normal(theta.shape)
        theta = theta.astype(dt).astype(np.float64)  # values exactly representable in the storage dtype
        xs = o.base_sample(spec.d, n, seed=123).astype(dt).astype(np.float64)
        if tkind == "diaggauss":
            mu = rng.standard_normal(spec.d).astype(dt).astype(np.float64)
            var = (rng.uniform(size=spec.d) + 0.5).astype(dt).astype(np.float64)
            tgt = ("diaggauss", mu, var)
            tp = np.stack([mu, var])
        elif tkind == "banana":
            tgt = ("banana", 1.0, 10.0)  # Banana(2, 1.0, 10.0): example/demo_planar_flow.jl:16
            tp = np.array([[1.0], [10.0]])
        elif tkind == "funnel":
            tgt = ("funnel", -2.0, 3.0)  # Funnel(d, mu, sigma): example/targets/neal_funnel.jl
            tp = np.array([[-2.0], [3.0]])
        elif tkind == "cross":
            tgt = ("cross", 2.0, 0.15)  # Cross(): example/targets/cross.jl:29
            tp = np.array([[2.0], [0.15]])
        else:
            tgt = ("warped", 1.0, 0.12)  # WarpedGauss(): example/targets/warped_gaussian.jl:37
            tp = np.array([[1.0], [0.12]])
        ys, ladj = o.flow_fwd(spec, theta, xs)
        xr, ladj_inv = o.flow_inv(spec, theta, ys)
        elbos = o.batched_elbos(spec, theta, tgt, xs)
        loss, grad = o.neg_elbo_value_and_grad(spec, theta, tgt, xs)
        th1 = theta.copy()
        o.adam_update(th1, grad, np.zeros_like(theta), np.zeros_like(theta), 1)
        ll = o.loglikelihood(spec, theta, ys)
        fkl_xs = ys.astype(dt).astype(np.float64)
        fkl_loss, fkl_grad = o.neg_loglik_value_and_grad(spec, theta, fkl_xs)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            kind=spec.kind, d=spec.d, nlayers=spec.nlayers, hdims=np.array(spec.hdims, dtype=np.int64), K=spec.K,
            B=spec.B, dtype=np.dtype(dt).name, target=tkind, target_params=tp,
            theta=theta.astype(dt), xs=xs.astype(dt), ys=ys, ladj=ladj, ladj_inv=ladj_inv, elbos=elbos, loss=loss,
            grad=grad.astype(dt), theta_adam1=th1.astype(dt), loglik_of_ys=ll,
            fkl_xs=fkl_xs.astype(dt), fkl_loss=fkl_loss, fkl_grad=fkl_grad.astype(dt),
        )
        print(f"{name}: P={theta.size} N={n} loss={loss:.6f} |g|inf={np.abs(grad).max():.3e} inv_err={np.abs(xr-xs).max():.1e}")


if __name__ == "__main__":
    main()
