"""Pins the CPU oracle (oracle/nf_oracle.py) with implementation-independent
definitions and with the reference's own property tests.

The reference (Julia) cannot run in the build container and ships no golden
vectors, so these are the strongest pins available:
  * log|det J| against slogdet of a finite-difference Jacobian;
  * the reference's invertibility tests at its tolerances
    (test/flow.jl:26-38,92-105,158-171,224-237);
  * ELBO(q == p) == 0 per sample (test/objectives.jl:15-25);
  * hand-derived gradients against central finite differences of the loss;
  * Philox4x32-10 against the Random123 known-answer vectors.
"""
import numpy as np
import pytest

import nf_oracle as o

SPECS = {
    "realnvp": o.FlowSpec("realnvp", 5, 2, (8, 8)),  # test/flow.jl:4-6 (hdims shrunk for FD cost)
    "nsf": o.FlowSpec("nsf", 5, 2, (8, 8), K=10, B=5.0),  # test/flow.jl:68-78
    "planar": o.FlowSpec("planar", 5, 10),  # test/flow.jl:137-139
    "radial": o.FlowSpec("radial", 5, 10),  # test/flow.jl:203-205
    "meanfield": o.FlowSpec("meanfield", 4, 1),
}


def _theta(spec, seed=0):
    rng = np.random.default_rng(seed)
    th = o.init_params(spec, rng)
    if spec.kind in ("realnvp", "nsf"):
        # non-zero biases so every term of the backward pass is exercised
        th = th + 0.1 * rng.standard_normal(th.shape)
    if spec.kind == "meanfield":
        th = th + 0.3 * rng.standard_normal(th.shape)
    return th


def _fd_jac(f, x, eps=1e-6):
    d = x.shape[0]
    J = np.zeros((d, d))
    for i in range(d):
        e = np.zeros((d, 1))
        e[i] = eps
        J[:, i] = ((f(x + e) - f(x - e)) / (2 * eps))[:, 0]
    return J


@pytest.mark.parametrize("kind", list(SPECS))
def test_logdet_is_slogdet_of_jacobian(kind):
    spec = SPECS[kind]
    th = _theta(spec)
    rng = np.random.default_rng(1)
    for _ in range(3):
        x = rng.standard_normal((spec.d, 1))
        _, ladj = o.flow_fwd(spec, th, x)
        J = _fd_jac(lambda v: o.flow_fwd(spec, th, v)[0], x)
        sign, ld = np.linalg.slogdet(J)
        assert sign > 0
        assert ladj[0] == pytest.approx(ld, rel=1e-6, abs=1e-7)


@pytest.mark.parametrize("kind,rtol", [("realnvp", 1e-6), ("nsf", 1e-4), ("planar", 1e-4), ("radial", 1e-4), ("meanfield", 1e-6)])
def test_inverse_compatibility(kind, rtol):
    """test/flow.jl 'Inverse compatibility': x ~= inv(fwd(x)), lj_fwd ~= -lj_bwd."""
    spec = SPECS[kind]
    th = _theta(spec)
    x = np.random.default_rng(2).standard_normal((spec.d, 10))
    y, lf = o.flow_fwd(spec, th, x)
    xr, lb = o.flow_inv(spec, th, y)
    np.testing.assert_allclose(xr, x, rtol=rtol, atol=1e-9)
    np.testing.assert_allclose(lf, -lb, rtol=rtol, atol=1e-9)
    # vector (N = 1) path gives the same numbers as the matrix path (test/flow.jl:26-31)
    y1, l1 = o.flow_fwd(spec, th, x[:, :1])
    np.testing.assert_allclose(y1[:, 0], y[:, 0], rtol=1e-12)
    np.testing.assert_allclose(l1[0], lf[0], rtol=1e-12)


def test_elbo_of_exact_posterior_is_zero():
    """test/objectives.jl:3-25: flow = Shift(mu) o Scale(sqrt(Sigma)) == target."""
    rng = np.random.default_rng(3)
    mu = rng.standard_normal(2)
    var = rng.uniform(size=2) + 1e-3
    spec = o.FlowSpec("meanfield", 2, 1)
    th = np.concatenate([mu, np.sqrt(var)])
    xs = rng.standard_normal((2, 10))
    tgt = ("diaggauss", mu, var)
    assert abs(o.elbo(spec, th, tgt, xs)) <= 1e-5
    assert abs(o.elbo_batch(spec, th, tgt, xs)) <= 1e-5
    np.testing.assert_allclose(o.batched_elbos(spec, th, tgt, xs), 0.0, atol=1e-12)
    # logpdf(flow, x) + el ~= logp(x)   (objectives.jl:18)
    x = rng.standard_normal((2, 1))
    xb, ladj = o.flow_inv(spec, th, x)
    lq = o.std_normal_logpdf(xb) + ladj
    np.testing.assert_allclose(lq, o.diaggauss_logp(x, mu, var), rtol=1e-12)


@pytest.mark.parametrize("kind", list(SPECS))
def test_elbo_equals_elbo_batch(kind):
    spec = SPECS[kind]
    th = _theta(spec)
    rng = np.random.default_rng(4)
    xs = rng.standard_normal((spec.d, 16))
    tgt = ("diaggauss", rng.standard_normal(spec.d), rng.uniform(size=spec.d) + 1e-3)
    assert o.elbo(spec, th, tgt, xs) == pytest.approx(o.elbo_batch(spec, th, tgt, xs), rel=1e-12)
    assert np.isfinite(o.elbo_batch(spec, th, tgt, xs))  # test/flow.jl:58-60


@pytest.mark.parametrize("kind", list(SPECS))
@pytest.mark.parametrize("target", ["diaggauss", "banana"])
def test_gradient_matches_finite_differences(kind, target):
    spec = SPECS[kind]
    if kind in ("planar", "radial"):
        spec = o.FlowSpec(kind, 3, 3)
    if kind in ("realnvp", "nsf"):
        spec = o.FlowSpec(kind, 3, 1, (4, 4), K=4 if kind == "nsf" else 0, B=3.0 if kind == "nsf" else 0.0)
    th = _theta(spec, 5)
    rng = np.random.default_rng(6)
    xs = rng.standard_normal((spec.d, 7))
    if target == "diaggauss":
        tgt = ("diaggauss", rng.standard_normal(spec.d), rng.uniform(size=spec.d) + 0.5)
    else:
        tgt = ("banana", 1.0, 10.0)
    loss, g = o.neg_elbo_value_and_grad(spec, th, tgt, xs)
    assert loss == pytest.approx(-o.elbo_batch(spec, th, tgt, xs), rel=1e-12)
    eps = 1e-6
    gfd = np.zeros_like(th)
    for i in range(len(th)):
        tp, tm = th.copy(), th.copy()
        tp[i] += eps
        tm[i] -= eps
        gfd[i] = (-o.elbo_batch(spec, tp, tgt, xs) + o.elbo_batch(spec, tm, tgt, xs)) / (2 * eps)
    np.testing.assert_allclose(g, gfd, rtol=2e-5, atol=1e-7)


def test_rqs_identity_outside_box_and_zero_net():
    spec = o.FlowSpec("nsf", 4, 1, (4, 4), K=4, B=2.0)
    th = _theta(spec)
    x = np.full((4, 3), 7.5)  # every coordinate outside [-B, B]
    y, ladj = o.flow_fwd(spec, th, x)
    np.testing.assert_array_equal(y, x)
    np.testing.assert_array_equal(ladj, 0.0)
    # all-zero conditioner weights => affine coupling is the identity with ladj = 0
    spec2 = o.FlowSpec("realnvp", 6, 2, (8, 8))
    y2, l2 = o.flow_fwd(spec2, np.zeros(o.param_count(spec2)), x[:3].repeat(2, 0))
    np.testing.assert_array_equal(l2, 0.0)
    np.testing.assert_array_equal(y2, x[:3].repeat(2, 0))


def test_param_counts_match_survey():
    """cfg2 P = 133 632 (h=64) / 50 688 (h=32); cfg3 P = 109 952; cfg1 P = 50."""
    assert o.param_count(o.FlowSpec("realnvp", 64, 4, (64, 64))) == 133632
    assert o.param_count(o.FlowSpec("realnvp", 64, 4, (32, 32))) == 50688
    assert o.param_count(o.FlowSpec("nsf", 32, 4, (32, 32), K=8, B=5.0)) == 109952
    assert o.param_count(o.FlowSpec("planar", 2, 10)) == 50
    assert o.param_count(o.FlowSpec("realnvp", 256, 8, (256, 256))) == 4214784


def test_flat_order_outer_first():
    """test/interface.jl:47-48: theta[1:2] is the shift (outer), theta[3:4] the scale."""
    spec = o.FlowSpec("meanfield", 2, 1)
    th = np.array([10.0, 10.0, 2.0, 2.0])
    y, ladj = o.flow_fwd(spec, th, np.zeros((2, 1)))
    np.testing.assert_allclose(y[:, 0], [10.0, 10.0])
    np.testing.assert_allclose(ladj, 2 * np.log(2.0))
    # realnvp: first coupling in flat order uses the ODD mask (0-based 0,2,4..) and runs LAST
    ls = o.layers_flat_order(o.FlowSpec("realnvp", 5, 2, (4, 4)))
    assert list(ls[0].idx_t) == [0, 2, 4] and list(ls[1].idx_t) == [1, 3]


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32-10."""
    u = lambda v: np.array([v], dtype=np.uint32)
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ]
    for ctr, key, want in kat:
        got = o.philox4x32_10(*[u(c) for c in ctr], np.uint32(key[0]), np.uint32(key[1]))
        assert tuple(int(g[0]) for g in got) == want


def test_base_sampler_statistics_and_shard_invariance():
    d, n = 6, 40000
    x = o.base_sample(d, n, seed=123)
    assert abs(x.mean()) < 0.01 and abs(x.var() - 1.0) < 0.02
    # mean(log q0) ~= -(d/2)(1 + log 2 pi)
    assert o.std_normal_logpdf(x).mean() == pytest.approx(-0.5 * d * (1 + o.LOG2PI), abs=0.05)
    # sharding by global sample index reproduces the same batch
    a = o.base_sample(d, 100, seed=123, sample_offset=0)
    b = o.base_sample(d, 50, seed=123, sample_offset=50)
    np.testing.assert_array_equal(a[:, 50:], b)


def test_adam_first_step_is_lr_sign():
    th = np.array([1.0, -2.0])
    g = np.array([0.5, -0.25])
    m, v = np.zeros(2), np.zeros(2)
    o.adam_update(th, g, m, v, 1, lr=1e-3)
    np.testing.assert_allclose(th, [1.0 - 1e-3, -2.0 + 1e-3], rtol=1e-6)


def test_meanfield_training_recovers_target():
    """test/interface.jl:14-50 in miniature: mean-field VI to N(10*1, 4I) with Adam."""
    spec = o.FlowSpec("meanfield", 2, 1)
    th = np.array([0.0, 0.0, 1.0, 1.0])
    tgt = ("diaggauss", np.full(2, 10.0), np.full(2, 4.0))
    m, v = np.zeros(4), np.zeros(4)
    rng = np.random.default_rng(0)
    el0 = o.elbo_batch(spec, th, tgt, rng.standard_normal((2, 1000)))
    for t in range(1, 3001):
        xs = rng.standard_normal((2, 10))
        _, g = o.neg_elbo_value_and_grad(spec, th, tgt, xs)
        o.adam_update(th, g, m, v, t, lr=0.01)
    assert np.all(np.abs(th[:2] - 10.0) < 0.2) and np.all(np.abs(th[2:] - 2.0) < 0.2)
    el1 = o.elbo_batch(spec, th, tgt, rng.standard_normal((2, 1000)))
    assert el1 > el0 and el1 > -1.0


@pytest.mark.parametrize("tgt,d", [(("funnel", 0.3, 2.0), 5), (("warped", 1.0, 0.12), 2), (("cross", 2.0, 0.15), 2)],
                         ids=["funnel", "warped", "cross"])
def test_synthetic_targets_gradient_matches_finite_differences(tgt, d):
    """example/targets/{neal_funnel,warped_gaussian,cross}.jl restated; the hand-derived scores are
    checked against central differences of the log-density."""
    rng = np.random.default_rng(3)
    y = rng.standard_normal((d, 9)) * 1.3
    g = o.target_grad(tgt, y)
    eps, gn = 1e-6, np.zeros_like(y)
    for i in range(d):
        yp, ym = y.copy(), y.copy()
        yp[i] += eps
        ym[i] -= eps
        gn[i] = (o.target_logp(tgt, yp) - o.target_logp(tgt, ym)) / (2 * eps)
    assert np.abs(g - gn).max() <= 1e-7 * max(1.0, np.abs(gn).max())


def test_synthetic_targets_analytic_anchors():
    """Known values: the funnel at y = (mu, 0, ..) ; the cross mixture integrates to one; the funnel's
    conditional is N(0, exp(y1)) (neal_funnel.jl:14-16)."""
    d, mu, sg = 4, 0.5, 3.0
    y = np.zeros((d, 1))
    y[0] = mu
    expect = -0.5 * o.LOG2PI - np.log(sg) - 0.5 * (d - 1) * (o.LOG2PI + mu)
    assert o.target_logp(("funnel", mu, sg), y)[0] == pytest.approx(expect, rel=1e-13)
    xs = np.linspace(-8, 8, 801)
    X, Y = np.meshgrid(xs, xs)
    P = np.stack([X.ravel(), Y.ravel()])
    mass = np.exp(o.target_logp(("cross", 2.0, 0.3), P)).sum() * (xs[1] - xs[0]) ** 2
    assert mass == pytest.approx(1.0, abs=1e-6)


@pytest.mark.parametrize("tname", ["funnel", "banana", "diaggauss"])
def test_hamiltonian_flow_oracle(tname):
    """example/demo_hamiltonian_flow.jl restated: LeapFrog is exactly inverted by -eps (:74-84) and has
    zero log-det (:86-93); the hand-derived reverse pass (with Hessian-vector products of the target)
    matches central differences of the loss; Hessian-vector products match differences of the score."""
    rng = np.random.default_rng(5)
    D, n, L = 3, 3, 3
    tgt = {"funnel": ("funnel", -2.0, 3.0), "banana": ("banana", 1.0, 10.0),
           "diaggauss": ("diaggauss", rng.standard_normal(D), rng.uniform(size=D) + 0.5)}[tname]
    x, v = rng.standard_normal((D, 6)), rng.standard_normal((D, 6))
    hv = (o.target_grad(tgt, x + 1e-6 * v) - o.target_grad(tgt, x - 1e-6 * v)) / 2e-6
    assert np.abs(hv - o.target_hvp(tgt, x, v)).max() < 1e-7
    P = o.hflow_param_count(D, n)
    th = np.concatenate([0.1 * rng.standard_normal(2 * D), 1 + 0.1 * rng.standard_normal(2 * D)]
                        + [np.concatenate([0.1 * rng.standard_normal(D), 1 + 0.1 * rng.standard_normal(D),
                                           np.log(0.05) + 0.1 * rng.standard_normal(D)]) for _ in range(n)])
    assert th.size == P
    x0 = rng.standard_normal((2 * D, 9))
    z, ladj = o.hflow_fwd(D, n, L, th, tgt, x0)
    xr, li = o.hflow_inv(D, n, L, th, tgt, z)
    assert np.abs(xr - x0).max() < 1e-12 and np.abs(ladj + li).max() < 1e-12
    _, scales = th[2 * D:4 * D], None
    expect = np.log(np.abs(th[2 * D:4 * D])).sum() + sum(np.log(np.abs(th[4 * D + 3 * D * b + D:4 * D + 3 * D * b + 2 * D])).sum() for b in range(n))
    assert np.allclose(ladj, expect)  # only the affine maps contribute: LeapFrog is symplectic
    loss, g = o.hflow_neg_elbo_value_and_grad(D, n, L, th, tgt, x0)

    def f(t):
        zz, ll = o.hflow_fwd(D, n, L, t, tgt, x0)
        return -np.mean(o.hflow_joint_logp(D, tgt, zz) - o.std_normal_logpdf(x0) + ll)

    assert loss == pytest.approx(f(th), rel=1e-13)
    gn = np.zeros_like(th)
    for i in range(P):
        tp, tm = th.copy(), th.copy()
        tp[i] += 1e-6
        tm[i] -= 1e-6
        gn[i] = (f(tp) - f(tm)) / 2e-6
    assert np.abs(g - gn).max() <= 1e-7 * max(1.0, np.abs(gn).max())


@pytest.mark.parametrize("kind", list(SPECS))
def test_loglikelihood_gradient_matches_finite_differences(kind):
    """Forward-KL training (train_flow(loglikelihood, ...)): the implicit-function reverse pass of the
    inverse chain against central differences of -loglikelihood."""
    spec = SPECS[kind]
    if kind in ("planar", "radial"):
        spec = o.FlowSpec(kind, 3, 3)
    if kind in ("realnvp", "nsf"):
        spec = o.FlowSpec(kind, 3, 1, (4, 4), K=4 if kind == "nsf" else 0, B=3.0 if kind == "nsf" else 0.0)
    th = _theta(spec, 9)
    rng = np.random.default_rng(10)
    ys = 0.8 * rng.standard_normal((spec.d, 6))
    loss, g = o.neg_loglik_value_and_grad(spec, th, ys)
    assert loss == pytest.approx(-o.loglikelihood(spec, th, ys), rel=1e-12)
    eps = 1e-6
    gfd = np.zeros_like(th)
    for i in range(len(th)):
        tp, tm = th.copy(), th.copy()
        tp[i] += eps
        tm[i] -= eps
        gfd[i] = (-o.loglikelihood(spec, tp, ys) + o.loglikelihood(spec, tm, ys)) / (2 * eps)
    np.testing.assert_allclose(g, gfd, rtol=5e-5, atol=2e-7)


def test_torch_cpu_baseline_graph_matches_the_oracle():
    """bench.py's CPU baseline (oracle/nf_torch_cpu.py: the reference's RealNVP step under torch-CPU autograd) computes
    the same loss and gradient as the hand-derived oracle -- so the thing timed as "CPU" is the algorithm, not a lookalike."""
    torch = pytest.importorskip("torch")
    import nf_torch_cpu as tc

    spec = o.FlowSpec("realnvp", 7, 2, (16, 12))
    rng = np.random.default_rng(5)
    th = o.init_params(spec, rng) + 0.1 * rng.standard_normal(o.param_count(spec))
    xs = rng.standard_normal((7, 33))
    mu, var = rng.standard_normal(7), rng.uniform(size=7) + 0.5
    l_ref, g_ref = o.neg_elbo_value_and_grad(spec, th, ("diaggauss", mu, var), xs)
    loss, g = tc.value_and_grad(spec, th, mu, var, xs)
    assert loss == pytest.approx(l_ref, rel=1e-12)
    np.testing.assert_allclose(g, g_ref, rtol=1e-9, atol=1e-12)
    ys, ladj = tc.realnvp_forward(spec, torch.tensor(th), torch.tensor(xs.T.copy()))
    y_ref, l_ref2 = o.flow_fwd(spec, th, xs)
    np.testing.assert_allclose(ys.numpy().T, y_ref, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(ladj.numpy(), l_ref2, rtol=1e-12, atol=1e-13)


def test_hamiltonian_theta_order_reference_map_first():
    """The documented theta order of the Hamiltonian demo flow (include/nfhip.h): the reference
    distribution's Shift / Scale occupy theta[0:4D], the blocks follow.  With vanishing leapfrog steps and identity
    momentum layers the flow IS that affine map -- the analogue of test/interface.jl:47-48 for this flow."""
    D, n, L = 2, 3, 2
    th = np.concatenate([np.arange(1.0, 5.0), np.arange(2.0, 6.0)] + [np.concatenate([np.zeros(D), np.ones(D), np.full(D, -60.0)])] * n)
    assert th.size == o.hflow_param_count(D, n)
    x0 = np.random.default_rng(0).standard_normal((2 * D, 7))
    z, ladj = o.hflow_fwd(D, n, L, th, ("diaggauss", np.zeros(D), np.ones(D)), x0)
    np.testing.assert_allclose(z, th[:4, None] + th[4:8, None] * x0, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ladj, np.log(th[4:8]).sum(), rtol=1e-12)


def test_general_base_density_is_the_multivariate_normal():
    """The oracle's general-base log-density against scipy's multivariate normal, and draws mu + L eps having the
    stated mean / covariance (the statistical check of test/ext/CUDA/cuda.jl:33-45, on the oracle)."""
    import scipy.stats

    rng = np.random.default_rng(0)
    d = 4
    A = rng.standard_normal((d, d))
    Sigma = A @ A.T + 0.5 * np.eye(d)
    mu = rng.standard_normal(d)
    L = np.linalg.cholesky(Sigma)
    x = rng.standard_normal((d, 50)) * 2
    np.testing.assert_allclose(o.base_logpdf(("dense", mu, L), x), scipy.stats.multivariate_normal(mu, Sigma).logpdf(x.T), rtol=1e-12)
    sig = np.sqrt(np.diag(Sigma))
    np.testing.assert_allclose(o.base_logpdf(("diag", mu, sig), x), scipy.stats.multivariate_normal(mu, np.diag(sig**2)).logpdf(x.T), rtol=1e-12)
    np.testing.assert_allclose(o.base_logpdf(None, x), scipy.stats.multivariate_normal(np.zeros(d), np.eye(d)).logpdf(x.T), rtol=1e-12)
    eps = o.base_sample(d, 200000, seed=3)
    xs = o.base_unwhiten(("dense", mu, L), eps)
    assert np.abs(xs.mean(axis=1) - mu).max() < 0.02 and np.abs(np.cov(xs) - Sigma).max() < 0.05


def test_composite_flow_gradient_matches_finite_differences():
    """Heterogeneous create_flow((radial..., realnvp..., planar...), q0) (src/flows/utils.jl:23-26): the oracle's chained
    reverse pass against central differences of its loss; inverse o forward = identity with opposite log-dets."""
    specs = [o.FlowSpec("radial", 4, 2), o.FlowSpec("realnvp", 4, 1, (6, 6)), o.FlowSpec("planar", 4, 2)]
    rng = np.random.default_rng(3)
    th = np.concatenate([o.init_params(sp, rng) * (0.4 if sp.kind != "realnvp" else 1.0) for sp in specs])
    xs = rng.standard_normal((4, 9))
    tgt = ("diaggauss", rng.standard_normal(4), rng.uniform(size=4) + 0.5)
    loss, g = o.comp_neg_elbo_value_and_grad(specs, th, tgt, xs)
    gfd = np.zeros_like(th)
    for i in range(th.size):
        tp, tm = th.copy(), th.copy()
        tp[i] += 1e-6
        tm[i] -= 1e-6
        gfd[i] = (o.comp_neg_elbo_value_and_grad(specs, tp, tgt, xs)[0] - o.comp_neg_elbo_value_and_grad(specs, tm, tgt, xs)[0]) / 2e-6
    np.testing.assert_allclose(g, gfd, rtol=2e-5, atol=1e-7)
    ys, lf = o.comp_fwd(specs, th, xs)
    xr, lb = o.comp_inv(specs, th, ys)
    np.testing.assert_allclose(xr, xs, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(lf, -lb, rtol=1e-6, atol=1e-8)


def test_general_forward_kl_gradient_matches_finite_differences():
    """train_flow(loglikelihood, flow, ys) for a heterogeneous composition over a dense MvNormal base: the oracle's
    reverse pass of the inverse chain against central differences; one segment over the standard base is the
    single-family function."""
    specs = [o.FlowSpec("radial", 4, 2), o.FlowSpec("realnvp", 4, 1, (6, 6)), o.FlowSpec("planar", 4, 2)]
    rng = np.random.default_rng(5)
    th = np.concatenate([o.init_params(sp, rng) * (0.4 if sp.kind != "realnvp" else 1.0) for sp in specs])
    A = rng.standard_normal((4, 4))
    base = ("dense", rng.standard_normal(4), np.linalg.cholesky(A @ A.T + 0.5 * np.eye(4)))
    ys = rng.standard_normal((4, 9))
    for b in (base, ("diag", base[1], np.abs(np.diag(base[2]))), None):
        loss, g = o.comp_neg_loglik_value_and_grad(specs, th, ys, b)
        gfd = np.zeros_like(th)
        for i in range(th.size):
            tp, tm = th.copy(), th.copy()
            tp[i] += 1e-6
            tm[i] -= 1e-6
            gfd[i] = (o.comp_neg_loglik_value_and_grad(specs, tp, ys, b)[0] - o.comp_neg_loglik_value_and_grad(specs, tm, ys, b)[0]) / 2e-6
        np.testing.assert_allclose(g, gfd, rtol=5e-5, atol=2e-7)
    sp = specs[1]
    th1 = o.init_params(sp, rng)
    l0, g0 = o.neg_loglik_value_and_grad(sp, th1, ys)
    l1, g1 = o.comp_neg_loglik_value_and_grad([sp], th1, ys, None)
    np.testing.assert_allclose(l1, l0, rtol=1e-13)
    np.testing.assert_allclose(g1, g0, rtol=1e-12, atol=1e-15)
    zs, ladj = o.comp_inv([sp], th1, ys)
    np.testing.assert_allclose(-(o.base_logpdf(base, zs) + ladj).mean(), o.comp_neg_loglik_value_and_grad([sp], th1, ys, base)[0], rtol=1e-13)


def test_float64_base_stream_has_53_bit_uniforms_and_reaches_the_tails():
    """The Float64 draw stream (oracle precision="f64", mirrored by philox_normals4<double> on the device): standard
    normal moments, shard invariance, and tails beyond the |z| <= 5.77 cap of the 23-bit Float32 stream."""
    x = o.base_sample(8, 200000, seed=5, precision="f64")
    assert abs(x.mean()) < 5e-3 and abs(x.var() - 1.0) < 1e-2
    b = o.base_sample(8, 100, seed=5, sample_offset=199900, precision="f64")
    np.testing.assert_array_equal(b, x[:, 199900:])
    x32 = o.base_sample(8, 200000, seed=5)
    # the 23-bit stream cannot exceed sqrt(-2 log 2^-24); the 53-bit one can (not necessarily within 1.6 M draws, but its cap is 8.6)
    assert np.abs(x32).max() <= np.sqrt(-2.0 * np.log(2.0**-24)) + 1e-9
    u_min_53 = 0.5 * 2.0**-53
    assert np.sqrt(-2.0 * np.log(u_min_53)) > 8.5


def test_cpp_openmp_baseline_matches_the_oracle():
    """bench.py's second CPU baseline (oracle/nf_cpu_step.cpp, built at run time): loss and gradient of the RealNVP step
    on supplied draws against the numpy oracle (fp32 arithmetic with -ffast-math, so fp32-level agreement), odd d."""
    import nf_cpu_omp as co

    if co.load() is None:
        pytest.skip("no C++ compiler on this host")
    for d, hd, nl in ((7, (16, 12), 2), (64, (64, 64), 1)):
        spec = o.FlowSpec("realnvp", d, nl, hd)
        rng = np.random.default_rng(d)
        th = (o.init_params(spec, rng) + 0.05 * rng.standard_normal(o.param_count(spec))).astype(np.float32)
        xs = rng.standard_normal((d, 128)).astype(np.float32)
        mu, var = rng.standard_normal(d).astype(np.float32), (rng.uniform(size=d) + 0.5).astype(np.float32)
        l_ref, g_ref = o.neg_elbo_value_and_grad(spec, th.astype(np.float64), ("diaggauss", mu.astype(np.float64), var.astype(np.float64)),
                                                 xs.astype(np.float64))
        loss, g = co.value_and_grad(d, hd, nl, th, mu, var, xs, nthreads=3)
        assert loss == pytest.approx(l_ref, rel=2e-5)
        assert np.abs(g - g_ref).max() <= 2e-4 * np.abs(g_ref).max()
    r = co.time_training_steps(8, (16, 16), 1, 256, threads=2, seconds_budget=0.2, max_steps=3)
    assert r["value"] > 0 and np.isfinite(r["loss"])
