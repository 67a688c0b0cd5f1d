"""CPU oracle for the ELBO / reverse-KL hot path of TuringLang/NormalizingFlows.jl.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this module;
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may.

PARITY STATUS: **parity unpinned** at the third-party boundary.  The reference is
pure Julia, no Julia toolchain exists in the build container, and the reference
ships no golden vectors.  The in-tree arithmetic
(AffineCoupling, ELBO assembly, composition order, parameter flattening) is a
restatement of files under /root/reference/src that is cited line by line below.
The arithmetic that lives in un-vendored dependencies -- Bijectors.jl 0.16.2
(PlanarLayer, RadialLayer), MonotonicSplines.jl 0.3.3 (rational-quadratic
splines), Distributions.jl 0.25 (MvNormal), Flux 0.16 (Dense, leakyrelu),
Optimisers 0.4 (Adam, destructure order) -- is restated from the published
algorithms and pinned only by (a) the reference's own property tests
(test/flow.jl, test/objectives.jl) and (b) implementation-independent
definitions (autograd Jacobian log-determinants), both exercised in
tests/test_oracle.py.

Conventions (reference: column = sample, src/objectives/elbo.jl:52,60):
  a batch is an array of shape (d, N); in C terms x[j*d + i].
All functions are dtype-generic (float64 by default; float32 when fed float32).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import numpy as np

LOG2PI = float(np.log(2.0 * np.pi))

# --------------------------------------------------------------------------
# Flow specification and the flat parameter layout
# --------------------------------------------------------------------------


@dataclass(frozen=True)
class FlowSpec:
    """Static (non-trainable) description of a flow.

    kind     : "planar" | "radial" | "realnvp" | "nsf" | "meanfield"
    d        : dimension of the base distribution
    nlayers  : planar/radial: number of layers (src/flows/planar_radial.jl:21-29,52-60)
               realnvp/nsf : number of RealNVP_layer / NSF_layer blocks, each of which is
               TWO couplings (src/flows/realnvp.jl:132-145, src/flows/neuralspline.jl:169-184)
    hdims    : hidden widths of the conditioner MLPs (src/flows/utils.jl:71-100)
    K, B     : number of spline bins and box bound (src/flows/neuralspline.jl:44-61)
    """

    kind: str
    d: int
    nlayers: int
    hdims: Tuple[int, ...] = ()
    K: int = 0
    B: float = 0.0


@dataclass
class LayerInfo:
    kind: str  # planar | radial | affine | rqs
    offset: int  # offset of this layer's parameters in theta_flat
    nparams: int
    # couplings only
    idx_t: np.ndarray = None  # transformed indices (0-based, increasing)
    idx_c: np.ndarray = None  # conditioner indices (complement, increasing)
    nets: list = field(default_factory=list)  # list of nets; net = list of (w_off, b_off, nout, nin)


def _mlp_layout(off: int, nin: int, hdims: Sequence[int], nout: int):
    """Flux Chain of Dense layers, destructured depth-first: weight (out x in,
    column-major) then bias, layer by layer (Optimisers.destructure; Flux.Dense
    fields are (weight, bias, sigma)).  src/flows/utils.jl:81-99."""
    dims = [nin] + list(hdims) + [nout]
    layers = []
    for a, b in zip(dims[:-1], dims[1:]):
        w_off = off
        off += a * b
        b_off = off
        off += b
        layers.append((w_off, b_off, b, a))
    return layers, off


def layers_flat_order(spec: FlowSpec) -> List[LayerInfo]:
    """Layers in FLAT (destructure) order L1..Ln.

    create_flow composes reduce(o, Ls) (src/flows/utils.jl:23-26) and
    ComposedFunction stores (outer, inner), so L1's parameters come first
    (pinned for two layers by test/interface.jl:47-48).  EXECUTION order on a
    base draw is the reverse: Ln first, L1 last.
    """
    d = spec.d
    out: List[LayerInfo] = []
    off = 0
    if spec.kind == "planar":
        for _ in range(spec.nlayers):  # fields w(d), u(d), b(1)  (Bijectors PlanarLayer)
            out.append(LayerInfo("planar", off, 2 * d + 1))
            off += 2 * d + 1
    elif spec.kind == "radial":
        for _ in range(spec.nlayers):  # fields alpha_(1), beta(1), z_0(d)  (Bijectors RadialLayer)
            out.append(LayerInfo("radial", off, d + 2))
            off += d + 2
    elif spec.kind == "meanfield":
        # Shift(mu) o Scale(sigma) on a standard normal: the mean-field Gaussian of
        # test/interface.jl:22-25 and test/objectives.jl:8-9.  Shift is the OUTER
        # function, so its parameters come first (test/interface.jl:47-48).
        out.append(LayerInfo("shift", 0, d))
        out.append(LayerInfo("scale", d, d))
    elif spec.kind in ("realnvp", "nsf"):
        allidx = np.arange(d)
        for _ in range(spec.nlayers):
            # mask_idx1 = 1:2:dims (0-based 0,2,4..), mask_idx2 = 2:2:dims (0-based 1,3,..)
            # src/flows/realnvp.jl:138-139, src/flows/neuralspline.jl:176-177
            for start in (0, 1):
                idx_t = np.arange(start, d, 2)
                idx_c = np.setdiff1d(allidx, idx_t)
                c, m = len(idx_t), len(idx_c)
                li = LayerInfo("affine" if spec.kind == "realnvp" else "rqs", off, 0, idx_t, idx_c)
                if spec.kind == "realnvp":
                    # @functor AffineCoupling (s, t): s first, then t (realnvp.jl:40)
                    s, off2 = _mlp_layout(off, m, spec.hdims, c)
                    t, off3 = _mlp_layout(off2, m, spec.hdims, c)
                    li.nets = [s, t]
                    li.nparams = off3 - off
                else:
                    # one net with (3K-1)*c outputs (neuralspline.jl:55-57)
                    nn, off3 = _mlp_layout(off, m, spec.hdims, (3 * spec.K - 1) * c)
                    li.nets = [nn]
                    li.nparams = off3 - off
                off = off3
                out.append(li)
    else:
        raise ValueError(f"unknown flow kind {spec.kind!r}")
    return out


def param_count(spec: FlowSpec) -> int:
    ls = layers_flat_order(spec)
    return ls[-1].offset + ls[-1].nparams if ls else 0


def init_params(spec: FlowSpec, rng: np.random.Generator, dtype=np.float64) -> np.ndarray:
    """Random initial parameters with the reference's init *distributions*:
    Flux.Dense -> Glorot-uniform weights, zero bias; PlanarLayer/RadialLayer(dim)
    -> randn.  (Specific draws are not reproducible outside Julia and no
    reference test depends on them; SURVEY.md App. B.)"""
    theta = np.zeros(param_count(spec), dtype=np.float64)
    for li in layers_flat_order(spec):
        if li.kind in ("planar", "radial"):
            theta[li.offset : li.offset + li.nparams] = rng.standard_normal(li.nparams)
        elif li.kind == "shift":
            theta[li.offset : li.offset + li.nparams] = 0.0
        elif li.kind == "scale":
            theta[li.offset : li.offset + li.nparams] = 1.0
        else:
            for net in li.nets:
                for (w_off, b_off, nout, nin) in net:
                    lim = np.sqrt(6.0 / (nin + nout))
                    theta[w_off : w_off + nout * nin] = rng.uniform(-lim, lim, nout * nin)
    return theta.astype(dtype)


# --------------------------------------------------------------------------
# Elementary functions
# --------------------------------------------------------------------------


def softplus(x):
    """log1pexp, numerically stable (LogExpFunctions.log1pexp)."""
    x = np.asarray(x)
    return np.where(x > 0, x + np.log1p(np.exp(-np.abs(x))), np.log1p(np.exp(-np.abs(x))))


def sigmoid(x):
    x = np.asarray(x)
    e = np.exp(-np.abs(x))
    return np.where(x >= 0, 1.0 / (1.0 + e), e / (1.0 + e))


def leakyrelu(x, slope=0.01):
    """Flux.leakyrelu default slope 0.01 (NNlib); hidden activation of fnn,
    src/flows/utils.jl:76."""
    return np.maximum(x, x.dtype.type(slope) * x)  # == where(x > 0, x, slope*x) for 0 < slope < 1


def _dense_views(theta, net):
    return [
        (theta[w_off : w_off + nout * nin].reshape(nin, nout).T, theta[b_off : b_off + nout])
        for (w_off, b_off, nout, nin) in net
    ]


def mlp_forward(theta, net, x, out_act=None, keep=False):
    """fnn: leakyrelu hidden layers, optional output activation
    (src/flows/utils.jl:71-100).  x: (nin, N)."""
    acts = [x]
    a = x
    layers = _dense_views(theta, net)
    for li, (W, b) in enumerate(layers):
        z = W @ a + b[:, None]
        if li < len(layers) - 1:
            a = leakyrelu(z)
        else:
            a = np.tanh(z) if out_act == "tanh" else z
        acts.append(a)
    return (a, acts) if keep else a


def mlp_backward(theta, net, acts, dout, grad, out_act=None):
    """Reverse pass of mlp_forward.  acts from keep=True, dout = dL/d(output).
    Accumulates dW, db into `grad` (flat) and returns dL/d(input)."""
    layers = _dense_views(theta, net)
    nl = len(layers)
    delta = dout
    for li in range(nl - 1, -1, -1):
        W, _ = layers[li]
        a_out = acts[li + 1]
        if li == nl - 1:
            if out_act == "tanh":
                delta = delta * (1.0 - a_out * a_out)
        else:
            delta = delta * (delta.dtype.type(0.01) + delta.dtype.type(0.99) * (a_out > 0))
        (w_off, b_off, nout, nin) = net[li]
        dW = delta @ acts[li].T  # (nout, nin)
        grad[w_off : w_off + nout * nin] += dW.T.reshape(-1)  # column-major out x in
        grad[b_off : b_off + nout] += delta.sum(axis=1)
        delta = W.T @ delta
    return delta


# --------------------------------------------------------------------------
# PlanarLayer  (Bijectors.jl planar_layer.jl; formulas corroborated in-tree by
# /root/reference/test/ext/CUDA/cuda.jl:12-30)
# --------------------------------------------------------------------------


def planar_uhat(w, u):
    """get_u_hat: test/ext/CUDA/cuda.jl:12-18."""
    wtu = w @ u
    scale = (softplus(-wtu) - 1.0) / (w @ w)
    uhat = u + scale * w
    wt_uhat = softplus(wtu) - 1.0
    return uhat, wt_uhat


def planar_fwd(p, z):
    """y = z + u_hat tanh(w'z+b); ladj = log1p(w'u_hat * sech^2(w'z+b))."""
    d = z.shape[0]
    w, u, b = p[:d], p[d : 2 * d], p[2 * d]
    uhat, c = planar_uhat(w, u)
    a = w @ z + b
    t = np.tanh(a)
    y = z + uhat[:, None] * t[None, :]
    ladj = np.log1p(c * (1.0 - t * t))
    return y, ladj


def planar_inv(p, y, iters=200):
    """Solve alpha + c*tanh(alpha + b) = w'y for alpha = w'z (monotone since c > -1),
    by bisection on [w'y - |c|, w'y + |c|]; z = y - u_hat*tanh(alpha+b).
    (Upstream uses Roots.jl; any bracketing solver converges to the same root.)"""
    d = y.shape[0]
    w, u, b = p[:d], p[d : 2 * d], p[2 * d]
    uhat, c = planar_uhat(w, u)
    wy = w @ y
    lo = wy - abs(c)
    hi = wy + abs(c)
    for _ in range(iters):
        mid = 0.5 * (lo + hi)
        f = mid + c * np.tanh(mid + b) - wy
        hi = np.where(f > 0, mid, hi)
        lo = np.where(f > 0, lo, mid)
    alpha = 0.5 * (lo + hi)
    t = np.tanh(alpha + b)
    z = y - uhat[:, None] * t[None, :]
    ladj = -np.log1p(c * (1.0 - t * t))
    return z, ladj


def planar_bwd(p, z, ybar, lbar, gp):
    """Hand-derived reverse pass.  lbar: (N,) cotangent of ladj."""
    d = z.shape[0]
    w, u, b = p[:d], p[d : 2 * d], p[2 * d]
    m = w @ u
    ww = w @ w
    sp_neg = softplus(-m) - 1.0
    uhat = u + sp_neg * w / ww
    c = softplus(m) - 1.0
    a = w @ z + b
    t = np.tanh(a)
    g = 1.0 - t * t
    D = 1.0 + c * g
    abar = (uhat @ ybar) * g - 2.0 * lbar * c * t * g / D
    zbar = ybar + w[:, None] * abar[None, :]
    bbar = abar.sum()
    wbar = z @ abar
    uhat_bar = ybar @ t
    cbar = (lbar * g / D).sum()
    sg = sigmoid(m)
    uw = uhat_bar @ w
    mbar = cbar * sg + uw * (sg - 1.0) / ww
    ubar = uhat_bar + mbar * w
    wbar = wbar + mbar * u + sp_neg * (uhat_bar / ww - 2.0 * uw * w / (ww * ww))
    gp[:d] += wbar
    gp[d : 2 * d] += ubar
    gp[2 * d] += bbar
    return zbar


# --------------------------------------------------------------------------
# RadialLayer  (Bijectors.jl radial_layer.jl)
# --------------------------------------------------------------------------


def radial_fwd(p, z):
    d = z.shape[0]
    alpha = softplus(p[0])
    beta_hat = -alpha + softplus(p[1])
    z0 = p[2 : 2 + d]
    delta = z - z0[:, None]
    r = np.sqrt((delta * delta).sum(axis=0))
    h = 1.0 / (alpha + r)
    y = z + beta_hat * h[None, :] * delta
    ladj = (d - 1) * np.log1p(beta_hat * h) + np.log1p(beta_hat * h - beta_hat * h * h * r)
    return y, ladj


def radial_inv(p, y):
    """Closed form: rho = |y - z0| = r (1 + beta_hat/(alpha + r)) -> quadratic in r."""
    d = y.shape[0]
    alpha = softplus(p[0])
    beta_hat = -alpha + softplus(p[1])
    z0 = p[2 : 2 + d]
    dy = y - z0[:, None]
    rho = np.sqrt((dy * dy).sum(axis=0))
    a = (alpha + beta_hat) - rho
    r = 0.5 * (np.sqrt(a * a + 4.0 * alpha * rho) - a)
    z = z0[:, None] + ((alpha + r) / (alpha + beta_hat + r))[None, :] * dy
    h = 1.0 / (alpha + r)
    ladj = -((d - 1) * np.log1p(beta_hat * h) + np.log1p(beta_hat * h - beta_hat * h * h * r))
    return z, ladj


def radial_bwd(p, z, ybar, lbar, gp):
    """Hand-derived reverse pass."""
    d = z.shape[0]
    alpha = softplus(p[0])
    beta_hat = -alpha + softplus(p[1])
    z0 = p[2 : 2 + d]
    delta = z - z0[:, None]
    r = np.sqrt((delta * delta).sum(axis=0))
    h = 1.0 / (alpha + r)
    q = beta_hat * h
    bah2 = beta_hat * alpha * h * h
    dL_dh = (d - 1) * beta_hat / (1.0 + q) + 2.0 * beta_hat * alpha * h / (1.0 + bah2)
    dL_db = (d - 1) * h / (1.0 + q) + alpha * h * h / (1.0 + bah2)
    dL_da = beta_hat * h * h / (1.0 + bah2)
    yd = (ybar * delta).sum(axis=0)
    hbar = beta_hat * yd + lbar * dL_dh
    rbar = -h * h * hbar
    rsafe = np.where(r > 0, r, 1.0)
    dbar = q[None, :] * ybar + (rbar / rsafe)[None, :] * delta
    zbar = ybar + dbar
    z0bar = -dbar.sum(axis=1)
    alpha_bar = (-h * h * hbar + lbar * dL_da).sum()
    bhat_bar = (h * yd + lbar * dL_db).sum()
    gp[0] += (alpha_bar - bhat_bar) * sigmoid(p[0])
    gp[1] += bhat_bar * sigmoid(p[1])
    gp[2 : 2 + d] += z0bar
    return zbar


# --------------------------------------------------------------------------
# AffineCoupling  (in-tree: src/flows/realnvp.jl:57-110)
# --------------------------------------------------------------------------


def affine_fwd(theta, li: LayerInfo, x):
    """realnvp.jl:77-83: y1 = exp(s(x2)) .* x1 .+ t(x2); logjac = sum(s(x2); dims=1)."""
    x1, x2 = x[li.idx_t], x[li.idx_c]
    S = mlp_forward(theta, li.nets[0], x2, "tanh")  # realnvp.jl:50 (tanh output)
    T = mlp_forward(theta, li.nets[1], x2, None)  # realnvp.jl:52
    y = x.copy()
    y[li.idx_t] = np.exp(S) * x1 + T
    return y, S.sum(axis=0)


def affine_inv(theta, li: LayerInfo, y):
    """realnvp.jl:99-110: x1 = (y1 - t(y2)) .* exp(-s(y2)); logjac = -sum(s)."""
    y1, y2 = y[li.idx_t], y[li.idx_c]
    S = mlp_forward(theta, li.nets[0], y2, "tanh")
    T = mlp_forward(theta, li.nets[1], y2, None)
    x = y.copy()
    x[li.idx_t] = (y1 - T) * np.exp(-S)
    return x, -S.sum(axis=0)


def affine_bwd(theta, li: LayerInfo, x, ybar, lbar, grad):
    """Reverse pass of affine_fwd."""
    x1, x2 = x[li.idx_t], x[li.idx_c]
    S, acts_s = mlp_forward(theta, li.nets[0], x2, "tanh", keep=True)
    T, acts_t = mlp_forward(theta, li.nets[1], x2, None, keep=True)
    eS = np.exp(S)
    y1bar = ybar[li.idx_t]
    Sbar = y1bar * x1 * eS + lbar[None, :]
    x2bar = ybar[li.idx_c].copy()
    x2bar += mlp_backward(theta, li.nets[0], acts_s, Sbar, grad, "tanh")
    x2bar += mlp_backward(theta, li.nets[1], acts_t, y1bar, grad, None)
    xbar = np.empty_like(ybar)
    xbar[li.idx_t] = y1bar * eS
    xbar[li.idx_c] = x2bar
    return xbar


# --------------------------------------------------------------------------
# Rational-quadratic splines  (MonotonicSplines.jl 0.3.3: rqs_params_from_nn,
# rqs_forward, rqs_inverse; Durkan et al. 2019).  Call sites:
# src/flows/neuralspline.jl:65-71,102-108,134-140.
# --------------------------------------------------------------------------


def rqs_params_from_nn(raw, c: int, B: float):
    """raw: ((3K-1)*c, N) -> pX, pY, dYdX each (K+1, c, N).
    Per transformed dim: rows 0:K widths, K:2K heights, 2K:3K-1 interior
    derivatives.  Knots = -B + 2B*cumsum(softmax); boundary derivatives = 1."""
    n = raw.shape[1]
    P = raw.shape[0] // c
    K = (P + 1) // 3
    th = raw.reshape(c, P, n).transpose(1, 0, 2)  # Julia reshape(:, c, N): params fastest

    def knots(v):
        v = v - v.max(axis=0, keepdims=True)
        e = np.exp(v)
        sm = e / e.sum(axis=0, keepdims=True)
        cs = np.cumsum(sm, axis=0)
        lead = np.full((1,) + cs.shape[1:], -B, dtype=raw.dtype)
        return np.concatenate([lead, (-B + 2.0 * B * cs).astype(raw.dtype)], axis=0)

    pX = knots(th[:K])
    pY = knots(th[K : 2 * K])
    one = np.ones((1,) + pX.shape[1:], dtype=raw.dtype)
    dYdX = np.concatenate([one, softplus(th[2 * K :]).astype(raw.dtype), one], axis=0)
    return pX, pY, dYdX


def _rqs_bins(p, v, K):
    """Index k of the bin with p[k] <= v < p[k+1]; -1 where v is outside."""
    inside = (v >= p[0]) & (v < p[K])
    k = (v[None] >= p[1:K]).sum(axis=0)  # 0..K-1
    return k, inside


def _take(p, k):
    return np.take_along_axis(p, k[None], axis=0)[0]


def rqs_forward(x1, pX, pY, dYdX):
    """y, logjac(N,).  Identity with zero log-derivative outside [-B, B]."""
    K = pX.shape[0] - 1
    k, inside = _rqs_bins(pX, x1, K)
    xk, xk1 = _take(pX, k), _take(pX, k + 1)
    yk, yk1 = _take(pY, k), _take(pY, k + 1)
    d0, d1 = _take(dYdX, k), _take(dYdX, k + 1)
    dx, dy = xk1 - xk, yk1 - yk
    s = dy / dx
    xi = np.where(inside, (x1 - xk) / dx, 0.5)  # lanes outside [-B, B] are the identity: keep their dummy xi in (0, 1)
    om = 1.0 - xi
    den = s + (d1 + d0 - 2.0 * s) * xi * om
    y = yk + dy * (s * xi * xi + d0 * xi * om) / den
    logd = 2.0 * np.log(s) + np.log(d1 * xi * xi + 2.0 * s * xi * om + d0 * om * om) - 2.0 * np.log(den)
    y = np.where(inside, y, x1)
    logd = np.where(inside, logd, 0.0)
    return y.astype(x1.dtype), logd.sum(axis=0).astype(x1.dtype)


def rqs_inverse(y1, pX, pY, dYdX):
    K = pX.shape[0] - 1
    k, inside = _rqs_bins(pY, y1, K)
    xk, xk1 = _take(pX, k), _take(pX, k + 1)
    yk, yk1 = _take(pY, k), _take(pY, k + 1)
    d0, d1 = _take(dYdX, k), _take(dYdX, k + 1)
    dx, dy = xk1 - xk, yk1 - yk
    s = dy / dx
    yy = np.where(inside, y1 - yk, 0.5 * dy)  # outside lanes: identity, dummy in-bin value
    a = dy * (s - d0) + yy * (d1 + d0 - 2.0 * s)
    b = dy * d0 - yy * (d1 + d0 - 2.0 * s)
    c = -s * yy
    disc = np.maximum(b * b - 4.0 * a * c, 0.0)
    xi = 2.0 * c / (-b - np.sqrt(disc))
    om = 1.0 - xi
    x = xi * dx + xk
    den = s + (d1 + d0 - 2.0 * s) * xi * om
    logd = 2.0 * np.log(s) + np.log(d1 * xi * xi + 2.0 * s * xi * om + d0 * om * om) - 2.0 * np.log(den)
    x = np.where(inside, x, y1)
    logd = np.where(inside, -logd, 0.0)
    return x.astype(y1.dtype), logd.sum(axis=0).astype(y1.dtype)


def rqs_fwd(theta, li: LayerInfo, x, K: int, B: float):
    """neuralspline.jl:102-108."""
    x1, x2 = x[li.idx_t], x[li.idx_c]
    raw = mlp_forward(theta, li.nets[0], x2, None)
    pX, pY, dd = rqs_params_from_nn(raw, len(li.idx_t), B)
    y1, lj = rqs_forward(x1, pX, pY, dd)
    y = x.copy()
    y[li.idx_t] = y1
    return y, lj


def rqs_inv(theta, li: LayerInfo, y, K: int, B: float):
    """neuralspline.jl:134-140."""
    y1, y2 = y[li.idx_t], y[li.idx_c]
    raw = mlp_forward(theta, li.nets[0], y2, None)
    pX, pY, dd = rqs_params_from_nn(raw, len(li.idx_t), B)
    x1, lj = rqs_inverse(y1, pX, pY, dd)
    x = y.copy()
    x[li.idx_t] = x1
    return x, lj


def rqs_bwd(theta, li: LayerInfo, x, ybar, lbar, grad, K: int, B: float):
    """Reverse pass of rqs_fwd, hand-derived through the bin-local rational
    quadratic, the softmax/cumsum knot construction and softplus derivatives,
    then the conditioner MLP."""
    x1, x2 = x[li.idx_t], x[li.idx_c]
    c = len(li.idx_t)
    n = x.shape[1]
    raw, acts = mlp_forward(theta, li.nets[0], x2, None, keep=True)
    P = 3 * K - 1
    th = raw.reshape(c, P, n).transpose(1, 0, 2)
    pX, pY, dd = rqs_params_from_nn(raw, c, B)
    k, inside = _rqs_bins(pX, x1, K)
    xk, xk1 = _take(pX, k), _take(pX, k + 1)
    yk, yk1 = _take(pY, k), _take(pY, k + 1)
    d0, d1 = _take(dd, k), _take(dd, k + 1)
    dx, dy = xk1 - xk, yk1 - yk
    s = dy / dx
    xi = (x1 - xk) / dx
    om = 1.0 - xi
    q = d1 + d0 - 2.0 * s
    den = s + q * xi * om
    num = s * xi * xi + d0 * xi * om
    nd = d1 * xi * xi + 2.0 * s * xi * om + d0 * om * om
    y1bar = ybar[li.idx_t]
    lb = lbar[None, :]
    # partials of y = yk + dy*num/den and L = 2 log s + log nd - 2 log den w.r.t. (xi, s, d0, d1, dy, yk)
    dnum_dxi = 2.0 * s * xi + d0 * (1.0 - 2.0 * xi)
    dden_dxi = q * (1.0 - 2.0 * xi)
    dnd_dxi = 2.0 * d1 * xi + 2.0 * s * (1.0 - 2.0 * xi) - 2.0 * d0 * om
    dy_dxi = dy * (dnum_dxi * den - num * dden_dxi) / (den * den)
    dL_dxi = dnd_dxi / nd - 2.0 * dden_dxi / den
    dden_ds = 1.0 - 2.0 * xi * om
    dy_ds = dy * (xi * xi * den - num * dden_ds) / (den * den)
    dL_ds = 2.0 / s + 2.0 * xi * om / nd - 2.0 * dden_ds / den
    dy_dd0 = dy * (xi * om * den - num * xi * om) / (den * den)
    dL_dd0 = om * om / nd - 2.0 * xi * om / den
    dy_dd1 = dy * (-num * xi * om) / (den * den)
    dL_dd1 = xi * xi / nd - 2.0 * xi * om / den
    xibar = y1bar * dy_dxi + lb * dL_dxi
    sbar = y1bar * dy_ds + lb * dL_ds
    d0bar = y1bar * dy_dd0 + lb * dL_dd0
    d1bar = y1bar * dy_dd1 + lb * dL_dd1
    dybar = y1bar * num / den + sbar / dx  # s = dy/dx
    dxbar = -sbar * s / dx - xibar * xi / dx  # xi = (x - xk)/dx
    xkbar = -xibar / dx - dxbar
    xk1bar = dxbar
    ykbar = y1bar - dybar
    yk1bar = dybar
    x1bar = np.where(inside, xibar / dx, y1bar)
    z = np.zeros_like
    pXbar, pYbar, ddbar = z(pX), z(pY), z(dd)

    def scat(dst, idx, val):
        np.add.at(dst, (idx, np.arange(c)[:, None], np.arange(n)[None, :]), np.where(inside, val, 0.0))

    scat(pXbar, k, xkbar)
    scat(pXbar, k + 1, xk1bar)
    scat(pYbar, k, ykbar)
    scat(pYbar, k + 1, yk1bar)
    scat(ddbar, k, d0bar)
    scat(ddbar, k + 1, d1bar)
    thbar = np.zeros_like(th)

    def knots_bwd(v, pbar):
        # p[j] = -B + 2B * sum_{i<j} sm_i  (j=1..K), p[0] = -B
        v = v - v.max(axis=0, keepdims=True)
        e = np.exp(v)
        sm = e / e.sum(axis=0, keepdims=True)
        # dL/dsm_i = 2B * sum_{j>i} pbar[j]   (j from i+1..K)
        rev = np.cumsum(pbar[1:][::-1], axis=0)[::-1]
        smbar = 2.0 * B * rev
        return sm * (smbar - (smbar * sm).sum(axis=0, keepdims=True))

    thbar[:K] = knots_bwd(th[:K], pXbar)
    thbar[K : 2 * K] = knots_bwd(th[K : 2 * K], pYbar)
    thbar[2 * K :] = ddbar[1:K] * sigmoid(th[2 * K :])
    rawbar = thbar.transpose(1, 0, 2).reshape(c * P, n)
    x2bar = ybar[li.idx_c] + mlp_backward(theta, li.nets[0], acts, rawbar.astype(x.dtype), grad, None)
    xbar = np.empty_like(ybar)
    xbar[li.idx_t] = x1bar
    xbar[li.idx_c] = x2bar
    return xbar


# --------------------------------------------------------------------------
# Composition  (ComposedFunction recursion, reached from src/objectives/elbo.jl:67)
# --------------------------------------------------------------------------


def _layer_fwd(spec, theta, li, x):
    p = theta[li.offset : li.offset + li.nparams]
    if li.kind == "planar":
        return planar_fwd(p, x)
    if li.kind == "radial":
        return radial_fwd(p, x)
    if li.kind == "affine":
        return affine_fwd(theta, li, x)
    if li.kind == "shift":  # Bijectors.Shift: y = x + a, ladj = 0
        return x + p[:, None], np.zeros(x.shape[1], dtype=x.dtype)
    if li.kind == "scale":  # Bijectors.Scale: y = a .* x, ladj = sum(log|a|)
        return x * p[:, None], np.full(x.shape[1], np.log(np.abs(p)).sum(), dtype=x.dtype)
    return rqs_fwd(theta, li, x, spec.K, spec.B)


def _layer_inv(spec, theta, li, y):
    p = theta[li.offset : li.offset + li.nparams]
    if li.kind == "planar":
        return planar_inv(p, y)
    if li.kind == "radial":
        return radial_inv(p, y)
    if li.kind == "affine":
        return affine_inv(theta, li, y)
    if li.kind == "shift":
        return y - p[:, None], np.zeros(y.shape[1], dtype=y.dtype)
    if li.kind == "scale":
        return y / p[:, None], np.full(y.shape[1], -np.log(np.abs(p)).sum(), dtype=y.dtype)
    return rqs_inv(theta, li, y, spec.K, spec.B)


def flow_fwd(spec: FlowSpec, theta, x, keep=False):
    """with_logabsdet_jacobian(flow.transform, xs): the LAST-listed layer is applied
    first, per-sample logdets are summed."""
    layers = layers_flat_order(spec)
    ladj = np.zeros(x.shape[1], dtype=x.dtype)
    states = [x]
    for li in reversed(layers):
        x, l = _layer_fwd(spec, theta, li, x)
        ladj = ladj + l
        states.append(x)
    return (x, ladj, states) if keep else (x, ladj)


def flow_inv(spec: FlowSpec, theta, y):
    """inverse(f o g) = inverse(g) o inverse(f): flat order L1 first."""
    layers = layers_flat_order(spec)
    ladj = np.zeros(y.shape[1], dtype=y.dtype)
    for li in layers:
        y, l = _layer_inv(spec, theta, li, y)
        ladj = ladj + l
    return y, ladj


def _layer_bwd(spec, theta, li, xin, ybar, lbar, grad):
    """Reverse pass of one layer at its input `xin`; parameter gradients are added to `grad`."""
    gp = grad[li.offset : li.offset + li.nparams]
    p = theta[li.offset : li.offset + li.nparams]
    if li.kind == "planar":
        return planar_bwd(p, xin, ybar, lbar, gp)
    if li.kind == "radial":
        return radial_bwd(p, xin, ybar, lbar, gp)
    if li.kind == "affine":
        return affine_bwd(theta, li, xin, ybar, lbar, grad)
    if li.kind == "shift":
        gp += ybar.sum(axis=1)
        return ybar
    if li.kind == "scale":
        gp += (ybar * xin).sum(axis=1) + lbar.sum() / p
        return ybar * p[:, None]
    return rqs_bwd(theta, li, xin, ybar, lbar, grad, spec.K, spec.B)


def flow_bwd(spec: FlowSpec, theta, states, ybar, lbar):
    """Reverse pass through the whole chain.  states from flow_fwd(keep=True).
    Returns (xbar, grad_theta)."""
    layers = layers_flat_order(spec)
    grad = np.zeros_like(theta)
    exec_order = list(reversed(layers))
    for i in range(len(exec_order) - 1, -1, -1):
        ybar = _layer_bwd(spec, theta, exec_order[i], states[i], ybar, lbar, grad)
    return ybar, grad


def _layer_inv_bwd(spec, theta, li, w, wbar, c, grad):
    """Reverse pass of ONE INVERSE layer  v -> w = T^-1(v),  ladj_inv = -ladj_fwd(w; theta),
    at its output w, by the implicit-function theorem on v = T(w; theta):
        dw/dv = J^-1,  dw/dtheta = -J^-1 dT/dtheta          (J = dT/dw).
    With cotangents (wbar, c) of (w, ladj_inv):
        vbar      = J^-T (wbar - c * grad_w ladj_fwd)
        thetabar  = -(dT/dtheta)^T vbar - c * grad_theta ladj_fwd
                  =  the FORWARD layer's reverse pass at w with cotangents (-vbar, -c).
    J^T is assembled column by column from the forward layer's reverse pass and solved densely,
    so this is independent of any closed-form inverse Jacobian (what the device kernels use).
    This is what reverse-mode AD of `loglikelihood` (src/objectives/loglikelihood.jl:26-33,
    differentiated by src/optimize.jl:77,86) evaluates.  Returns vbar."""
    d, n = w.shape
    scratch = np.zeros_like(theta)
    zeros_n = np.zeros(n, dtype=w.dtype)
    D = _layer_bwd(spec, theta, li, w, np.zeros_like(w), np.ones(n, dtype=w.dtype), scratch)
    ap = wbar - c[None, :] * D
    JT = np.empty((n, d, d), dtype=w.dtype)
    for i in range(d):
        e = np.zeros_like(w)
        e[i] = 1.0
        JT[:, :, i] = _layer_bwd(spec, theta, li, w, e, zeros_n, scratch).T
    vbar = np.linalg.solve(JT, ap.T[:, :, None])[:, :, 0].T
    _layer_bwd(spec, theta, li, w, -vbar, -c, grad)
    return vbar


# --------------------------------------------------------------------------
# Base distribution, targets, objectives
# --------------------------------------------------------------------------


def std_normal_logpdf(x):
    """logpdf(MvNormal(zeros(d), I), xs) per column (Distributions.jl)."""
    d = x.shape[0]
    return (-0.5 * d * LOG2PI - 0.5 * (x * x).sum(axis=0)).astype(x.dtype)


def diaggauss_logp(y, mu, var):
    """logpdf(MvNormal(mu, Diagonal(var)), y) per column -- the target of
    test/flow.jl:43-46."""
    r = y - mu[:, None]
    return (-0.5 * (LOG2PI + np.log(var)).sum() - 0.5 * (r * r / var[:, None]).sum(axis=0)).astype(y.dtype)


def diaggauss_grad(y, mu, var):
    return (-(y - mu[:, None]) / var[:, None]).astype(y.dtype)


def banana_logp(y, b, var):
    """example/targets/banana.jl:58-63,77-83."""
    d = y.shape[0]
    y2 = y[1] + b * y[0] ** 2 - var * b
    logz = (np.log(var) / d + LOG2PI) * d / 2.0
    ss = y[0] ** 2 / var + y2**2 + (y[2:] ** 2).sum(axis=0)
    return (-logz - 0.5 * ss).astype(y.dtype)


def banana_grad(y, b, var):
    g = -y.copy()
    y2 = y[1] + b * y[0] ** 2 - var * b
    g[0] = -y[0] / var - 2.0 * b * y[0] * y2
    g[1] = -y2
    return g.astype(y.dtype)


def funnel_logp(y, mu, sigma):
    """Neal's funnel, example/targets/neal_funnel.jl:53-60: y1 ~ N(mu, sigma^2), y_{2..d} | y1 ~ N(0, exp(y1) I)."""
    d = y.shape[0]
    l1 = -0.5 * LOG2PI - np.log(sigma) - 0.5 * ((y[0] - mu) / sigma) ** 2
    l2 = -0.5 * (d - 1) * (LOG2PI + y[0]) - 0.5 * np.exp(-y[0]) * (y[1:] ** 2).sum(axis=0)
    return (l1 + l2).astype(y.dtype)


def funnel_grad(y, mu, sigma):
    """`score`, example/targets/neal_funnel.jl:62-72."""
    d = y.shape[0]
    a = np.exp(-y[0])
    g = -a * y
    g[0] = (mu - y[0]) / sigma**2 - (d - 1) / 2.0 + a * (y[1:] ** 2).sum(axis=0) / 2.0
    return g.astype(y.dtype)


def _warped_parts(y):
    r = np.hypot(y[0], y[1])
    th = np.arctan2(y[1], y[0]) + r / 2.0
    return r, th


def warped_logp(y, s1, s2):
    """WarpedGauss(s1, s2), example/targets/warped_gaussian.jl:51-87: z = phi^-1(y) (rotate the angle by
    r/2), logJ = log r."""
    r, th = _warped_parts(y)
    zx, zy = r * np.cos(th), r * np.sin(th)
    return (-0.5 * (zx**2 / s1**2 + zy**2 / s2**2) - LOG2PI - np.log(s1) - np.log(s2) + np.log(r)).astype(y.dtype)


def warped_grad(y, s1, s2):
    x, yy = y[0], y[1]
    r, th = _warped_parts(y)
    c, sn = np.cos(th), np.sin(th)
    zx, zy = r * c, r * sn
    out = np.empty_like(y)
    for k, (dr, dth) in enumerate(((x / r, -yy / r**2 + x / (2 * r)), (yy / r, x / r**2 + yy / (2 * r)))):
        dzx = dr * c - r * sn * dth
        dzy = dr * sn + r * c * dth
        out[k] = -zx / s1**2 * dzx - zy / s2**2 * dzy + dr / r
    return out.astype(y.dtype)


def _cross_components(mu, sigma):
    """Cross(mu, sigma), example/targets/cross.jl:30-37, components exactly as the code builds them
    (means and per-coordinate standard deviations; equal weights)."""
    means = np.array([[0.0, mu], [-mu, 1.0], [mu, 1.0], [0.0, -mu]])
    stds = np.array([[sigma, 1.0], [1.0, sigma], [1.0, sigma], [sigma, 1.0]])
    return means, stds


def _cross_comp_logs(y, mu, sigma):
    means, stds = _cross_components(mu, sigma)
    logs = []
    for m, sd in zip(means, stds):
        logs.append(-LOG2PI - np.log(sd[0]) - np.log(sd[1])
                    - 0.5 * (((y[0] - m[0]) / sd[0]) ** 2 + ((y[1] - m[1]) / sd[1]) ** 2))
    return np.stack(logs), means, stds


def cross_logp(y, mu, sigma):
    logs, _, _ = _cross_comp_logs(y, mu, sigma)
    mx = logs.max(axis=0)
    return (np.log(0.25) + mx + np.log(np.exp(logs - mx).sum(axis=0))).astype(y.dtype)


def cross_grad(y, mu, sigma):
    logs, means, stds = _cross_comp_logs(y, mu, sigma)
    w = np.exp(logs - logs.max(axis=0))
    w = w / w.sum(axis=0)
    g = np.zeros_like(y)
    for k in range(4):
        g[0] += w[k] * (-(y[0] - means[k, 0]) / stds[k, 0] ** 2)
        g[1] += w[k] * (-(y[1] - means[k, 1]) / stds[k, 1] ** 2)
    return g.astype(y.dtype)


def target_logp(target, y):
    if target[0] == "funnel":
        return funnel_logp(y, target[1], target[2])
    if target[0] == "warped":
        return warped_logp(y, target[1], target[2])
    if target[0] == "cross":
        return cross_logp(y, target[1], target[2])
    if target[0] == "diaggauss":
        return diaggauss_logp(y, target[1], target[2])
    if target[0] == "banana":
        return banana_logp(y, target[1], target[2])
    raise ValueError(target[0])


def target_grad(target, y):
    if target[0] == "funnel":
        return funnel_grad(y, target[1], target[2])
    if target[0] == "warped":
        return warped_grad(y, target[1], target[2])
    if target[0] == "cross":
        return cross_grad(y, target[1], target[2])
    if target[0] == "diaggauss":
        return diaggauss_grad(y, target[1], target[2])
    if target[0] == "banana":
        return banana_grad(y, target[1], target[2])
    raise ValueError(target[0])


def batched_elbos(spec, theta, target, xs):
    """_batched_elbos: src/objectives/elbo.jl:65-70."""
    ys, ladj = flow_fwd(spec, theta, xs)
    return target_logp(target, ys) - std_normal_logpdf(xs) + ladj


def elbo_batch(spec, theta, target, xs):
    """elbo_batch(flow, logp, xs): src/objectives/elbo.jl:89-92."""
    return batched_elbos(spec, theta, target, xs).mean()


def elbo(spec, theta, target, xs):
    """elbo(flow, logp, xs): per-column map of elbo_single_sample,
    src/objectives/elbo.jl:4-7,31-34."""
    vals = [batched_elbos(spec, theta, target, xs[:, j : j + 1])[0] for j in range(xs.shape[1])]
    return np.mean(vals)


def loglikelihood(spec, theta, ys):
    """src/objectives/loglikelihood.jl:26-33: mean_j logpdf(flow, y_j), with
    logpdf(td, y) = logpdf(td.dist, x) + ladj_inv (Bijectors)."""
    xs, ladj = flow_inv(spec, theta, ys)
    return (std_normal_logpdf(xs) + ladj).mean()


def neg_loglik_value_and_grad(spec, theta, ys, n_global=None):
    """loss(theta) = -loglikelihood(flow, ys) and its gradient: `train_flow(loglikelihood, flow, ys)`
    (src/NormalizingFlows.jl:69 with vo = loglikelihood; gradient by src/optimize.jl:77,86).
    The chain is inverted once, then walked in forward execution order with _layer_inv_bwd."""
    n = ys.shape[1]
    ng = n if n_global is None else n_global
    layers = layers_flat_order(spec)
    z, ladj = flow_inv(spec, theta, ys)
    loss = -(std_normal_logpdf(z) + ladj).sum() / ng
    a = z / ng  # d loss / dz through -log q0(z)
    c = np.full(n, -1.0 / ng, dtype=ys.dtype)
    grad = np.zeros_like(theta)
    w = z
    for li in reversed(layers):
        a = _layer_inv_bwd(spec, theta, li, w, a, c, grad)
        w, _ = _layer_fwd(spec, theta, li, w)
    return loss, grad


def neg_elbo_value_and_grad(spec, theta, target, xs):
    """loss(theta) = -elbo_batch (src/NormalizingFlows.jl:69) and its gradient
    (what _value_and_gradient returns, src/optimize.jl:12-14,86)."""
    n = xs.shape[1]
    ys, ladj, states = flow_fwd(spec, theta, xs, keep=True)
    elbos = target_logp(target, ys) - std_normal_logpdf(xs) + ladj
    loss = -elbos.mean()
    ybar = (-target_grad(target, ys) / n).astype(xs.dtype)
    lbar = np.full(n, -1.0 / n, dtype=xs.dtype)
    _, grad = flow_bwd(spec, theta, states, ybar, lbar)
    return loss, grad


def adam_update(theta, g, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """Optimisers.Adam (0.4): t is the 1-based step count AFTER this update."""
    m[:] = b1 * m + (1.0 - b1) * g
    v[:] = b2 * v + (1.0 - b2) * g * g
    mhat = m / (1.0 - b1**t)
    vhat = v / (1.0 - b2**t)
    theta[:] = theta - lr * mhat / (np.sqrt(vhat) + eps)
    return theta


# --------------------------------------------------------------------------
# Base sampler specification: Philox4x32-10 + Box-Muller (this build's own
# sampler; the reference's Xoshiro/Ziggurat stream is not reproducible outside
# Julia and no reference test depends on specific draws -- SURVEY.md App. B).
# --------------------------------------------------------------------------

_PH_M0, _PH_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PH_W0, _PH_W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 (Salmon et al. 2011).  All inputs uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint32).copy() for c in (c0, c1, c2, c3)]
    k0 = np.asarray(k0, dtype=np.uint32)
    k1 = np.asarray(k1, dtype=np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _PH_M0 * c0.astype(np.uint64)
            p1 = _PH_M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = k0 + _PH_W0
            k1 = k1 + _PH_W1
    return c0, c1, c2, c3


def base_sample(d: int, n: int, seed: int, sample_offset: int = 0, stream: int = 0, dtype=np.float64, precision: str = "f32"):
    """x ~ N(0, I) of shape (d, n).  Features are generated four at a time:
    counter = (global_sample_lo, global_sample_hi, feature_group, stream), key = (seed_lo, seed_hi).
      precision "f32" (the Float32 device stream; the default, also as float64 reference values of it):
        u = ((r >> 9) + 0.5) * 2^-23 (exactly representable in fp32); Box-Muller on (u0,u1) -> features 4g, 4g+1 and
        (u2,u3) -> 4g+2, 4g+3.
      precision "f64" (the Float64 device stream): 53-bit uniforms u = ((hi << 21 | lo >> 11) + 0.5) * 2^-53 from word
        pairs (r0,r1), (r2,r3); the call with counter word 2 = g gives features 4g, 4g+1, the call with g | 2^31 gives
        4g+2, 4g+3.
    The global sample index makes the batch invariant to how it is sharded over GPUs."""
    ng = (d + 3) // 4
    j = np.arange(n, dtype=np.uint64) + np.uint64(sample_offset)
    jj, gg = np.meshgrid(j, np.arange(ng, dtype=np.uint32), indexing="ij")
    c0 = (jj & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    c1 = (jj >> np.uint64(32)).astype(np.uint32)
    c3 = np.full(jj.shape, stream, dtype=np.uint32)
    k0, k1 = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    out = np.empty((n, ng, 4), dtype=np.float64)
    if precision == "f32":
        r = philox4x32_10(c0, c1, gg, c3, k0, k1)
        u = [((ri >> np.uint32(9)).astype(np.float64) + 0.5) * (2.0**-23) for ri in r]
        for a in (0, 1):
            rad = np.sqrt(-2.0 * np.log(u[2 * a]))
            ang = 2.0 * np.pi * u[2 * a + 1]
            out[:, :, 2 * a] = rad * np.cos(ang)
            out[:, :, 2 * a + 1] = rad * np.sin(ang)
    elif precision == "f64":
        for a, word2 in ((0, gg), (1, gg | np.uint32(0x80000000))):
            r = philox4x32_10(c0, c1, word2, c3, k0, k1)
            u = []
            for hi, lo in ((r[0], r[1]), (r[2], r[3])):
                bits = (hi.astype(np.uint64) << np.uint64(21)) | (lo.astype(np.uint64) >> np.uint64(11))
                u.append((bits.astype(np.float64) + 0.5) * (2.0**-53))
            rad = np.sqrt(-2.0 * np.log(u[0]))
            ang = 2.0 * np.pi * u[1]
            out[:, :, 2 * a] = rad * np.cos(ang)
            out[:, :, 2 * a + 1] = rad * np.sin(ang)
    else:
        raise ValueError(precision)
    return out.reshape(n, ng * 4)[:, :d].T.copy().astype(dtype)


# --------------------------------------------------------------------------
# Heterogeneous compositions: create_flow((L1, ..., Ln), q0) = transformed(q0, reduce(o, Ls)) for ANY list of
# bijectors (src/flows/utils.jl:23-26).  `specs` are single-family FlowSpecs in flat order (first = outermost =
# applied last); theta is their parameter vectors concatenated (the order destructure walks the composition).
# --------------------------------------------------------------------------


def _comp_slices(specs):
    offs, o_ = [], 0
    for sp in specs:
        offs.append((o_, o_ + param_count(sp)))
        o_ += param_count(sp)
    return offs


def comp_fwd(specs, theta, x, keep=False):
    sl = _comp_slices(specs)
    ladj = np.zeros(x.shape[1], dtype=x.dtype)
    inputs = [None] * len(specs)
    for s in range(len(specs) - 1, -1, -1):
        inputs[s] = x
        x, l = flow_fwd(specs[s], theta[sl[s][0] : sl[s][1]], x)
        ladj = ladj + l
    return (x, ladj, inputs) if keep else (x, ladj)


def comp_inv(specs, theta, y):
    sl = _comp_slices(specs)
    ladj = np.zeros(y.shape[1], dtype=y.dtype)
    for s in range(len(specs)):
        y, l = flow_inv(specs[s], theta[sl[s][0] : sl[s][1]], y)
        ladj = ladj + l
    return y, ladj


def comp_neg_elbo_value_and_grad(specs, theta, target, xs):
    n = xs.shape[1]
    sl = _comp_slices(specs)
    ys, ladj, inputs = comp_fwd(specs, theta, xs, keep=True)
    loss = -(target_logp(target, ys) - std_normal_logpdf(xs) + ladj).mean()
    gbar = (-target_grad(target, ys) / n).astype(xs.dtype)
    lbar = np.full(n, -1.0 / n, dtype=xs.dtype)
    grad = np.zeros_like(theta)
    for s in range(len(specs)):
        th = theta[sl[s][0] : sl[s][1]]
        _, _, states = flow_fwd(specs[s], th, inputs[s], keep=True)
        gbar, g = flow_bwd(specs[s], th, states, gbar, lbar)
        grad[sl[s][0] : sl[s][1]] = g
    return loss, grad


# --------------------------------------------------------------------------
# General MvNormal(mu, Sigma) base distributions: _device_specific_rand(rng, ::MvNormal, n) draws mu + L eps
# (Distributions' unwhiten; device version ext/NormalizingFlowsCUDAExt.jl:43-48, dense Sigma exercised by
# test/ext/CUDA/cuda.jl:33-45) and logpdf(flow.dist, xs) enters the ELBO at src/objectives/elbo.jl:6,68.
# base = None (standard normal) | ("diag", mu, sigma) | ("dense", mu, L) with Sigma = L L'.
# --------------------------------------------------------------------------


def base_unwhiten(base, eps):
    if base is None:
        return eps
    kind, mu, sc = base
    return mu[:, None] + (sc[:, None] * eps if kind == "diag" else sc @ eps)


def base_logpdf(base, x):
    """logpdf(MvNormal(mu, Sigma), x) per column (Distributions: -(d log 2pi + logdet Sigma)/2 - sqmahal/2)."""
    if base is None:
        return std_normal_logpdf(x)
    kind, mu, sc = base
    r = x - mu[:, None]
    if kind == "diag":
        z, logdet = r / sc[:, None], np.log(sc).sum()
    else:
        import scipy.linalg

        z, logdet = scipy.linalg.solve_triangular(sc, r, lower=True), np.log(np.diag(sc)).sum()
    return -0.5 * x.shape[0] * LOG2PI - logdet - 0.5 * (z * z).sum(axis=0)


def batched_elbos_base(spec, theta, target, xs, base):
    ys, ladj = flow_fwd(spec, theta, xs)
    return target_logp(target, ys) - base_logpdf(base, xs) + ladj


def neg_elbo_value_and_grad_base(spec, theta, target, xs, base):
    """-elbo_batch with a general base: q0 is not trainable, so only the value's log q0 term differs."""
    loss, grad = neg_elbo_value_and_grad(spec, theta, target, xs)
    return loss + (base_logpdf(base, xs) - std_normal_logpdf(xs)).mean(), grad


def loglikelihood_base(spec, theta, ys, base):
    xs, ladj = flow_inv(spec, theta, ys)
    return (base_logpdf(base, xs) + ladj).mean()


def base_score(base, x):
    """d logpdf(base, x) / dx per column: -Sigma^-1 (x - mu)."""
    if base is None:
        return -x
    kind, mu, sc = base
    r = x - mu[:, None]
    if kind == "diag":
        return -r / (sc * sc)[:, None]
    import scipy.linalg

    z = scipy.linalg.solve_triangular(sc, r, lower=True)
    return -scipy.linalg.solve_triangular(sc.T, z, lower=False)


def comp_neg_loglik_value_and_grad(specs, theta, ys, base=None, n_global=None):
    """-loglikelihood(flow, ys) and its gradient (src/objectives/loglikelihood.jl:26-33 through src/optimize.jl:77,86)
    for a composition of homogeneous segments (one segment: a single-family flow) over any MvNormal base:
    the segments are inverted first to last, log q0 is the base's own density, and the reverse pass walks the
    inverse chain backwards -- last segment first, each one's layers in forward execution order."""
    n = ys.shape[1]
    ng = n if n_global is None else n_global
    sl = _comp_slices(specs)
    outs, w, ladj = [], ys, np.zeros(n, dtype=ys.dtype)
    for s in range(len(specs)):
        w, l = flow_inv(specs[s], theta[sl[s][0] : sl[s][1]], w)
        outs.append(w)
        ladj = ladj + l
    loss = -(base_logpdf(base, w) + ladj).sum() / ng
    a = (-base_score(base, w) / ng).astype(ys.dtype)
    c = np.full(n, -1.0 / ng, dtype=ys.dtype)
    grad = np.zeros_like(theta)
    for s in range(len(specs) - 1, -1, -1):
        th = theta[sl[s][0] : sl[s][1]]
        g = np.zeros_like(th)
        w = outs[s]
        for li in reversed(layers_flat_order(specs[s])):
            a = _layer_inv_bwd(specs[s], th, li, w, a, c, g)
            w, _ = _layer_fwd(specs[s], th, li, w)
        grad[sl[s][0] : sl[s][1]] = g
    return loss, grad


# --------------------------------------------------------------------------
# Hamiltonian flow (example/demo_hamiltonian_flow.jl:27-146): a mean-field Gaussian reference on the joint
# z = [x; rho] followed by n blocks (momentum Shift o Scale) o LeapFrog(L steps, per-dimension step sizes
# exp(log_eps), score of the target).  theta = [shift0(2D), scale0(2D), then per block: shift_rho(D),
# scale_rho(D), log_eps(D)]: `transformed(q0, T)` destructures its dist (q0's own Shift o Scale, :135-137)
# before its transform, and each block is ComposedFunction(outer = momentum layer, inner = LeapFrog) (:144).
# --------------------------------------------------------------------------


def target_hvp(target, x, v):
    """Hessian of log p at x (columns) applied to v: what reverse-mode AD through `score` needs."""
    kind = target[0]
    if kind == "diaggauss":
        return -v / target[2][:, None]
    if kind == "banana":
        b, s = target[1], target[2]
        y2 = x[1] + b * x[0] ** 2 - s * b
        out = -v.copy()
        h11 = -1.0 / s - 2.0 * b * y2 - 4.0 * b * b * x[0] ** 2
        h12 = -2.0 * b * x[0]
        out[0] = h11 * v[0] + h12 * v[1]
        out[1] = h12 * v[0] - v[1]
        return out
    if kind == "funnel":
        mu, sg = target[1], target[2]
        a = np.exp(-x[0])
        s2 = (x[1:] ** 2).sum(axis=0)
        out = np.empty_like(v)
        out[0] = (-1.0 / sg**2 - 0.5 * a * s2) * v[0] + a * (x[1:] * v[1:]).sum(axis=0)
        out[1:] = a * x[1:] * v[0] - a * v[1:]
        return out
    raise ValueError(f"no Hessian-vector product for target {kind!r}")


def hflow_param_count(D: int, n: int) -> int:
    return 4 * D + 3 * D * n


def _hflow_views(D, n, theta):
    sh0, sc0 = theta[: 2 * D], theta[2 * D : 4 * D]
    blocks = []
    off = 4 * D
    for _ in range(n):
        blocks.append((theta[off : off + D], theta[off + D : off + 2 * D], theta[off + 2 * D : off + 3 * D]))
        off += 3 * D
    return sh0, sc0, blocks


def _leapfrog(target, eps, L, x, v, keep=None):
    """_leapfrog of demo_hamiltonian_flow.jl:49-60 (eps is a column of per-dimension step sizes)."""
    g = target_grad(target, x)
    v = v + 0.5 * eps * g
    if keep is not None:
        keep.append((x, v, g))
    for _ in range(L - 1):
        x = x + eps * v
        g = target_grad(target, x)
        v = v + eps * g
        if keep is not None:
            keep.append((x, v, g))
    xl = x + eps * v
    gl = target_grad(target, xl)
    vl = v + 0.5 * eps * gl
    if keep is not None:
        keep.append((xl, vl, gl))
    return xl, vl


def hflow_fwd(D, n, L, theta, target, x0, keep=None):
    """Base draws x0 (2D x N) -> (z, ladj).  Blocks execute last-listed first (utils.jl:23-26)."""
    sh0, sc0, blocks = _hflow_views(D, n, theta)
    z = sh0[:, None] + sc0[:, None] * x0
    ladj = np.full(x0.shape[1], np.log(np.abs(sc0)).sum())
    for bi in range(n - 1, -1, -1):
        shr, scr, leps = blocks[bi]
        eps = np.exp(leps)[:, None]
        tr = [] if keep is not None else None
        x, v = _leapfrog(target, eps, L, z[:D], z[D:], tr)
        if keep is not None:
            keep.append((bi, z.copy(), tr, v.copy()))
        z = np.concatenate([x, shr[:, None] + scr[:, None] * v], axis=0)
        ladj = ladj + np.log(np.abs(scr)).sum()
    return z, ladj


def hflow_inv(D, n, L, theta, target, z):
    """Inverse chain: momentum layers undone, LeapFrog run with -eps (demo :74-84); returns (x0, ladj_inv)."""
    sh0, sc0, blocks = _hflow_views(D, n, theta)
    ladj = np.zeros(z.shape[1])
    for bi in range(n):
        shr, scr, leps = blocks[bi]
        v = (z[D:] - shr[:, None]) / scr[:, None]
        ladj = ladj - np.log(np.abs(scr)).sum()
        x, v = _leapfrog(target, -np.exp(leps)[:, None], L, z[:D], v)
        z = np.concatenate([x, v], axis=0)
    ladj = ladj - np.log(np.abs(sc0)).sum()
    return (z - sh0[:, None]) / sc0[:, None], ladj


def hflow_joint_logp(D, target, z):
    """logp_joint of the demo (:121-128): log p(x) + log N(rho; 0, I)."""
    return target_logp(target, z[:D]) + std_normal_logpdf(z[D:])


def hflow_neg_elbo_value_and_grad(D, n, L, theta, target, x0):
    """loss = -mean_j [logp_joint(z_j) - log N(x0_j) + ladj_j] and its gradient, hand-derived: the LeapFrog
    reverse pass needs Hessian-vector products of log p (target_hvp)."""
    N = x0.shape[1]
    keep = []
    z, ladj = hflow_fwd(D, n, L, theta, target, x0, keep)
    loss = -np.mean(hflow_joint_logp(D, target, z) - std_normal_logpdf(x0) + ladj)
    sh0, sc0, blocks = _hflow_views(D, n, theta)
    grad = np.zeros_like(theta)
    zbar = np.concatenate([target_grad(target, z[:D]), -z[D:]], axis=0) * (-1.0 / N)
    lbar = -1.0  # d loss / d (sum_j ladj_j / N) per sample, constant terms handled per parameter below
    off_of = lambda bi: 4 * D + 3 * D * bi  # noqa: E731
    for bi, zin, tr, vout in keep[::-1]:  # reverse of execution order = flat order bi = 0 .. n-1
        shr, scr, leps = blocks[bi]
        eps = np.exp(leps)[:, None]
        o0 = off_of(bi)
        xbar, rbar = zbar[:D].copy(), zbar[D:].copy()
        # momentum layer: rho' = shr + scr * v
        grad[o0 : o0 + D] += rbar.sum(axis=1)
        grad[o0 + D : o0 + 2 * D] += (rbar * vout).sum(axis=1) + lbar / scr
        vbar = rbar * scr[:, None]
        # LeapFrog reverse (states tr[s] = (x_s, v_s, g(x_s)), s = 0..L)
        ebar = np.zeros((D, N))
        xs_, vs_, gs_ = tr[L]
        xbar = xbar + target_hvp(target, xs_, 0.5 * eps * vbar)
        ebar += 0.5 * vbar * gs_
        vprev = tr[L - 1][1]
        ebar += xbar * vprev
        vbar = vbar + eps * xbar
        for s in range(L - 1, 0, -1):
            xs_, vs_, gs_ = tr[s]
            xbar = xbar + target_hvp(target, xs_, eps * vbar)
            ebar += vbar * gs_
            vprev = tr[s - 1][1]
            ebar += xbar * vprev
            vbar = vbar + eps * xbar
        xs_, vs_, gs_ = tr[0]
        xbar = xbar + target_hvp(target, xs_, 0.5 * eps * vbar)
        ebar += 0.5 * vbar * gs_
        grad[o0 + 2 * D : o0 + 3 * D] += (ebar * eps).sum(axis=1)
        zbar = np.concatenate([xbar, vbar], axis=0)
    grad[: 2 * D] += zbar.sum(axis=1)
    grad[2 * D : 4 * D] += (zbar * x0).sum(axis=1) + lbar / sc0
    return float(loss), grad
