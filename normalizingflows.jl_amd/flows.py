"""Host-side mirror of the reference's flow constructors and of the Bijectors.jl surface the
hot path uses.  All arithmetic is done by libnfhip.so (HIP, gfx950); this module only owns
descriptors, the flat parameter vector and argument marshalling.

Reference files mirrored (paths in the reference checkout):
  src/flows/utils.jl            create_flow, fnn layout
  src/flows/planar_radial.jl    planarflow, radialflow
  src/flows/realnvp.jl          AffineCoupling, RealNVP_layer, realnvp
  src/flows/neuralspline.jl     NeuralSplineCoupling, NSF_layer, nsf
  src/NormalizingFlows.jl       _device_specific_rand (:94-127)

Array convention = the reference's: a batch is a (d, N) matrix with one sample per COLUMN,
stored column-major (here: a torch tensor of shape (d, N) whose transpose is contiguous).
A vector of shape (d,) is a single sample and gives scalar log-determinants
(src/flows/realnvp.jl:69-75).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import Base, FlowDesc, NFHipError, Target, check, context_for


def _dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return _lib.NF_DTYPE_F32
    if dt == torch.float64:
        return _lib.NF_DTYPE_F64
    raise NFHipError(f"unsupported parameter type {dt}")


def new_batch(d: int, n: int, dtype, device) -> torch.Tensor:
    """(d, n) matrix in the reference's column-major layout."""
    return torch.empty((n, d), dtype=dtype, device=device).t()


def as_batch(x: torch.Tensor):
    """Returns (matrix (d, N) column-major, was_vector)."""
    vec = x.dim() == 1
    if vec:
        x = x.reshape(-1, 1)
    if x.dim() != 2:
        raise NFHipError("expected a vector (d,) or a matrix (d, N)")
    if not x.t().is_contiguous():
        x = x.t().contiguous().t()
    return x, vec


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


# --------------------------------------------------------------------------------------
# base distribution and RNG seam
# --------------------------------------------------------------------------------------
class MvNormal:
    """Distributions.MvNormal as the reference uses it for q0.

    MvNormal(d)            -> MvNormal(zeros(T, d), I): the base of every reference flow config (test/flow.jl:9,
                              example/demo_planar_flow.jl:24); the form the fused kernels draw in registers.
    MvNormal(mu, Sigma)    -> general base: Sigma a vector of VARIANCES (Diagonal(Sigma)) or a d x d covariance matrix
                              (Cholesky-factored here, once, on the host side of the boundary).  Draws are mu + L eps
                              (`unwhiten`, ext/NormalizingFlowsCUDAExt.jl:43-48; dense Sigma: test/ext/CUDA/cuda.jl:33-45).
    q0 is a leaf of destructure (@leaf MvNormal): none of this is trainable."""

    def __init__(self, mu_or_d, cov: Optional[torch.Tensor] = None):
        if cov is None and not torch.is_tensor(mu_or_d):
            self.d, self.mu, self.scale, self.c = int(mu_or_d), None, None, None
            return
        mu = mu_or_d
        if cov is None:
            raise NFHipError("MvNormal(mu, Sigma): give the covariance (a vector of variances or a matrix)")
        if mu.dim() != 1 or cov.shape[0] != mu.numel() or cov.dtype != mu.dtype or cov.device != mu.device:
            raise NFHipError("MvNormal(mu, Sigma): mu (d,), Sigma (d,) or (d, d), one element type and device")
        self.d = mu.numel()
        self.mu = mu.contiguous()
        if cov.dim() == 1:
            if not bool((cov > 0).all()):
                raise NFHipError("MvNormal: variances must be positive")
            self.scale = cov.sqrt().contiguous()
            kind, logdet = _lib.NF_BASE_DIAG, float(self.scale.double().log().sum())
        else:
            L = torch.linalg.cholesky(cov)  # raises for a matrix that is not positive definite, as PDMats does
            self.scale = L.t().contiguous()  # column-major lower triangle: element (i, k) at [k * d + i]
            kind, logdet = _lib.NF_BASE_DENSE, float(torch.diagonal(L).double().log().sum())
        self.c = Base(kind, self.mu.data_ptr(), self.scale.data_ptr(), logdet)

    @property
    def standard(self) -> bool:
        return self.c is None

    def base_ptr(self):
        return None if self.c is None else C.byref(self.c)

    def __len__(self):
        return self.d


class PhiloxRNG:
    """Device RNG handle (the analogue of CUDA.RNG in ext/NormalizingFlowsCUDAExt.jl).
    Philox4x32-10 keyed by `seed`; every draw call consumes one stream id, so successive
    calls give independent batches, and `sample_offset` places a shard inside a global batch."""

    def __init__(self, seed: int = 0, sample_offset: int = 0):
        self.seed = int(seed)
        self.stream = 0
        self.sample_offset = int(sample_offset)

    def next_stream(self) -> int:
        s = self.stream
        self.stream += 1
        return s


def device_specific_rand(rng: PhiloxRNG, dist, n: Optional[int] = None, *, device=None, dtype=torch.float32):
    """NormalizingFlows._device_specific_rand(rng, dist[, n])  (src/NormalizingFlows.jl:94-127).
    `dist` is an MvNormal base or a Flow (then base draws are pushed through the transform,
    as rand(td, n) does)."""
    if isinstance(dist, Flow):
        flow = dist
        nn = 1 if n is None else int(n)
        y = new_batch(flow.dist.d, nn, flow.theta.dtype, flow.theta.device)
        ctx = flow.ctx
        check(ctx.lib.nf_flow_rand(ctx.ptr, C.byref(flow.desc), _ptr(flow.theta), nn, rng.seed, rng.sample_offset,
                                   rng.next_stream(), _ptr(y)))
        return y[:, 0] if n is None else y
    if not dist.standard:
        device, dtype = dist.mu.device, dist.mu.dtype
    device = torch.device(device if device is not None else "cuda")
    nn = 1 if n is None else int(n)
    x = new_batch(dist.d, nn, dtype, device)
    ctx = context_for(device)
    check(ctx.lib.nf_base_rand(ctx.ptr, _dtype_code(dtype), dist.base_ptr(), dist.d, nn, rng.seed, rng.sample_offset,
                               rng.next_stream(), _ptr(x), _ptr(None)))
    return x[:, 0] if n is None else x


# --------------------------------------------------------------------------------------
# flows
# --------------------------------------------------------------------------------------
class Transform:
    """flow.transform: a composed bijector.  `inverse(t)` gives the Inverse{...} view."""

    def __init__(self, flow: "Flow", inverted: bool = False, layer: Optional[int] = None):
        self.flow = flow
        self.inverted = inverted
        self.layer = layer

    def __call__(self, x):
        return with_logabsdet_jacobian(self, x)[0]


class Flow:
    """Bijectors.TransformedDistribution: `dist` (base) + `transform`.  `theta` is the flat
    parameter vector of Optimisers.destructure(flow) (src/NormalizingFlows.jl:67)."""

    def __init__(self, kind: str, dist: MvNormal, nlayers: int, hdims: Sequence[int] = (), K: int = 0, B: float = 0.0,
                 dtype=torch.float32, device="cuda", theta: Optional[torch.Tensor] = None, score=None):
        self.kind, self.dist, self.nlayers = kind, dist, int(nlayers)
        self.score = score  # Hamiltonian flows: the target whose score drives LeapFrog (kept alive here)
        self.hdims, self.K, self.B = tuple(int(h) for h in hdims), int(K), float(B)
        if len(self.hdims) > _lib.NF_MAX_HIDDEN:
            raise NFHipError("at most 4 hidden layers")
        self.desc = FlowDesc()
        self.desc.kind = _lib.NF_KIND[kind]
        self.desc.dtype = _dtype_code(dtype)
        self.desc.d = dist.d
        self.desc.nlayers = self.nlayers
        self.desc.n_hidden = len(self.hdims)
        for i, h in enumerate(self.hdims):
            self.desc.hdims[i] = h
        self.desc.K = self.K
        self.desc.B = self.B
        self.desc.score = C.addressof(score.c) if score is not None else None
        if not dist.standard:
            if dist.mu.dtype != dtype:
                raise NFHipError(f"base distribution is {dist.mu.dtype}, flow parameters are {dtype}")
            self.desc.base = C.addressof(dist.c)  # kept alive by self.dist
        self.P = int(_lib.load_library().nf_param_count(C.byref(self.desc)))
        if self.P < 0:
            check(self.P)
        dev = torch.device(device)
        self.theta = theta if theta is not None else torch.zeros(self.P, dtype=dtype, device=dev)
        if self.theta.numel() != self.P:
            raise NFHipError(f"theta has {self.theta.numel()} entries, flow has {self.P} parameters")
        self.transform = Transform(self)

    # Optimisers.destructure(flow) -> (theta_flat, re)
    def destructure(self):
        def re(theta):
            return self.with_theta(theta)

        return self.theta.clone(), re

    def with_theta(self, theta: torch.Tensor) -> "Flow":
        return Flow(self.kind, self.dist, self.nlayers, self.hdims, self.K, self.B, self.theta.dtype,
                    self.theta.device, theta, self.score)

    @property
    def ctx(self):
        return context_for(self.theta.device)


class CompositeFlow(Flow):
    """create_flow((L1, ..., Ln), q0) with bijectors of different families (src/flows/utils.jl:23-26: any list of
    bijectors composes).  `segments` are single-family flows in flat order (the first is the outermost = applied
    last); theta is their thetas concatenated -- the order Optimisers.destructure walks the composition."""

    def __init__(self, segments: Sequence[Flow], dist: MvNormal, theta: Optional[torch.Tensor] = None):
        if not segments:
            raise NFHipError("create_flow: empty layer list")
        dt, dev = segments[0].theta.dtype, segments[0].theta.device
        for f in segments:
            if isinstance(f, CompositeFlow) or f.kind == "hamiltonian":
                raise NFHipError("create_flow: segments must be single-family flows")
            if f.dist.d != dist.d or f.theta.dtype != dt or f.theta.device != dev:
                raise NFHipError("create_flow: every layer must share the dimension, element type and device")
        self.kind, self.dist, self.nlayers = "composite", dist, 1
        self.hdims, self.K, self.B, self.score = (), 0, 0.0, None
        self.segments = list(segments)
        self._seg_descs = (FlowDesc * len(segments))()
        for i, f in enumerate(segments):
            C.memmove(C.addressof(self._seg_descs[i]), C.addressof(f.desc), C.sizeof(FlowDesc))
            self._seg_descs[i].base = None  # q0 belongs to the composition
        self.desc = FlowDesc()
        self.desc.kind = _lib.NF_KIND["composite"]
        self.desc.dtype = _dtype_code(dt)
        self.desc.d = dist.d
        self.desc.nlayers = 1
        self.desc.nsegments = len(segments)
        self.desc.segments = C.addressof(self._seg_descs)
        if not dist.standard:
            if dist.mu.dtype != dt:
                raise NFHipError(f"base distribution is {dist.mu.dtype}, flow parameters are {dt}")
            self.desc.base = C.addressof(dist.c)
        self.P = int(_lib.load_library().nf_param_count(C.byref(self.desc)))
        if self.P < 0:
            check(self.P)
        self.theta = theta if theta is not None else torch.cat([f.theta for f in segments])
        if self.theta.numel() != self.P:
            raise NFHipError(f"theta has {self.theta.numel()} entries, flow has {self.P} parameters")
        self.transform = Transform(self)

    def with_theta(self, theta: torch.Tensor) -> "CompositeFlow":
        return CompositeFlow(self.segments, self.dist, theta)


def create_flow(Ls: Sequence[Flow], q0: MvNormal) -> Flow:
    """create_flow(Ls, q0) = transformed(q0, reduce(o, Ls))  (src/flows/utils.jl:23-26).  `Ls` are flows built by the
    constructors below (each contributes its transform); one element returns that flow on q0, several compose."""
    Ls = list(Ls)
    if len(Ls) == 1 and not isinstance(Ls[0], CompositeFlow) and Ls[0].dist is q0:
        return Ls[0]
    return CompositeFlow(Ls, q0)


def inverse(t: Transform) -> Transform:
    """Bijectors.inverse"""
    return Transform(t.flow, not t.inverted, t.layer)


def layer(flow: Flow, index: int) -> Transform:
    """The `index`-th bijector of the composition in FLAT order (0 = outermost = applied last)."""
    return Transform(flow, False, index)


def with_logabsdet_jacobian(t: Transform, x: torch.Tensor):
    """Bijectors.with_logabsdet_jacobian(t, x) -> (y, logabsdetjac)
    (src/flows/realnvp.jl:69-110, src/flows/neuralspline.jl:94-140; ComposedFunction recursion
    reached from src/objectives/elbo.jl:67)."""
    flow = t.flow
    xm, vec = as_batch(x.to(flow.theta.dtype))
    d, n = xm.shape
    if d != flow.dist.d:
        raise NFHipError(f"dimension mismatch: flow has d={flow.dist.d}, input has {d}")
    y = new_batch(d, n, xm.dtype, xm.device)
    ladj = torch.empty(n, dtype=xm.dtype, device=xm.device)
    ctx = flow.ctx
    if t.layer is None:
        fn = ctx.lib.nf_flow_inv if t.inverted else ctx.lib.nf_flow_fwd
        check(fn(ctx.ptr, C.byref(flow.desc), _ptr(flow.theta), _ptr(xm), n, _ptr(y), _ptr(ladj)))
    else:
        check(ctx.lib.nf_layer_apply(ctx.ptr, C.byref(flow.desc), t.layer, int(t.inverted), _ptr(flow.theta), _ptr(xm),
                                     n, _ptr(y), _ptr(ladj)))
    if vec:
        return y[:, 0], ladj[0]
    return y, ladj


def rrule_with_logabsdet_jacobian(t: Transform, x: torch.Tensor):
    """ChainRulesCore.rrule(with_logabsdet_jacobian, t, x) -> ((y, logabsdetjac), pullback): the forward keeps its tape in
    a device buffer owned by the returned closure (nf_flow_fwd_keep), `pullback(ybar, lbar) -> (xbar, gtheta)` consumes it
    (nf_flow_bwd_kept).  This is what lets an arbitrary `logp` closure train through the library: the user's AD supplies
    ybar, the tape supplies the forward's own activations (the reference: Zygote on the forward's tape,
    src/optimize.jl:12-14; the rrule mechanism MonotonicSplines uses, test/ad.jl:126-127).  Whole forward transforms
    only (no Inverse, no single layer)."""
    flow = t.flow
    if t.inverted or t.layer is not None:
        raise NFHipError("rrule_with_logabsdet_jacobian: whole forward transform only")
    xm, vec = as_batch(x.to(flow.theta.dtype))
    d, n = xm.shape
    if d != flow.dist.d:
        raise NFHipError(f"dimension mismatch: flow has d={flow.dist.d}, input has {d}")
    dt, dev = xm.dtype, xm.device
    ctx = flow.ctx
    nbytes = int(ctx.lib.nf_tape_bytes(ctx.ptr, C.byref(flow.desc), n))
    if nbytes < 0:
        check(nbytes)
    tape = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)  # torch allocations are 512-byte aligned
    y = new_batch(d, n, dt, dev)
    ladj = torch.empty(n, dtype=dt, device=dev)
    theta = flow.theta
    check(ctx.lib.nf_flow_fwd_keep(ctx.ptr, C.byref(flow.desc), _ptr(theta), _ptr(xm), n, _ptr(y), _ptr(ladj), _ptr(tape),
                                   nbytes))

    def pullback(ybar: torch.Tensor, lbar: torch.Tensor, want_xbar: bool = True):
        """(xbar, gtheta); want_xbar=False returns (None, gtheta) and skips the cotangent's layout conversion where the
        library allows it (the reverse-KL loss does not differentiate through the base draws, src/objectives/elbo.jl:94)."""
        yb, _ = as_batch(ybar.to(dt))
        lb = lbar.to(dt).reshape(-1).contiguous()
        g = torch.empty(flow.P, dtype=dt, device=dev)
        c = flow.ctx
        if not want_xbar:
            code = c.lib.nf_flow_bwd_kept(c.ptr, C.byref(flow.desc), _ptr(theta), _ptr(tape), nbytes, _ptr(yb), _ptr(lb), n,
                                          _ptr(None), _ptr(g))
            if code == 0:
                return None, g
            if code != -1:  # NF_ERR_ARG: this flow needs the buffer (composition, or not on the tiled kernels)
                check(code)
        xbar = new_batch(d, n, dt, dev)
        check(c.lib.nf_flow_bwd_kept(c.ptr, C.byref(flow.desc), _ptr(theta), _ptr(tape), nbytes, _ptr(yb), _ptr(lb), n,
                                     _ptr(xbar), _ptr(g)))
        return (xbar[:, 0] if vec else xbar) if want_xbar else None, g

    if vec:
        return (y[:, 0], ladj[0]), pullback
    return (y, ladj), pullback


def transform(t: Transform, x: torch.Tensor):
    """Bijectors.transform(t, x)"""
    return with_logabsdet_jacobian(t, x)[0]


def base_logpdf(dist: MvNormal, xs: torch.Tensor):
    """logpdf(q0, xs) per column (MvNormal(zeros, I) or a general MvNormal(mu, Sigma))"""
    xm, vec = as_batch(xs)
    d, n = xm.shape
    if d != dist.d:
        raise NFHipError(f"dimension mismatch: distribution has d={dist.d}, input has {d}")
    if not dist.standard and dist.mu.dtype != xm.dtype:
        raise NFHipError(f"base distribution is {dist.mu.dtype}, input is {xm.dtype}")
    out = torch.empty(n, dtype=xm.dtype, device=xm.device)
    ctx = context_for(xm.device)
    check(ctx.lib.nf_base_logpdf_general(ctx.ptr, _dtype_code(xm.dtype), dist.base_ptr(), d, n, _ptr(xm), _ptr(out)))
    return out[0] if vec else out


def logpdf(flow, ys: torch.Tensor):
    """logpdf(td, y) = logpdf(td.dist, x) + ladj_inv  (Bijectors; used by
    src/objectives/loglikelihood.jl:23,31 and test/flow.jl:15)."""
    if isinstance(flow, MvNormal):
        return base_logpdf(flow, ys)
    ym, vec = as_batch(ys.to(flow.theta.dtype))
    d, n = ym.shape
    if d != flow.dist.d:
        raise NFHipError(f"dimension mismatch: flow has d={flow.dist.d}, input has {d}")
    out = torch.empty(n, dtype=ym.dtype, device=ym.device)
    val = C.c_double(0.0)
    ctx = flow.ctx
    check(ctx.lib.nf_loglikelihood(ctx.ptr, C.byref(flow.desc), _ptr(flow.theta), _ptr(ym), n, _ptr(out), C.byref(val)))
    return out[0] if vec else out


def rand(flow, n: Optional[int] = None, rng: Optional[PhiloxRNG] = None):
    """rand(rng, flow, n): base draws pushed through the transform (batched)."""
    rng = rng if rng is not None else _default_rng
    if isinstance(flow, MvNormal):
        return device_specific_rand(rng, flow, n)
    return device_specific_rand(rng, flow, n)


_default_rng = PhiloxRNG(0)


# --------------------------------------------------------------------------------------
# constructors (parameter initialisation follows the reference's init distributions:
# Flux.Dense Glorot-uniform weights / zero bias; PlanarLayer / RadialLayer randn)
# --------------------------------------------------------------------------------------
def _mlp_shapes(nin, hdims, nout):
    dims = [nin] + list(hdims) + [nout]
    return list(zip(dims[:-1], dims[1:]))


def _init_couplings(flow: Flow, gen: torch.Generator, outs_per_c):
    d = flow.dist.d
    parts = []
    for _ in range(flow.nlayers):
        for start in (0, 1):  # mask 1:2:d then 2:2:d (src/flows/realnvp.jl:138-139)
            c = len(range(start, d, 2))
            m = d - c
            for nout in outs_per_c(c):
                for (a, b) in _mlp_shapes(m, flow.hdims, nout):
                    lim = math.sqrt(6.0 / (a + b))
                    w = (torch.rand(a * b, generator=gen, dtype=torch.float64) * 2 - 1) * lim
                    parts += [w, torch.zeros(b, dtype=torch.float64)]
    return torch.cat(parts)


def _finish(flow: Flow, theta64: torch.Tensor) -> Flow:
    assert theta64.numel() == flow.P, (theta64.numel(), flow.P)
    flow.theta = theta64.to(flow.theta.dtype).to(flow.theta.device)
    return flow


def realnvp(q0: MvNormal, hdims: Sequence[int] = (32, 32), nlayers: int = 10, *, paramtype=torch.float64,
            device="cuda", seed: int = 0) -> Flow:
    """realnvp(q0, hdims, nlayers; paramtype)  (src/flows/realnvp.jl:170-192)."""
    flow = Flow("realnvp", q0, nlayers, hdims, dtype=paramtype, device=device)
    gen = torch.Generator().manual_seed(seed)
    return _finish(flow, _init_couplings(flow, gen, lambda c: (c, c)))


def nsf(q0: MvNormal, hdims: Sequence[int] = (32, 32), K: int = 10, B: float = 30.0, nlayers: int = 10, *,
        paramtype=torch.float64, device="cuda", seed: int = 0) -> Flow:
    """nsf(q0, hdims, K, B, nlayers; paramtype)  (src/flows/neuralspline.jl:218-234)."""
    flow = Flow("nsf", q0, nlayers, hdims, K, B, dtype=paramtype, device=device)
    gen = torch.Generator().manual_seed(seed)
    return _finish(flow, _init_couplings(flow, gen, lambda c: ((3 * K - 1) * c,)))


def planarflow(q0: MvNormal, nlayers: int, *, paramtype=torch.float64, device="cuda", seed: int = 0) -> Flow:
    """planarflow(q0, nlayers; paramtype)  (src/flows/planar_radial.jl:21-29)."""
    flow = Flow("planar", q0, nlayers, dtype=paramtype, device=device)
    gen = torch.Generator().manual_seed(seed)
    return _finish(flow, torch.randn(flow.P, generator=gen, dtype=torch.float64))


def radialflow(q0: MvNormal, nlayers: int, *, paramtype=torch.float64, device="cuda", seed: int = 0) -> Flow:
    """radialflow(q0, nlayers; paramtype)  (src/flows/planar_radial.jl:52-60)."""
    flow = Flow("radial", q0, nlayers, dtype=paramtype, device=device)
    gen = torch.Generator().manual_seed(seed)
    return _finish(flow, torch.randn(flow.P, generator=gen, dtype=torch.float64))


def meanfield(q0: MvNormal, *, paramtype=torch.float64, device="cuda") -> Flow:
    """transformed(q0, Shift(zeros) o Scale(ones))  (test/interface.jl:22-25)."""
    flow = Flow("meanfield", q0, 1, dtype=paramtype, device=device)
    d = q0.d
    return _finish(flow, torch.cat([torch.zeros(d, dtype=torch.float64), torch.ones(d, dtype=torch.float64)]))


def hamiltonianflow(dims: int, nblocks: int, nleapfrog: int, target, *, logeps0: float = math.log(0.05),
                    paramtype=torch.float64, device="cuda") -> Flow:
    """The Hamiltonian flow of example/demo_hamiltonian_flow.jl:132-146 on the joint z = [x; rho] (2*dims):
    a mean-field Gaussian reference, then `nblocks` blocks (momentum Shift o Scale) o LeapFrog(dims, logeps0,
    nleapfrog, score(target)).  `target` is a built-in target of dimension `dims` (diagonal Gaussian, Banana,
    Funnel -- the ones with a closed-form Hessian-vector product); pass the same object as `logp` to
    elbo / train_flow: the library forms logp_joint(z) = logp(x) + log N(rho; 0, I) (demo :121-128)."""
    check_target(target, paramtype, device, dims)
    flow = Flow("hamiltonian", MvNormal(2 * dims), nblocks, K=nleapfrog, dtype=paramtype, device=device, score=target)
    th = [torch.zeros(2 * dims, dtype=torch.float64), torch.ones(2 * dims, dtype=torch.float64)]
    for _ in range(nblocks):
        th += [torch.zeros(dims, dtype=torch.float64), torch.ones(dims, dtype=torch.float64),
               torch.full((dims,), float(logeps0), dtype=torch.float64)]
    return _finish(flow, torch.cat(th))


# --------------------------------------------------------------------------------------
# built-in targets (the `logp` closures of the reference's tests / demos)
# --------------------------------------------------------------------------------------
class DiagGaussTarget:
    """logp(z) = logpdf(MvNormal(mu, Diagonal(var)), z)  (test/flow.jl:43-46)."""

    def __init__(self, mu: torch.Tensor, var: torch.Tensor):
        if mu.dtype != var.dtype or mu.device != var.device or mu.shape != var.shape or mu.dim() != 1:
            raise NFHipError("DiagGaussTarget: mu and var must be vectors of one length, element type and device")
        self.mu, self.var = mu.contiguous(), var.contiguous()
        self.c = Target(_lib.NF_TARGET_DIAGGAUSS, self.mu.data_ptr(), self.var.data_ptr(), 0.0, 0.0)

    def check_compatible(self, dtype, device, d=None):
        """The C ABI passes mu / var as untyped device pointers that the kernels read in the FLOW's element type:
        a Float32 target under a Float64 flow would be read out of bounds.  Refuse instead."""
        if self.mu.dtype != dtype:
            raise NFHipError(f"target parameters are {self.mu.dtype} but the flow computes in {dtype}: "
                             "build the target in the flow's element type (the reference's logp closure would promote; "
                             "the device kernels cannot)")
        dv = torch.device(device)
        if self.mu.device.type != dv.type or (dv.index is not None and self.mu.device.index is not None and dv.index != self.mu.device.index):
            raise NFHipError(f"target parameters live on {self.mu.device}, the flow on {device}")
        if d is not None and self.mu.numel() != d:
            raise NFHipError(f"target has dimension {self.mu.numel()}, expected {d}")

    def __call__(self, ys):
        return target_logp(self, ys)


class BananaTarget:
    """Banana(d, b, var)  (example/targets/banana.jl; demo_planar_flow.jl:16)."""

    def __init__(self, d: int, b: float, var: float):
        self.d, self.b, self.variance = d, float(b), float(var)
        self.c = Target(_lib.NF_TARGET_BANANA, 0, 0, self.b, self.variance)

    def __call__(self, ys):
        return target_logp(self, ys)


class FunnelTarget:
    """Funnel(d, mu, sigma)  (example/targets/neal_funnel.jl:26-44; default Funnel(d) = Funnel(d, 0, 9))."""

    def __init__(self, d: int, mu: float = 0.0, sigma: float = 9.0):
        if d < 2:
            raise ValueError("dim must be >= 2")  # neal_funnel.jl:32
        if not sigma > 0:
            raise ValueError("σ must be > 0")  # neal_funnel.jl:33
        self.d, self.mu, self.sigma = d, float(mu), float(sigma)
        self.c = Target(_lib.NF_TARGET_FUNNEL, 0, 0, self.mu, self.sigma)

    def __call__(self, ys):
        return target_logp(self, ys)


class WarpedGaussTarget:
    """WarpedGauss(σ1, σ2), 2-dimensional  (example/targets/warped_gaussian.jl:25-37; default (1.0, 0.12))."""

    def __init__(self, sigma1: float = 1.0, sigma2: float = 0.12):
        if not (sigma1 > 0 and sigma2 > 0):
            raise ValueError("σ₁, σ₂ must be > 0")  # warped_gaussian.jl:31-32
        self.d, self.sigma1, self.sigma2 = 2, float(sigma1), float(sigma2)
        self.c = Target(_lib.NF_TARGET_WARPED, 0, 0, self.sigma1, self.sigma2)

    def __call__(self, ys):
        return target_logp(self, ys)


class CrossTarget:
    """Cross(μ, σ), 2-dimensional 4-component mixture  (example/targets/cross.jl:29-38; default (2.0, 0.15))."""

    def __init__(self, mu: float = 2.0, sigma: float = 0.15):
        if not sigma > 0:
            raise ValueError("σ must be > 0")
        self.d, self.mu, self.sigma = 2, float(mu), float(sigma)
        self.c = Target(_lib.NF_TARGET_CROSS, 0, 0, self.mu, self.sigma)

    def __call__(self, ys):
        return target_logp(self, ys)


def check_target(target, dtype, device=None, d=None):
    """Element-type / device / dimension agreement between a built-in target and the flow that will read it."""
    if isinstance(target, DiagGaussTarget):
        target.check_compatible(dtype, device if device is not None else target.mu.device, d)


def target_logp(target, ys: torch.Tensor, with_grad: bool = False):
    ym, vec = as_batch(ys)
    d, n = ym.shape
    check_target(target, ym.dtype, ym.device, d)
    out = torch.empty(n, dtype=ym.dtype, device=ym.device)
    grad = new_batch(d, n, ym.dtype, ym.device) if with_grad else None
    ctx = context_for(ym.device)
    check(ctx.lib.nf_target_logp(ctx.ptr, _dtype_code(ym.dtype), C.byref(target.c), d, n, _ptr(ym), _ptr(out), _ptr(grad)))
    res = out[0] if vec else out
    return (res, grad) if with_grad else res
