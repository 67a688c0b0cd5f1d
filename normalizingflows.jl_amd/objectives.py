"""Variational objectives and the training loop: host mirror of
src/objectives/elbo.jl, src/objectives/loglikelihood.jl, src/optimize.jl and
src/NormalizingFlows.jl:train_flow.  The arithmetic runs in libnfhip.so.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Callable, Optional

import torch

from ._lib import NFHipError, check
from .flows import (BananaTarget, CrossTarget, DiagGaussTarget, FunnelTarget, WarpedGaussTarget, Flow, PhiloxRNG, _dtype_code, _ptr, as_batch, base_logpdf,
                    check_target, device_specific_rand, new_batch, rrule_with_logabsdet_jacobian, with_logabsdet_jacobian)

_BUILTIN = (DiagGaussTarget, BananaTarget, FunnelTarget, WarpedGaussTarget, CrossTarget)


def _target_dim(flow: Flow) -> int:
    return flow.dist.d // 2 if flow.kind == "hamiltonian" else flow.dist.d  # Hamiltonian targets describe x of [x; rho]


def _builtin(flow: Flow, logp) -> bool:
    """True for a built-in device target -- after checking that its parameters match the flow's element type, device
    and dimension (the ABI carries them as untyped pointers)."""
    if not isinstance(logp, _BUILTIN):
        return False
    check_target(logp, flow.theta.dtype, flow.theta.device, _target_dim(flow))
    return True


def _host_double():
    return C.c_double(0.0)


# --------------------------------------------------------------------------------------
# reverse KL
# --------------------------------------------------------------------------------------
def batched_elbos(flow: Flow, logp, xs: torch.Tensor) -> torch.Tensor:
    """_batched_elbos(flow, logp, xs)  (src/objectives/elbo.jl:65-70)."""
    xm, _ = as_batch(xs.to(flow.theta.dtype))
    d, n = xm.shape
    if _builtin(flow, logp):
        out = torch.empty(n, dtype=xm.dtype, device=xm.device)
        val = _host_double()
        ctx = flow.ctx
        check(ctx.lib.nf_elbo_batch(ctx.ptr, C.byref(flow.desc), C.byref(logp.c), _ptr(flow.theta), _ptr(xm), n,
                                    _ptr(out), C.byref(val)))
        return out
    ys, ladj = with_logabsdet_jacobian(flow.transform, xm)
    return logp(ys) - base_logpdf(flow.dist, xm) + ladj


def elbo_batch(*args):
    """elbo_batch(flow, logp, xs) / elbo_batch([rng,] flow, logp, n_samples)
    (src/objectives/elbo.jl:89-99)."""
    rng, flow, logp, last = _split_args(args)
    if isinstance(last, int):
        if _builtin(flow, logp):
            val = _host_double()
            ctx = flow.ctx
            check(ctx.lib.nf_elbo_batch_rng(ctx.ptr, C.byref(flow.desc), C.byref(logp.c), _ptr(flow.theta), last,
                                            rng.seed, rng.sample_offset, rng.next_stream(), C.byref(val)))
            return val.value
        last = device_specific_rand(rng, flow.dist, last, device=flow.theta.device, dtype=flow.theta.dtype)
    xm, _ = as_batch(last.to(flow.theta.dtype))
    if _builtin(flow, logp):
        val = _host_double()
        ctx = flow.ctx
        check(ctx.lib.nf_elbo_batch(ctx.ptr, C.byref(flow.desc), C.byref(logp.c), _ptr(flow.theta), _ptr(xm),
                                    xm.shape[1], _ptr(None), C.byref(val)))
        return val.value
    return float(batched_elbos(flow, logp, xm).double().mean())


def elbo(*args):
    """elbo(flow, logp, xs) / elbo([rng,] flow, logp, n_samples): per-column map of
    elbo_single_sample (src/objectives/elbo.jl:4-7,26-46).  Each column goes through the same
    kernels as the batched path with N = 1, so the value equals elbo_batch up to summation order."""
    rng, flow, logp, last = _split_args(args)
    if isinstance(last, int):
        last = device_specific_rand(rng, flow.dist, last, device=flow.theta.device, dtype=flow.theta.dtype)
    xm, _ = as_batch(last.to(flow.theta.dtype))
    vals = [float(batched_elbos(flow, logp, xm[:, j : j + 1])[0]) for j in range(xm.shape[1])]
    return sum(vals) / len(vals)


def _split_args(args):
    if isinstance(args[0], PhiloxRNG):
        rng, flow, logp, last = args
    else:
        from .flows import _default_rng

        rng = _default_rng
        flow, logp, last = args
    return rng, flow, logp, last


# --------------------------------------------------------------------------------------
# forward KL
# --------------------------------------------------------------------------------------
def loglikelihood(rng, flow: Flow, xs: torch.Tensor) -> float:
    """loglikelihood(rng, flow, xs)  (src/objectives/loglikelihood.jl:26-33); `rng` is the
    unused placeholder argument the reference keeps for a uniform objective signature."""
    xm, _ = as_batch(xs.to(flow.theta.dtype))
    val = _host_double()
    ctx = flow.ctx
    check(ctx.lib.nf_loglikelihood(ctx.ptr, C.byref(flow.desc), _ptr(flow.theta), _ptr(xm), xm.shape[1], _ptr(None),
                                   C.byref(val)))
    return val.value


def loglikelihood_value_and_gradient(flow: Flow, xs: torch.Tensor, n_global: Optional[int] = None):
    """(loss, grad) of loss(theta) = -loglikelihood(rng, re(theta), xs) -- what
    `train_flow(loglikelihood, flow, xs)` differentiates (src/NormalizingFlows.jl:69 with
    src/objectives/loglikelihood.jl:26-33; gradient by src/optimize.jl:77,86).  For a shard of a global
    data set pass n_global and all-reduce the returned loss and grad over ranks."""
    dt, dev = flow.theta.dtype, flow.theta.device
    xm, _ = as_batch(xs.to(dt))
    n = xm.shape[1]
    ng = n if n_global is None else int(n_global)
    out = torch.empty(flow.P + 1, dtype=dt, device=dev)
    ctx = flow.ctx
    check(ctx.lib.nf_loglikelihood_value_and_grad(ctx.ptr, C.byref(flow.desc), _ptr(flow.theta), _ptr(xm), n, ng, _ptr(out)))
    return float(out[flow.P]), out[: flow.P]


# --------------------------------------------------------------------------------------
# gradients (the device analogue of _value_and_gradient, src/optimize.jl:12-14)
# --------------------------------------------------------------------------------------
def value_and_gradient(vo, flow: Flow, logp, xs_or_n, rng: Optional[PhiloxRNG] = None, n_global: Optional[int] = None):
    """(loss, grad) of loss(theta) = -vo(rng, re(theta), logp, ...) (src/NormalizingFlows.jl:69).

    Built-in targets run the whole step inside the library (nf_elbo_value_and_grad).  An
    arbitrary `logp` callable takes the split path: library forward that keeps its tape
    (nf_flow_fwd_keep), the callable's own torch-autograd gradient w.r.t. ys, library pullback
    from that tape (nf_flow_bwd_kept) -- the same reverse kernels as the built-in step.
    Returns (loss: float, grad: tensor[P]) -- for a shard of a global batch pass n_global and
    all-reduce the returned grad and loss over ranks.
    """
    if vo is loglikelihood:
        # forward KL: the second positional slot of the reference's loss closure is the data
        return loglikelihood_value_and_gradient(flow, logp if xs_or_n is None else xs_or_n, n_global)
    if vo not in (elbo, elbo_batch):
        raise NFHipError("value_and_gradient supports elbo, elbo_batch and loglikelihood")
    dt, dev = flow.theta.dtype, flow.theta.device
    ctx = flow.ctx
    P = flow.P
    if isinstance(xs_or_n, int):
        n = xs_or_n
        xm = None
    else:
        xm, _ = as_batch(xs_or_n.to(dt))
        n = xm.shape[1]
    ng = n if n_global is None else int(n_global)
    rng = rng if rng is not None else PhiloxRNG(0)
    if _builtin(flow, logp):
        out = torch.empty(P + 1, dtype=dt, device=dev)
        check(ctx.lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(logp.c), _ptr(flow.theta), _ptr(xm), n,
                                             ng, rng.seed, rng.sample_offset, rng.next_stream() if xm is None else 0,
                                             _ptr(out)))
        return float(out[P]), out[:P]
    # generic closure
    if xm is None:
        xm = device_specific_rand(rng, flow.dist, n, device=dev, dtype=dt)
    (ys, ladj), pullback = rrule_with_logabsdet_jacobian(flow.transform, xm)
    yreq = ys.detach().requires_grad_(True)
    lp = logp(yreq)
    (glp,) = torch.autograd.grad(lp.sum(), yreq)
    elbos = lp.detach() - base_logpdf(flow.dist, xm) + ladj
    lbar = torch.full((n,), -1.0 / ng, dtype=dt, device=dev)
    _, g = pullback(-glp / ng, lbar, want_xbar=False)
    return float(-elbos.double().sum() / ng), g


# --------------------------------------------------------------------------------------
# optimiser + training loop
# --------------------------------------------------------------------------------------
@dataclass
class Adam:
    """Optimisers.Adam(eta, beta, epsilon)"""

    eta: float = 1e-3
    beta: tuple = (0.9, 0.999)
    epsilon: float = 1e-8


@dataclass
class AdamState:
    """st returned by optimize: Adam moments and step count (src/optimize.jl:106-107)."""

    m: torch.Tensor
    v: torch.Tensor
    t: int = 0


@dataclass
class Descent:
    """Optimisers.Descent(eta)"""

    eta: float = 0.1


@dataclass
class Momentum:
    """Optimisers.Momentum(eta, rho)"""

    eta: float = 0.01
    rho: float = 0.9


@dataclass
class SGDState:
    """st of Descent (vel is None) / Momentum."""

    vel: Optional[torch.Tensor]
    t: int = 0


def setup(opt, theta: torch.Tensor):
    """Optimisers.setup(rule, theta)  (src/optimize.jl:80)."""
    if isinstance(opt, Adam):
        return AdamState(torch.zeros_like(theta), torch.zeros_like(theta), 0)
    if isinstance(opt, Momentum):
        return SGDState(torch.zeros_like(theta), 0)
    if isinstance(opt, Descent):
        return SGDState(None, 0)
    raise TypeError(f"unsupported optimiser rule {type(opt).__name__}")


def update(opt, st, theta: torch.Tensor, g: torch.Tensor, want_norm: bool = True):
    """Optimisers.update!(st, theta, g) for any supported rule; returns norm(g)."""
    if isinstance(opt, Adam):
        return adam_update(opt, st, theta, g, want_norm)
    from ._lib import context_for

    ctx = context_for(theta.device)
    st.t += 1
    gn = torch.empty(1, dtype=theta.dtype, device=theta.device) if want_norm else None
    rho = opt.rho if isinstance(opt, Momentum) else 0.0
    check(ctx.lib.nf_sgd_update(ctx.ptr, _dtype_code(theta.dtype), _ptr(theta), _ptr(g), _ptr(st.vel), theta.numel(),
                                opt.eta, rho, _ptr(gn)))
    return gn


def adam_update(opt: Adam, st: AdamState, theta: torch.Tensor, g: torch.Tensor, want_norm: bool = True):
    """Optimisers.update!(st, theta, g)  (src/optimize.jl:99); also returns norm(g) (:89)."""
    from ._lib import context_for

    ctx = context_for(theta.device)
    st.t += 1
    gn = torch.empty(1, dtype=theta.dtype, device=theta.device) if want_norm else None
    check(ctx.lib.nf_adam_update(ctx.ptr, _dtype_code(theta.dtype), _ptr(theta), _ptr(g), _ptr(st.m), _ptr(st.v),
                                 theta.numel(), opt.eta, opt.beta[0], opt.beta[1], opt.epsilon, st.t, _ptr(gn)))
    return gn


def optimize(loss_and_grad: Callable, theta0: torch.Tensor, reconstruct, *, max_iters: int = 10000,
             optimiser=None, show_progress: bool = False, callback=None,
             hasconverged=lambda i, stats, re, theta, st: False, all_reduce=None, state=None):
    """optimize(adbackend, loss, theta0, re, args...; kwargs...)  (src/optimize.jl:57-108).
    `loss_and_grad(theta) -> (loss, grad)` plays the role of DI.value_and_gradient(loss, ...).
    `all_reduce(buf)` (optional) sums [grad ; loss] over data-parallel ranks.  `state` (optional): the `st`
    returned by an earlier call, to continue training where it stopped (the reference returns `st` "for
    potential continuation of training", src/optimize.jl:106)."""
    optimiser = optimiser or Adam()
    theta = theta0.clone()
    st = state if state is not None else setup(optimiser, theta)
    opt_stats = []
    converged = False
    i = 1
    while i <= max_iters and not converged:
        ls, g = loss_and_grad(theta)
        if all_reduce is not None:
            buf = torch.cat([g, torch.tensor([ls], dtype=g.dtype, device=g.device)])
            all_reduce(buf)
            g, ls = buf[:-1], float(buf[-1])
        # the reference records the stat and runs the callback on the parameters BEFORE the update
        # (src/optimize.jl:88-99); the gradient norm comes out of the same kernel as the update
        theta_before = theta.clone() if callback is not None else None
        gn = update(optimiser, st, theta, g)
        stat = {"iteration": i, "loss": ls, "gradient_norm": float(gn)}
        if callback is not None:
            new_stat = callback(i, opt_stats, reconstruct, theta_before)
            if new_stat is not None:
                stat.update(new_stat)
        opt_stats.append(stat)
        i += 1
        converged = hasconverged(i, stat, reconstruct, theta, st)
        if show_progress and (i % 100 == 0):
            print(f"Training iter {i}: loss {ls:.6g} |g| {stat['gradient_norm']:.3g}")
    return theta, opt_stats, st


def _fused_steps_apply(vo, flow: Flow, rest, rng: PhiloxRNG, optimiser, kwargs) -> bool:
    """True when a training run is what nf_elbo_step computes in one call per iteration: reverse KL on in-library draws
    of the whole batch, a built-in device target, Adam, no data-parallel hook or communicator, and the draw counter in step
    with Adam's step count (a fresh run, or `state` and `rng` continued together) -- nf_elbo_step uses ONE index for both."""
    if vo not in (elbo, elbo_batch) or len(rest) != 2 or not isinstance(rest[1], int):
        return False
    if not (optimiser is None or isinstance(optimiser, Adam)) or kwargs.get("all_reduce") is not None:
        return False
    st = kwargs.get("state")
    if st is not None and not isinstance(st, AdamState):
        return False
    if rng.sample_offset != 0 or rng.stream != (st.t if st is not None else 0):
        return False
    # a context that holds a communicator makes nf_elbo_step data-parallel (draw offset rank * n, global batch n * world, the
    # all-reduce inside the library): not the run `optimize` over value_and_gradient would be
    if int(flow.ctx.lib.nf_comm_size(flow.ctx.ptr)) > 1:
        return False
    return _builtin(flow, rest[0])


def _optimize_fused(flow: Flow, theta0: torch.Tensor, reconstruct, rng: PhiloxRNG, logp, n: int, *, max_iters: int,
                    optimiser: Adam, show_progress: bool = False, callback=None, hasconverged=None, all_reduce=None,
                    state=None):
    """The loop of `optimize` (src/optimize.jl:85-104) with each iteration ONE library call: nf_elbo_step = draws,
    forward, reverse pass, Adam, norm(g) (three launches for the LDS-resident RealNVP shapes, what bench.py times).  The
    loop owns theta between steps, so it opts in to the library's packed-weight cache (nf_ctx_set_weight_cache) and out
    again on return; a user `hasconverged` sees the live theta and is followed by nf_ctx_weights_changed.  Same numbers
    as `optimize` over value_and_gradient + update (tests/test_gpu_tape.py)."""
    from ._lib import NF_ERR_NONFINITE

    theta = theta0.clone()
    st = state if state is not None else setup(optimiser, theta)
    ctx, lib = flow.ctx, flow.ctx.lib
    opt_stats = []
    converged = False
    i = 1
    loss, gn = _host_double(), _host_double()
    check(lib.nf_ctx_set_weight_cache(ctx.ptr, 1))
    try:
        while i <= max_iters and not converged:
            theta_before = theta.clone() if callback is not None else None
            code = lib.nf_elbo_step(ctx.ptr, C.byref(flow.desc), C.byref(logp.c), _ptr(theta), _ptr(st.m), _ptr(st.v), n,
                                    rng.seed, rng.next_stream(), optimiser.eta, optimiser.beta[0], optimiser.beta[1],
                                    optimiser.epsilon, C.byref(loss), C.byref(gn))
            if code != NF_ERR_NONFINITE:  # a non-finite loss is recorded, as the reference's loop would record it
                check(code)
            st.t += 1
            stat = {"iteration": i, "loss": loss.value, "gradient_norm": gn.value}
            if callback is not None:
                new_stat = callback(i, opt_stats, reconstruct, theta_before)
                if new_stat is not None:
                    stat.update(new_stat)
            opt_stats.append(stat)
            i += 1
            if hasconverged is not None:
                converged = hasconverged(i, stat, reconstruct, theta, st)
                check(lib.nf_ctx_weights_changed(ctx.ptr))  # it was handed the live theta
            if show_progress and (i % 100 == 0):
                print(f"Training iter {i}: loss {stat['loss']:.6g} |g| {stat['gradient_norm']:.3g}")
    finally:
        lib.nf_ctx_set_weight_cache(ctx.ptr, 0)
    return theta, opt_stats, st


def train_flow(*args, max_iters: int = 1000, optimiser: Adam = None, ADbackend=None, **kwargs):
    """train_flow([rng,] vo, flow, args...; max_iters, optimiser, ADbackend, kwargs...)
    (src/NormalizingFlows.jl:51-86) -> (flow_trained, opt_stats, st).

    `ADbackend` is accepted for signature compatibility; gradients come from the library's
    hand-derived reverse pass (the role a custom ADTypes backend plays in the reference).  Reverse-KL runs on a
    built-in target with Adam go through nf_elbo_step, one library call per iteration (`_optimize_fused`); everything
    else through `optimize` over value_and_gradient + update."""
    if isinstance(args[0], PhiloxRNG):
        rng, vo, flow, *rest = args
    else:
        rng = PhiloxRNG(0)
        vo, flow, *rest = args
    theta_flat, re = flow.destructure()
    if _fused_steps_apply(vo, flow, rest, rng, optimiser, kwargs):
        theta, stats, st = _optimize_fused(flow, theta_flat, re, rng, rest[0], rest[1], max_iters=max_iters,
                                           optimiser=optimiser or Adam(), **kwargs)
        return re(theta), stats, st

    def loss_and_grad(theta):
        f = re(theta)
        if vo in (elbo, elbo_batch):
            logp, n = rest
            return value_and_gradient(vo, f, logp, n, rng)
        if vo is loglikelihood:  # train_flow(loglikelihood, flow, xs): forward KL on the data xs
            (xs,) = rest
            return loglikelihood_value_and_gradient(f, xs)
        raise NFHipError("train_flow: objective must be elbo, elbo_batch or loglikelihood")

    theta, stats, st = optimize(loss_and_grad, theta_flat, re, max_iters=max_iters, optimiser=optimiser, **kwargs)
    return re(theta), stats, st
