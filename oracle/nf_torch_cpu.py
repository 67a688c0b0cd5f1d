"""torch-CPU autograd restatement of the reference's reverse-KL training step for RealNVP.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (same rule as nf_oracle.py): imported by tests/ (to pin it against
nf_oracle.py) and by bench.py's `cpu_baseline` leg.  Nothing in the product path imports it.

Why it exists: the reference is Julia (Zygote/Mooncake reverse-mode AD over Flux `Dense` layers with OpenBLAS GEMMs)
and cannot run on the GPU box.  The closest CPU analogue that CAN run there is the same graph under torch-CPU
autograd with MKL/oneDNN GEMMs on all host cores (form 1).  The graph is the reference's, op
by op:

    x1, x2 = partition(mask, x)                 src/flows/realnvp.jl:77-79   (row gather)
    s, t   = s_net(x2), t_net(x2)               :80  ; fnn = Dense/leakyrelu chain, tanh on s (src/flows/utils.jl:71-100,
                                                       realnvp.jl:50-52)
    y1     = x1 .* exp.(s) .+ t                 :81
    ladj   = sum(s; dims=1)                     :82
    y      = combine(mask, y1, x2)              :83  (row scatter)
    elbo   = mean(logp(ys) - logpdf(q0, xs) + ladj)    src/objectives/elbo.jl:65-70,96
    g      = d(-elbo)/d theta ; Adam ; norm(g)         src/optimize.jl:86,89,99

A batch is kept as (N, d) row-major, which is the memory image of Julia's d x N column-major matrix, so the GEMMs
have the reference's shapes (W[out x in] times in x N).  theta is the same flat vector (Optimisers.destructure order,
nf_oracle.layers_flat_order), so a gradient from here is directly comparable with the oracle's and the device's.
"""
from __future__ import annotations

import math
import os
import time

import numpy as np
import torch

import nf_oracle as orc

LOG2PI = math.log(2.0 * math.pi)


def _mlp(theta: torch.Tensor, net, x: torch.Tensor, out_tanh: bool) -> torch.Tensor:
    a = x
    for i, (w_off, b_off, nout, nin) in enumerate(net):
        # Flux Dense weight is out x in, column-major in theta: element (o, i) at w_off + i*nout + o
        w = theta[w_off : w_off + nout * nin].view(nin, nout).t()
        b = theta[b_off : b_off + nout]
        a = torch.nn.functional.linear(a, w, b)
        if i < len(net) - 1:
            a = torch.nn.functional.leaky_relu(a, 0.01)
    return torch.tanh(a) if out_tanh else a


def realnvp_forward(spec: orc.FlowSpec, theta: torch.Tensor, xs: torch.Tensor):
    """xs: (N, d).  Returns (ys (N, d), ladj (N,)).  Layers execute last-listed first (src/flows/utils.jl:23-26)."""
    layers = orc.layers_flat_order(spec)
    x = xs
    ladj = torch.zeros(xs.shape[0], dtype=xs.dtype)
    for li in reversed(layers):
        it = torch.as_tensor(li.idx_t)
        ic = torch.as_tensor(li.idx_c)
        x1, x2 = x[:, it], x[:, ic]
        s = _mlp(theta, li.nets[0], x2, True)
        t = _mlp(theta, li.nets[1], x2, False)
        y1 = x1 * torch.exp(s) + t
        ladj = ladj + s.sum(dim=1)
        y = torch.empty_like(x)
        y[:, it] = y1
        y[:, ic] = x2
        x = y
    return x, ladj


def neg_elbo(spec: orc.FlowSpec, theta: torch.Tensor, mu: torch.Tensor, var: torch.Tensor, xs: torch.Tensor) -> torch.Tensor:
    ys, ladj = realnvp_forward(spec, theta, xs)
    d = xs.shape[1]
    logp = -0.5 * (LOG2PI + torch.log(var) + (ys - mu) ** 2 / var).sum(dim=1)
    logq = -0.5 * d * LOG2PI - 0.5 * (xs * xs).sum(dim=1)
    return -(logp - logq + ladj).mean()


def value_and_grad(spec, theta_np, mu_np, var_np, xs_np, dtype=torch.float64):
    """(loss, grad) as numpy, for the parity pin against nf_oracle.neg_elbo_value_and_grad.  xs_np is (d, N)."""
    theta = torch.tensor(theta_np, dtype=dtype, requires_grad=True)
    loss = neg_elbo(spec, theta, torch.tensor(mu_np, dtype=dtype), torch.tensor(var_np, dtype=dtype),
                    torch.tensor(np.ascontiguousarray(xs_np.T), dtype=dtype))
    (g,) = torch.autograd.grad(loss, theta)
    return float(loss), g.numpy()


def usable_cpus() -> int:
    """Host cpus this process may actually use: the affinity mask, capped by the cgroup cpu quota when there is one
    (a container that sees 256 cpus but is throttled to a few must not be oversubscribed with 256 threads)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, int(quota / int(g.read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def time_training_steps(d: int, hdims, nlayers: int, batch: int, seconds_budget: float = 20.0, max_steps: int = 50):
    """Times full training steps (draw, forward, backward, Adam, gradient norm) in float32 on all usable host cpus.
    Returns a dict for bench.py's `cpu_baseline` object.  Bounded: one warm-up step, then steps until the budget is
    spent (at least one)."""
    threads = int(os.environ.get("NF_CPU_THREADS", "0")) or usable_cpus()
    torch.set_num_threads(threads)
    spec = orc.FlowSpec("realnvp", d, nlayers, tuple(hdims))
    rng = np.random.default_rng(123)
    theta = torch.tensor(orc.init_params(spec, rng, dtype=np.float32), requires_grad=True)
    mu = torch.tensor(rng.standard_normal(d).astype(np.float32))
    var = torch.tensor((rng.uniform(size=d) + 1e-3).astype(np.float32))
    opt = torch.optim.Adam([theta], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    gen = torch.Generator().manual_seed(123)

    def step():
        xs = torch.randn(batch, d, generator=gen)  # the reference's randn-based draw (src/NormalizingFlows.jl:109-115)
        opt.zero_grad(set_to_none=True)
        loss = neg_elbo(spec, theta, mu, var, xs)
        loss.backward()
        gn = theta.grad.norm()
        opt.step()
        return float(loss.detach()), float(gn)

    step()  # warm-up: thread pools, oneDNN primitive caches
    times = []
    t_all = time.perf_counter()
    while len(times) < max_steps and (time.perf_counter() - t_all < seconds_budget or len(times) < 1):
        t0 = time.perf_counter()
        loss, gn = step()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    assert np.isfinite(loss) and np.isfinite(gn)
    return {
        "value": batch / med,
        "unit": "samples/s",
        "cores": threads,
        "kind": "port",
        "sample": (f"{len(times)} full training steps at batch {batch} (the GPU workload's batch, flow, target and dtype; median "
                   f"step {1e3 * med:.1f} ms), torch-CPU autograd + MKL/oneDNN GEMM, {threads} threads of {os.cpu_count()} host "
                   "cpus: CPU restatement of the reference algorithm (the Julia reference cannot run on this box)"),
        "ms_per_step": 1e3 * med,
    }
