"""Parity probe (round 6): WHERE the log-determinant error of the weight-streaming RealNVP kernels comes from.  The suite's table has
`wide d=256 h=(256,256) nl=1: ladj` at 4.7 x the stated tolerance against 0.9 x for numpy's float32 on the same inputs, while `ys` of the
same run sits at the floor -- per-element accuracy is fine, the SUM of the 128 tanh values per coupling is not.  Prints, per coupling and
for the chain: mean SIGNED error, rms and max of (device - float64 oracle) and of (numpy float32 - float64 oracle), in units of the
element-wise tolerance; a mean of the order of the rms is a bias, not noise."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402

nf = load_package()
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nf_oracle as o  # noqa: E402


def cm(a):
    return torch.tensor(np.ascontiguousarray(a.T), dtype=torch.float32, device="cuda").t()


def stats(tag, got, ref):
    e = np.asarray(got, dtype=np.float64) - ref
    tol = 1e-6 + 1e-5 * np.abs(ref)
    r = e / tol
    print(f"   {tag:34s} mean {r.mean():+8.3f}  rms {np.sqrt((r * r).mean()):7.3f}  max {np.abs(r).max():7.3f}   (mean |ref| {np.abs(ref).mean():.3g})")


for d, hd, nl, n in ((256, (256, 256), 1, 200), (256, (256, 256), 1, 300), (120, (128, 100), 2, 150), (120, (128, 100), 2, 257), (64, (64, 64), 4, 512)):
    spec = o.FlowSpec("realnvp", d, nl, hd)
    rng = np.random.default_rng(d + n)
    th = (o.init_params(spec, rng) + 0.02 * rng.standard_normal(o.param_count(spec))).astype(np.float32)
    flow = nf.Flow("realnvp", nf.MvNormal(d), nl, hd, dtype=torch.float32, device="cuda", theta=torch.tensor(th, device="cuda"))
    xs = rng.standard_normal((d, n)).astype(np.float32)
    th64, xs64 = th.astype(np.float64), xs.astype(np.float64)
    print(f"realnvp d={d} h={hd} nl={nl} n={n}")
    # coupling by coupling: the oracle's own state is the input of every coupling, so errors do not chain
    z64 = xs64
    layers = o.layers_flat_order(spec)
    for k in reversed(range(2 * nl)):
        li = layers[k]
        y64, l64 = o._layer_fwd(spec, th64, li, z64)
        y32, l32 = o._layer_fwd(spec, th64.astype(np.float32), li, z64.astype(np.float32))
        yd, ld = nf.with_logabsdet_jacobian(nf.layer(flow, k), cm(z64.astype(np.float32)))
        stats(f"coupling {k}: ladj  device", ld.cpu().numpy(), l64)
        stats(f"coupling {k}: ladj  numpy f32", l32, l64)
        stats(f"coupling {k}: ys    device", yd.cpu().numpy(), y64)
        stats(f"coupling {k}: ys    numpy f32", y32, y64)
        if k > 0:  # how the NEXT coupling's log-determinant (float64 arithmetic) moves when its input carries this coupling's output error
            ln64 = o._layer_fwd(spec, th64, layers[k - 1], y64)[1]
            for nm, yy in (("device", yd.cpu().numpy().astype(np.float64)), ("numpy f32", y32.astype(np.float64))):
                stats(f"  -> next ladj (f64) at {nm} y", o._layer_fwd(spec, th64, layers[k - 1], yy)[1], ln64)
            e_d, e_n = yd.cpu().numpy().astype(np.float64) - y64, y32.astype(np.float64) - y64
            print(f"      y error: device mean {e_d.mean():+.3e} rms {np.sqrt((e_d**2).mean()):.3e} | numpy f32 mean {e_n.mean():+.3e} rms {np.sqrt((e_n**2).mean()):.3e}"
                  f" | corr(device, numpy) {np.corrcoef(e_d.ravel(), e_n.ravel())[0, 1]:+.3f}")
        z64 = y64
    if nl == 1:  # the chain's second coupling on the DEVICE's own state: arithmetic error at that input, absolute
        yd1, ld1 = nf.with_logabsdet_jacobian(nf.layer(flow, 1), cm(xs))
        yd0, ld0 = nf.with_logabsdet_jacobian(nf.layer(flow, 0), yd1)
        l0_at_dev = o._layer_fwd(spec, th64, layers[0], yd1.cpu().numpy().astype(np.float64))[1]
        y1_64, l1_64 = o._layer_fwd(spec, th64, layers[1], xs64)
        l0_64 = o._layer_fwd(spec, th64, layers[0], y1_64)[1]
        a = ld0.cpu().numpy().astype(np.float64) - l0_at_dev
        b = ld1.cpu().numpy().astype(np.float64) - l1_64
        c = l0_at_dev - l0_64
        tot = (ld1 + ld0).cpu().numpy().astype(np.float64) - (l1_64 + l0_64)
        for nm, e in (("coupling 1 arithmetic", b), ("coupling 0 arithmetic at device y1", a), ("coupling 0 propagation of y1's error", c), ("sum in fp32 - exact sum", tot)):
            print(f"      abs error, {nm:38s} mean {e.mean():+.3e} rms {np.sqrt((e * e).mean()):.3e} max {np.abs(e).max():.3e}")
        tol = 1e-6 + 1e-5 * np.abs(l1_64 + l0_64)
        w = np.argsort(-np.abs(tot) / tol)[:5]
        for j in w:
            print(f"      sample {j}: l1 {l1_64[j]:+.5f} l0 {l0_64[j]:+.5f} sum {l1_64[j] + l0_64[j]:+.6f} tol {tol[j]:.2e} | errors c1 {b[j]:+.2e} c0 {a[j]:+.2e} prop {c[j]:+.2e} total {tot[j]:+.2e} = {abs(tot[j]) / tol[j]:.2f} tol")
    ys_ref, l_ref = o.flow_fwd(spec, th64, xs64)
    y32, l32 = o.flow_fwd(spec, th64.astype(np.float32), xs64.astype(np.float32))
    ys, ladj = nf.with_logabsdet_jacobian(flow.transform, cm(xs))
    stats("chain: ladj  device", ladj.cpu().numpy(), l_ref)
    stats("chain: ladj  numpy f32", l32, l_ref)
