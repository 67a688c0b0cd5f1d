"""Pins the CPU oracle to outputs of the REAL reference when they are present.

`tools/make_reference_goldens.jl` (run by a maintainer who has Julia + the reference's test environment) writes
tests/golden/ref_<name>.npz next to every committed fixture: what NormalizingFlows.jl / Bijectors / MonotonicSplines /
Flux / Optimisers / Zygote compute from the fixture's own theta, xs and target.  This module compares
oracle/nf_oracle.py with those files at the parity tolerances (tests/parity.py) -- the step that turns "parity
unpinned"  into "pinned".  No Julia in the build container, so without ref files the value
tests are SKIPPED (reported as such, never silently green); the consistency test of the Julia script's case table
always runs.
"""
import glob
import os
import re

import numpy as np
import pytest

import nf_oracle as o
import parity as P
from __graft_entry__ import ROOT

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
REFS = sorted(glob.glob(os.path.join(GOLDEN_DIR, "ref_*.npz")))


def _make_golden_cases():
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN_DIR, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.CASES


def test_julia_case_table_matches_the_fixture_generator():
    """tools/make_reference_goldens.jl restates the fixture specs (NPZ.jl cannot read numpy strings): same names, flow
    kinds, sizes, element types and targets as tests/golden/make_golden.py:CASES and as the committed fixtures."""
    src = open(os.path.join(ROOT, "tools", "make_reference_goldens.jl")).read()
    rows = re.findall(r'"(\w+)"\s*=>\s*\(:(\w+),\s*(\d+),\s*(\d+),\s*(?:Int)?\[([\d,\s]*)\],\s*(\d+),\s*([\d.]+),\s*(Float\d+),\s*:(\w+)\)', src)
    table = {r[0]: r[1:] for r in rows}
    cases = _make_golden_cases()
    assert set(table) == set(cases), set(table) ^ set(cases)
    for name, (spec, n, tkind, dt) in cases.items():
        kind, d, nl, hd, K, B, T, tk = table[name]
        hd = tuple(int(h) for h in hd.replace(" ", "").split(",") if h)
        assert (kind, int(d), int(nl), hd, tk) == (spec.kind, spec.d, spec.nlayers, tuple(spec.hdims), tkind), name
        if spec.kind == "nsf":
            assert int(K) == spec.K and float(B) == spec.B, name
        assert T == ("Float32" if dt == np.float32 else "Float64"), name
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        assert str(z["kind"]) == kind and int(z["d"]) == int(d) and str(z["dtype"]) == T.lower(), name


def _oracle_inputs(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    kind, d, nl = str(z["kind"]), int(z["d"]), int(z["nlayers"])
    spec = o.FlowSpec(kind, d, nl, tuple(int(h) for h in z["hdims"]), int(z["K"]), float(z["B"]))
    tk = str(z["target"])
    tp = z["target_params"]
    tgt = ("diaggauss", tp[0], tp[1]) if tk == "diaggauss" else (tk, float(tp[0, 0]), float(tp[1, 0]))
    return z, spec, tgt


@pytest.mark.skipif(not REFS, reason="no tests/golden/ref_*.npz: run tools/make_reference_goldens.jl with Julia + the "
                                     "reference's test environment to pin the oracle (none can be produced in the build container)")
@pytest.mark.parametrize("path", REFS, ids=[os.path.basename(p)[4:-4] for p in REFS])
def test_oracle_equals_the_reference(path):
    name = os.path.basename(path)[4:-4]
    ref = np.load(path)
    z, spec, tgt = _oracle_inputs(name)
    f64 = str(z["dtype"]) == "float64"
    th, xs = z["theta"].astype(np.float64), z["xs"].astype(np.float64)
    # the reference ran in the fixture's element type; the oracle in float64 on the same representable inputs
    rt, at = (1e-10, 1e-12) if f64 else (P.Y_RTOL, P.Y_ATOL)
    ys, ladj = o.flow_fwd(spec, th, xs)
    fl = None if f64 else o.flow_fwd(spec, *P.f32(th, xs))
    P.elementwise(f"ref {name}: ys", ref["ys"], ys, rt, at, None if f64 else fl[0])
    P.elementwise(f"ref {name}: ladj", ref["ladj"], ladj, rt, at, None if f64 else fl[1])
    elbos = o.batched_elbos(spec, th, tgt, xs)
    P.elementwise(f"ref {name}: elbos", ref["elbos"], elbos, rt, at, None if f64 else o.batched_elbos(spec, *P.f32(th, tgt, xs)))
    loss, grad = o.neg_elbo_value_and_grad(spec, th, tgt, xs)
    P.scalar(f"ref {name}: loss", float(ref["loss"]), loss, 1e-10 if f64 else P.LOSS_RTOL)
    g32 = None if f64 else o.neg_elbo_value_and_grad(spec, *P.f32(th, tgt, xs))[1]
    P.gradient(f"ref {name}: grad (Zygote)", ref["grad"], grad, P.F64_GRAD if f64 else P.GRAD_RTOL, g32)
    th1 = th.copy()
    o.adam_update(th1, grad, np.zeros_like(th), np.zeros_like(th), 1)
    assert np.abs(ref["theta_adam1"] - th1).max() <= (1e-12 if f64 else 2e-7) + 1e-6 * 1e-3, "Optimisers.Adam step"
    fkl_loss, fkl_grad = o.neg_loglik_value_and_grad(spec, th, z["fkl_xs"].astype(np.float64))
    P.scalar(f"ref {name}: forward-KL loss", float(ref["fkl_loss"]), fkl_loss, 1e-9 if f64 else 10 * P.LOSS_RTOL)
    P.gradient(f"ref {name}: forward-KL grad", ref["fkl_grad"], fkl_grad, 1e-7 if f64 else 10 * P.GRAD_RTOL)
    if spec.kind == "nsf" and "rqs_raw" in ref.files:
        # MonotonicSplines known answer: the least certain piece of the restatement (row order widths / heights /
        # derivatives, reshape(:, c, N), no minimum-bin floor)
        raw, x1 = ref["rqs_raw"], ref["rqs_x1"]
        c = x1.shape[0]
        pX, pY, dY = o.rqs_params_from_nn(raw, c, spec.B)
        for key, mine in (("rqs_pX", pX), ("rqs_pY", pY), ("rqs_dYdX", dY)):
            assert ref[key].shape == mine.shape, (key, ref[key].shape, mine.shape)
            P.elementwise(f"ref {name}: {key}", ref[key], mine, 1e-10 if f64 else 1e-5, 1e-12 if f64 else 1e-6)
        y1, lj = o.rqs_forward(x1, pX, pY, dY)
        P.elementwise(f"ref {name}: rqs_forward y", ref["rqs_y1"], y1, 1e-10 if f64 else 1e-5, 1e-12 if f64 else 1e-6)
        P.elementwise(f"ref {name}: rqs_forward logjac", ref["rqs_logjac"], lj, 1e-9 if f64 else 1e-5, 1e-11 if f64 else 1e-5)


@pytest.mark.gpu
@pytest.mark.skipif(not REFS, reason="no tests/golden/ref_*.npz (see tools/make_reference_goldens.jl)")
@pytest.mark.parametrize("path", REFS, ids=[os.path.basename(p)[4:-4] for p in REFS])
def test_hip_library_equals_the_reference(path):
    """The HIP path through the C ABI against the reference's own numbers, fixture by fixture."""
    import torch

    from __graft_entry__ import load_package

    nf = load_package()
    name = os.path.basename(path)[4:-4]
    ref = np.load(path)
    z, spec, tgt = _oracle_inputs(name)
    dt = torch.float32 if str(z["dtype"]) == "float32" else torch.float64
    f64 = dt == torch.float64
    flow = nf.Flow(spec.kind, nf.MvNormal(spec.d), spec.nlayers, spec.hdims, spec.K, spec.B, dtype=dt, device="cuda",
                   theta=torch.tensor(z["theta"], dtype=dt, device="cuda"))
    xs = torch.tensor(np.ascontiguousarray(z["xs"].T), dtype=dt, device="cuda").t()
    ys, ladj = nf.with_logabsdet_jacobian(flow.transform, xs)
    th, x64 = z["theta"].astype(np.float64), z["xs"].astype(np.float64)
    fl = None if f64 else o.flow_fwd(spec, *P.f32(th, x64))
    rt, at = (1e-10, 1e-12) if f64 else (P.Y_RTOL, P.Y_ATOL)
    P.elementwise(f"ref/hip {name}: ys", ys, ref["ys"], rt, at, None if f64 else fl[0])
    P.elementwise(f"ref/hip {name}: ladj", ladj, ref["ladj"], rt, at, None if f64 else fl[1])
    if tgt[0] == "diaggauss":
        t = nf.DiagGaussTarget(torch.tensor(tgt[1], dtype=dt, device="cuda"), torch.tensor(tgt[2], dtype=dt, device="cuda"))
    else:
        t = {"banana": lambda: nf.BananaTarget(spec.d, tgt[1], tgt[2]), "funnel": lambda: nf.FunnelTarget(spec.d, tgt[1], tgt[2]),
             "warped": lambda: nf.WarpedGaussTarget(tgt[1], tgt[2]), "cross": lambda: nf.CrossTarget(tgt[1], tgt[2])}[tgt[0]]()
    loss, g = nf.value_and_gradient(nf.elbo_batch, flow, t, xs)
    P.scalar(f"ref/hip {name}: loss", loss, float(ref["loss"]), 1e-10 if f64 else P.LOSS_RTOL)
    g32 = None if f64 else o.neg_elbo_value_and_grad(spec, *P.f32(th, tgt, x64))[1]
    P.gradient(f"ref/hip {name}: grad", g, ref["grad"], P.F64_GRAD if f64 else P.GRAD_RTOL, g32)
