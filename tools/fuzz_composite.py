"""Randomised parity sweep (GPU) of heterogeneous compositions -- create_flow((L1, ..., Ln), q0) with mixed bijector
families (src/flows/utils.jl:23-26) -- and general MvNormal(mu, Sigma) bases, against the oracle: forward, inverse,
ELBO value (xs and rng forms), training-step gradient, loglikelihood.  Usage: python tools/fuzz_composite.py [n_cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nf_oracle as o  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

nf = load_package()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


bad = 0
for case in range(ncases):
    rng = np.random.default_rng([seed, case, 77])
    f64 = bool(rng.integers(0, 3) == 0)
    dt, npdt = (torch.float64, np.float64) if f64 else (torch.float32, np.float32)
    d = int(rng.integers(2, 25))
    nseg = int(rng.integers(2, 5))
    specs = []
    for _ in range(nseg):
        kind = rng.choice(["planar", "radial", "realnvp", "nsf", "meanfield"])
        if specs and specs[-1].kind == kind:  # maximal runs: two adjacent segments of one family would be one segment
            kind = "radial" if kind != "radial" else "planar"
        if kind == "realnvp":
            specs.append(o.FlowSpec("realnvp", d, int(rng.integers(1, 3)), (int(rng.integers(2, 33)), int(rng.integers(2, 33)))))
        elif kind == "nsf":
            specs.append(o.FlowSpec("nsf", d, 1, (int(rng.integers(2, 33)), int(rng.integers(2, 33))), int(rng.choice([5, 8, 10])), 5.0))
        elif kind == "meanfield":
            specs.append(o.FlowSpec("meanfield", d, 1))
        else:
            specs.append(o.FlowSpec(kind, d, int(rng.integers(1, 4))))
    ths = []
    for sp in specs:
        t = o.init_params(sp, rng)
        if sp.kind in ("planar", "radial"):
            t = 0.4 * t
        elif sp.kind == "meanfield":
            t = t + 0.2 * rng.standard_normal(t.shape)
        else:
            t = t + 0.05 * rng.standard_normal(t.shape)
        ths.append(t.astype(npdt).astype(np.float64))
    th = np.concatenate(ths)
    bk = rng.choice(["std", "diag", "dense"])
    if bk == "std":
        q0, obase = nf.MvNormal(d), None
    elif bk == "diag":
        mu0, var0 = rng.standard_normal(d).astype(npdt), (rng.uniform(size=d) + 0.4).astype(npdt)
        q0 = nf.MvNormal(torch.tensor(mu0, device="cuda"), torch.tensor(var0, device="cuda"))
        obase = ("diag", mu0.astype(np.float64), np.sqrt(var0.astype(np.float64)))
    else:
        mu0 = rng.standard_normal(d).astype(npdt)
        A = rng.standard_normal((d, d)) / np.sqrt(d)
        Sig = (A @ A.T + 0.5 * np.eye(d)).astype(npdt)
        q0 = nf.MvNormal(torch.tensor(mu0, device="cuda"), torch.tensor(Sig, device="cuda"))
        obase = ("dense", mu0.astype(np.float64), np.linalg.cholesky(Sig.astype(np.float64)))
    tag = f"case {case}: d={d} {'f64' if f64 else 'f32'} base={bk} segments=" + "+".join(f"{s.kind}{s.nlayers}" for s in specs)
    try:
        segs = [nf.Flow(sp.kind, nf.MvNormal(d), sp.nlayers, sp.hdims, sp.K, sp.B, dtype=dt, device="cuda", theta=torch.tensor(t, dtype=dt, device="cuda"))
                for sp, t in zip(specs, ths)]
        flow = nf.create_flow(segs, q0)
        n = int(rng.choice([1, 31, 64, 100]))
        xs = nf.device_specific_rand(nf.PhiloxRNG(case), q0, n, dtype=dt, device="cuda")
        xs64 = xs.cpu().numpy().astype(np.float64)
        ys, ladj = nf.with_logabsdet_jacobian(flow.transform, xs)
        y_ref, l_ref = o.comp_fwd(specs, th, xs64)
        e_y, e_l = rel(ys.cpu().numpy(), y_ref), np.abs(ladj.cpu().numpy() - l_ref).max() / max(np.abs(l_ref).max(), 1.0)
        xr, lb = nf.with_logabsdet_jacobian(nf.inverse(flow.transform), ys)
        e_inv = rel(xr.cpu().numpy(), xs64)
        tmu, tvar = rng.standard_normal(d).astype(npdt), (rng.uniform(size=d) + 0.5).astype(npdt)
        tgt = nf.DiagGaussTarget(torch.tensor(tmu, device="cuda"), torch.tensor(tvar, device="cuda"))
        otgt = ("diaggauss", tmu.astype(np.float64), tvar.astype(np.float64))
        lo, go = o.comp_neg_elbo_value_and_grad(specs, th, otgt, xs64)
        lo = lo + (o.base_logpdf(obase, xs64) - o.std_normal_logpdf(xs64)).mean()
        e_v = abs(nf.elbo_batch(nf.PhiloxRNG(case), flow, tgt, n) + lo) / max(abs(lo), 1e-30)
        loss, g = nf.value_and_gradient(nf.elbo_batch, flow, tgt, xs)
        e_loss = abs(loss - lo) / max(abs(lo), 1e-30)
        e_g = np.abs(g.cpu().numpy() - go).max() / max(np.abs(go).max(), 1e-30)
        zi, li = o.comp_inv(specs, th, ys.cpu().numpy().astype(np.float64))
        llr = (o.base_logpdf(obase, zi) + li).mean()
        e_ll = abs(nf.loglikelihood(None, flow, ys) - llr) / max(abs(llr), 1e-30)
        # forward-KL training (and the closure pullback, nf_flow_bwd through the composition) on well-conditioned data
        ysd = (0.7 * xs).contiguous()
        ysd64 = ysd.cpu().numpy().astype(np.float64)
        lf, gf = nf.loglikelihood_value_and_gradient(flow, ysd)
        lfo, gfo = o.comp_neg_loglik_value_and_grad(specs, th, ysd64, obase)
        e_fl = abs(lf - lfo) / max(abs(lfo), 1e-30)
        e_fg = np.abs(gf.cpu().numpy() - gfo).max() / max(np.abs(gfo).max(), 1e-30)
        tm, tv = torch.tensor(tmu, device="cuda"), torch.tensor(tvar, device="cuda")
        lc, gc = nf.value_and_gradient(nf.elbo_batch, flow, lambda y: (-0.5 * (y - tm[:, None]) ** 2 / tv[:, None]).sum(0) -
                                       0.5 * torch.log(2 * np.pi * tv).sum(), xs)
        e_cg = np.abs(gc.cpu().numpy() - go).max() / max(np.abs(go).max(), 1e-30)
        has_planar = any(s.kind == "planar" for s in specs)
        ty, tg, ti = (1e-10, 1e-8, 1e-6) if f64 else (5e-5, 5e-4, 2e-2 if has_planar else 1e-3)
        ok = e_y < ty and e_l < 10 * ty and e_v < 20 * ty and e_loss < 20 * ty and e_g < tg and e_inv < ti and e_ll < (100 * ty if not f64 else 1e-8)
        ok = ok and e_fl < 100 * ty and e_fg < 4 * tg and e_cg < tg
        print(("ok   " if ok else "FAIL ") + tag + f" n={n}  y {e_y:.1e} ladj {e_l:.1e} inv {e_inv:.1e} elbo(rng) {e_v:.1e} loss {e_loss:.1e} grad {e_g:.1e} loglik {e_ll:.1e}"
              f" fkl {e_fl:.1e} fklgrad {e_fg:.1e} closure-grad {e_cg:.1e}")
        bad += 0 if ok else 1
    except nf.NFHipError as e:
        print("skip " + tag + f"  ({e})")
print(f"{bad} failures")
sys.exit(1 if bad else 0)
