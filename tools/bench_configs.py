"""Times the BASELINE.json configurations other than the headline one (they are parity-test cases,
not bench.py lines) and prints a small table.  Usage: python tools/bench_configs.py [--steps K]

cfg1  planar flow d=2, 10 layers, Banana target, batch 1024, Float64  (ELBO step)
cfg2  RealNVP d=64, 8 couplings, h=64, batch 65536                     (ELBO step)   [headline]
cfg2b RealNVP d=64, 8 couplings, h=32 (reference default widths)       (ELBO step)
cfg3  NSF d=32, 8 RQ-spline couplings, K=8, B=5, h=32, batch 131072    (ELBO step)
cfg4  RealNVP d=256, 16 couplings, h=256, batch 32768 = one GPU's shard of 262144/8  (ELBO step)
cfg5  RealNVP d=64 inverse + logdet + log q0 on 1 M samples           (loglikelihood)
cfg5t the same data set, one forward-KL TRAINING step (value + gradient of -loglikelihood, Adam)
cfg2c cfg 2 with the target as an arbitrary torch `logp` closure: nf_flow_fwd_keep + torch autograd of logp + nf_flow_bwd_kept
      (the path every user-defined target takes), next to the built-in-target step on the same flow
f64   Float64 coupling flows (scalar MLP, one thread per sample): the reference's own Float64 test shapes, timed
fwd   forward-only elbo_batch(rng, ...) evaluation of cfg 2 / cfg 3 (no gradient)
gen   the GENERAL coupling kernels (nf_generic64.hip: one thread per sample, scalar loops, atomics) on a shape the MFMA
      kernels do not build: NSF d=32, hidden [64,64] (the reference's docstring example nsf(q0, [64,64], 8, 3.0, 6),
      src/flows/neuralspline.jl:215), K=8, batch 131072 -- so that the cost of falling off the MFMA path is on record
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

nf = load_package()
lib = nf.load_library()
dev = torch.device("cuda", 0)
vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731


def time_step(flow, tgt, n, steps, warmup=5):
    ctx = nf.context_for(dev)
    theta = flow.theta.clone()
    m, v = torch.zeros_like(theta), torch.zeros_like(theta)
    out = torch.zeros(flow.P + 1, dtype=theta.dtype, device=dev)
    gn = torch.zeros(1, dtype=theta.dtype, device=dev)
    dt = 0 if theta.dtype == torch.float32 else 1

    def step(i):
        nf._lib.check(lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(theta), None, n, n, 123, 0, i, vp(out)))
        nf._lib.check(lib.nf_adam_update(ctx.ptr, dt, vp(theta), vp(out), vp(m), vp(v), flow.P, 1e-3, 0.9, 0.999, 1e-8, i + 1, vp(gn)))

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    # three blocks of `steps`; the fastest block counts (the pool's boxes stall for 1-30 ms now and then, which a mean over a
    # few steps of a 0.4 ms workload turns into a 4x outlier) and all three are printed
    blocks = []
    for r in range(3):
        t0 = time.perf_counter()
        for i in range(warmup + r * steps, warmup + (r + 1) * steps):
            step(i)
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / steps)
    el = min(blocks)
    steps = 3 * steps
    lib.nf_prof_enable(ctx.ptr, 2)
    for i in range(3):
        step(warmup + steps + i)
    torch.cuda.synchronize()
    kern = {}
    for name in (b"base_sample", b"pack_weights", b"affine_chain", b"rqs_chain", b"simple_apply", b"simple_step", b"planar_step", b"radial_step", b"simple_finalize", b"target", b"affine_bwd",
                 b"rqs_bwd", b"simple_bwd", b"wide_apply", b"wide_bwd", b"wide_dw", b"deep_chain", b"deep_bwd", b"g64m_apply", b"g64m_bwd", b"g64_apply", b"g64_bwd", b"l64_fwd", b"l64_couple", b"l64_top_fwd", b"l64_top_bwd", b"l64_hidden_bwd", b"l64_dw", b"l64_bwdx", b"reduce_slabs", b"adam"):
        a, c = C.c_double(0.0), C.c_int64(0)
        lib.nf_prof_read(ctx.ptr, name, C.byref(a), C.byref(c))
        if c.value:
            kern[name.decode()] = [round(1e3 * a.value, 1), c.value // 3]  # [avg us, launches per step]
    lib.nf_prof_enable(ctx.ptr, 0)
    return {"ms_per_step": round(1e3 * el, 4), "ms_per_step_blocks": [round(1e3 * b, 4) for b in blocks], "samples_per_s": round(n / el),
            "loss": float(out[flow.P]), "kernels_us": kern}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--only", default="", help="comma-separated subset, e.g. cfg3,cfg4")
    args = ap.parse_args()
    want = lambda name: not args.only or any(name.startswith(o) for o in args.only.split(","))  # noqa: E731
    res = {}
    g = torch.Generator().manual_seed(1)

    def dg(d, dt=torch.float32):
        return nf.DiagGaussTarget(torch.randn(d, generator=g).to(dev, dt), (torch.rand(d, generator=g) + 1e-3).to(dev, dt))

    if want("cfg1"):
        flow = nf.planarflow(nf.MvNormal(2), 10, paramtype=torch.float64, device=dev, seed=123)
        res["cfg1_planar_d2_f64_n1024"] = time_step(flow, nf.BananaTarget(2, 1.0, 10.0), 1024, args.steps)
    if want("cfg2_"):
        flow = nf.realnvp(nf.MvNormal(64), (64, 64), 4, paramtype=torch.float32, device=dev, seed=123)
        res["cfg2_realnvp_d64_h64_n65536"] = time_step(flow, dg(64), 65536, args.steps)
    if want("cfg2b"):
        flow = nf.realnvp(nf.MvNormal(64), (32, 32), 4, paramtype=torch.float32, device=dev, seed=123)
        res["cfg2b_realnvp_d64_h32_n65536"] = time_step(flow, dg(64), 65536, args.steps)
    if want("cfg3"):
        flow = nf.nsf(nf.MvNormal(32), (32, 32), 8, 5.0, 4, paramtype=torch.float32, device=dev, seed=123)
        res["cfg3_nsf_d32_k8_n131072"] = time_step(flow, dg(32), 131072, args.steps)
    if want("cfg4"):
        flow = nf.realnvp(nf.MvNormal(256), (256, 256), 8, paramtype=torch.float32, device=dev, seed=123)
        res["cfg4_realnvp_d256_h256_n32768_per_gpu"] = time_step(flow, dg(256), 32768, max(5, args.steps // 3), warmup=3)
    if want("cfg2c"):
        flow = nf.realnvp(nf.MvNormal(64), (64, 64), 4, paramtype=torch.float32, device=dev, seed=123)
        tgt = dg(64)
        mu_t, var_t = tgt.mu, tgt.var
        c0 = float(-0.5 * torch.log(2 * torch.pi * var_t).sum())

        def logp(ys):  # the same density as the built-in target, written by the "user" in torch
            return c0 - 0.5 * ((ys - mu_t[:, None]) ** 2 / var_t[:, None]).sum(0)

        n = 65536
        st = nf.setup(nf.Adam(1e-3), flow.theta)
        rng = nf.PhiloxRNG(123)

        def closure_step():
            l, g = nf.value_and_gradient(nf.elbo_batch, flow, logp, n, rng=rng)
            nf.adam_update(nf.Adam(1e-3), st, flow.theta, g)

        for _ in range(5):
            closure_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            closure_step()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / args.steps
        builtin = time_step(nf.realnvp(nf.MvNormal(64), (64, 64), 4, paramtype=torch.float32, device=dev, seed=123), tgt, n, args.steps)
        # the built-in target through the SAME Python call (value_and_gradient returns the loss as a host float: one
        # synchronisation per step in both forms)
        flow_b = nf.realnvp(nf.MvNormal(64), (64, 64), 4, paramtype=torch.float32, device=dev, seed=123)
        st_b = nf.setup(nf.Adam(1e-3), flow_b.theta)

        def builtin_step():
            l, g = nf.value_and_gradient(nf.elbo_batch, flow_b, tgt, n, rng=rng)
            nf.adam_update(nf.Adam(1e-3), st_b, flow_b.theta, g)

        for _ in range(5):
            builtin_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            builtin_step()
        torch.cuda.synchronize()
        el_b = (time.perf_counter() - t0) / args.steps
        # the library's share of the closure step, kernel by kernel
        ctx = nf.context_for(dev)
        lib.nf_prof_enable(ctx.ptr, 2)
        for _ in range(3):
            closure_step()
        torch.cuda.synchronize()
        kern = {}
        for name in (b"base_sample", b"base_logpdf", b"layout_convert", b"pack_weights", b"affine_chain", b"affine_bwd", b"reduce_slabs", b"adam"):
            a_, c_ = C.c_double(0.0), C.c_int64(0)
            lib.nf_prof_read(ctx.ptr, name, C.byref(a_), C.byref(c_))
            if c_.value:
                kern[name.decode()] = [round(1e3 * a_.value, 1), c_.value // 3]
        lib.nf_prof_enable(ctx.ptr, 0)
        res["cfg2c_realnvp_d64_h64_n65536_torch_closure_target"] = {
            "ms_per_step": round(1e3 * el, 4), "samples_per_s": round(n / el),
            "builtin_target_same_python_call_ms_per_step": round(1e3 * el_b, 4),
            "builtin_target_raw_abi_async_ms_per_step": builtin["ms_per_step"],
            "closure_over_builtin_same_call": round(el / el_b, 3), "library_kernels_us": kern,
            "note": "closure step = base draws + nf_flow_fwd_keep + the user's torch logp and its autograd + nf_flow_bwd_kept + Adam; "
                    "both Python forms read the loss back to the host every step"}
    if want("gen"):
        flow = nf.nsf(nf.MvNormal(32), (64, 64), 8, 3.0, 3, paramtype=torch.float32, device=dev, seed=123)
        res["gen_nsf_d32_h64_k8_n131072_general_kernels"] = time_step(flow, dg(32), 131072, max(3, args.steps // 10), warmup=2)
        flow = nf.realnvp(nf.MvNormal(64), (64, 64, 64), 4, paramtype=torch.float32, device=dev, seed=123)  # cfg 2 with a third hidden layer
        res["gen_realnvp_d64_h64x3_n65536"] = time_step(flow, dg(64), 65536, max(3, args.steps // 10), warmup=2)
        flow = nf.realnvp(nf.MvNormal(64), (64,), 4, paramtype=torch.float32, device=dev, seed=123)  # ... and with one
        res["gen_realnvp_d64_h64x1_n65536"] = time_step(flow, dg(64), 65536, max(3, args.steps // 10), warmup=2)
    if want("f64"):
        # Float64 coupling flows (test/flow.jl:7,72 runs RealNVP and NSF in Float64): the general kernels' scalar MLP, one thread
        # per sample -- the reference's test shape and a d = 64 case, so that the cost of this path is on record
        flow = nf.realnvp(nf.MvNormal(5), (32, 32), 2, paramtype=torch.float64, device=dev, seed=123)
        res["f64_realnvp_d5_h32_n65536_f64_mfma"] = time_step(flow, dg(5, torch.float64), 65536, max(3, args.steps // 10), warmup=2)
        flow = nf.realnvp(nf.MvNormal(64), (64, 64), 4, paramtype=torch.float64, device=dev, seed=123)
        res["f64_realnvp_d64_h64_n65536_f64_mfma"] = time_step(flow, dg(64, torch.float64), 65536, max(3, args.steps // 10), warmup=2)
        flow = nf.nsf(nf.MvNormal(5), (32, 32), 10, 30.0, 2, paramtype=torch.float64, device=dev, seed=123)
        res["f64_nsf_d5_h32_k10_n65536_f64_mfma"] = time_step(flow, dg(5, torch.float64), 65536, max(3, args.steps // 10), warmup=2)
        # the reference's DEFAULT constructor nsf(q0) = nsf(q0, [32, 32], 10, 30.0, 10; paramtype = Float64) (neuralspline.jl:232-234) on a
        # 32-dimensional base, two of its ten layers: 464 net outputs per coupling -- the output layer in six passes (round 6)
        flow = nf.nsf(nf.MvNormal(32), (32, 32), 10, 30.0, 2, paramtype=torch.float64, device=dev, seed=123)
        res["f64_nsf_d32_h32_k10_4couplings_n65536_f64_mfma"] = time_step(flow, dg(32, torch.float64), 65536, max(3, args.steps // 10), warmup=2)
    if want("fwd"):
        # forward-only objective evaluation, elbo_batch(rng, flow, logp, n) (src/objectives/elbo.jl:93-97): SURVEY 8(d)'s
        # "forward-only ELBO samples/s" -- draws + chain + target + mean in one launch, no stash (the B6 chain for RealNVP)
        for name, flow, n in (("fwd_cfg2_realnvp_d64_h64_n65536", nf.realnvp(nf.MvNormal(64), (64, 64), 4, paramtype=torch.float32, device=dev, seed=123), 65536),
                              ("fwd_cfg2_realnvp_d64_h64_n1M", nf.realnvp(nf.MvNormal(64), (64, 64), 4, paramtype=torch.float32, device=dev, seed=123), 1 << 20),
                              ("fwd_cfg3_nsf_d32_k8_n131072", nf.nsf(nf.MvNormal(32), (32, 32), 8, 5.0, 4, paramtype=torch.float32, device=dev, seed=123), 131072)):
            tgt = dg(flow.dist.d)
            ctx = nf.context_for(dev)
            val = C.c_double(0.0)
            for i in range(5):
                nf._lib.check(lib.nf_elbo_batch_rng(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), n, 123, 0, i, C.byref(val)))
            els = []
            for r in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(20):
                    nf._lib.check(lib.nf_elbo_batch_rng(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), n, 123, 0, 5 + 20 * r + i, C.byref(val)))
                torch.cuda.synchronize()
                els.append((time.perf_counter() - t0) / 20)
            res[name] = {"ms_per_call": round(1e3 * min(els), 4), "samples_per_s": round(n / min(els)), "elbo": val.value,
                         "note": "each call reads the scalar back (one synchronisation per call)"}
    if not want("cfg5"):
        for k, v in res.items():
            print(k, json.dumps(v))
        return
    # cfg 5: forward-KL path
    flow = nf.realnvp(nf.MvNormal(64), (64, 64), 4, paramtype=torch.float32, device=dev, seed=123)
    n = 1 << 20
    ys = nf.device_specific_rand(nf.PhiloxRNG(123), nf.MvNormal(64), n) * 2 + 1
    for _ in range(3):
        ll = nf.loglikelihood(None, flow, ys)
    torch.cuda.synchronize()
    els = []
    for _ in range(3):  # fastest of three blocks, as in time_step
        t0 = time.perf_counter()
        for _ in range(10):
            ll = nf.loglikelihood(None, flow, ys)
        torch.cuda.synchronize()
        els.append((time.perf_counter() - t0) / 10)
    el = min(els)
    res["cfg5_loglik_realnvp_d64_n1M"] = {"ms_per_call": 1e3 * el, "samples_per_s": n / el, "loglik": ll}
    # forward-KL training step on the same data set: train_flow(loglikelihood, flow, ys)
    ctx = nf.context_for(dev)
    theta = flow.theta.clone()
    m, v = torch.zeros_like(theta), torch.zeros_like(theta)
    out = torch.zeros(flow.P + 1, dtype=theta.dtype, device=dev)

    def fkl_step(i):
        nf._lib.check(lib.nf_loglikelihood_value_and_grad(ctx.ptr, C.byref(flow.desc), vp(theta), vp(ys), n, n, vp(out)))
        nf._lib.check(lib.nf_adam_update(ctx.ptr, 0, vp(theta), vp(out), vp(m), vp(v), flow.P, 1e-3, 0.9, 0.999, 1e-8, i + 1, None))

    for i in range(3):
        fkl_step(i)
    torch.cuda.synchronize()
    els = []
    for r in range(3):
        t0 = time.perf_counter()
        for i in range(3 + 10 * r, 13 + 10 * r):
            fkl_step(i)
        torch.cuda.synchronize()
        els.append((time.perf_counter() - t0) / 10)
    el = min(els)
    lib.nf_prof_enable(ctx.ptr, 2)
    fkl_step(13)
    torch.cuda.synchronize()
    kern = {}
    for name in (b"layout_convert", b"affine_chain", b"target", b"affine_bwd_inv", b"reduce_slabs", b"adam"):
        a, c = C.c_double(0.0), C.c_int64(0)
        lib.nf_prof_read(ctx.ptr, name, C.byref(a), C.byref(c))
        if c.value:
            kern[name.decode()] = [round(1e3 * a.value, 1), c.value]
    lib.nf_prof_enable(ctx.ptr, 0)
    res["cfg5t_forward_kl_step_realnvp_d64_n1M"] = {"ms_per_step": 1e3 * el, "samples_per_s": n / el, "loss": float(out[flow.P]),
                                                    "kernels_us": kern}
    for k, v in res.items():
        print(k, json.dumps(v))


if __name__ == "__main__":
    main()
