"""HBM-bound kernels at scale: ELBO step of planar / radial flows, d = 64, 10 layers, 1 M samples (fp32), with the
bytes each kernel must move and the resulting GB/s.  usage: python tools/bench_simple.py [N]"""
import ctypes as C
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
d, nl = 64, 10
g = torch.Generator().manual_seed(1)
tgt = nf.DiagGaussTarget(torch.randn(d, generator=g).to(dev), (torch.rand(d, generator=g) + 0.5).to(dev))
for kind in ("planar", "radial"):
    flow = (nf.planarflow if kind == "planar" else nf.radialflow)(nf.MvNormal(d), nl, paramtype=torch.float32, device=dev, seed=3)
    with torch.no_grad():
        flow.theta.mul_(0.1)
    ctx = nf.context_for(dev)
    out = torch.zeros(flow.P + 1, device=dev)

    def step(i):
        nf._lib.check(lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), None, N, N, 1, 0, i, vp(out)))

    # the event pool first (its creation idles the GPU for milliseconds), then enough steps for the clocks to reach their
    # sustained state (a cold MI355X needs ~40 ms of load: profiles/r3n_step_ramp.txt), then the timed steps
    lib.nf_prof_enable(ctx.ptr, 2)
    lib.nf_prof_enable(ctx.ptr, 0)
    for i in range(300):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(100):
        step(300 + i)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 100
    lib.nf_prof_enable(ctx.ptr, 2)
    for i in range(20):
        step(400 + i)
    torch.cuda.synchronize()
    row = N * d * 4
    lpp = 4  # layers per pass of the reverse kernel (float, d = 64)
    # fused forward: the per-layer inputs (stash) and ybar are written, nothing is read;
    # reverse: every layer's input once, gbar read + written once per pass of `lpp` layers
    need = {"simple_apply": row * (nl + 1), "simple_bwd": row * (nl + 2 * -(-nl // lpp))}
    print(f"{kind} d={d} layers={nl} N={N}: {1e3 * el:.3f} ms/step = {N / el / 1e6:.1f} M samples/s, loss {float(out[-1]):.4f}")
    a, c = C.c_double(0.0), C.c_int64(0)
    kname = "simple_step"
    lib.nf_prof_read(ctx.ptr, b"simple_step", C.byref(a), C.byref(c))
    if not c.value:  # planar, d <= 64, <= 16 layers: the matrix-pipe step (k_planar_step); radial: k_radial_step
        kname = "radial_step"
        lib.nf_prof_read(ctx.ptr, b"radial_step", C.byref(a), C.byref(c))
    if not c.value:
        kname = "planar_step"
        lib.nf_prof_read(ctx.ptr, b"planar_step", C.byref(a), C.byref(c))
    if c.value:  # the stash-free step: one launch, no activation traffic; SURVEY 8(d) algorithmic bytes: 8d + 4 per sample
        alg = N * (8 * d + 4)
        print(f"   {kname}: {1e3 * a.value:.1f} us; algorithmic {alg / 1e6:.0f} MB -> {alg / (a.value * 1e-3) / 1e12:.2f} TB/s "
              f"= {100 * alg / (a.value * 1e-3) / 8e12:.1f} % of the 8 TB/s HBM roofline (the kernel moves only parameter slabs: issue-bound)")
    for name, b in need.items():
        a, c = C.c_double(0.0), C.c_int64(0)
        lib.nf_prof_read(ctx.ptr, name.encode(), C.byref(a), C.byref(c))
        if not c.value:
            continue
        if c.value:
            print(f"   {name:13s} {1e3 * a.value:8.1f} us  {b / 1e6:8.1f} MB  -> {b / (a.value * 1e-3) / 1e12:.2f} TB/s ({b / (a.value * 1e-3) / 8e12:.0%} of 8 TB/s)")
    for name in (b"simple_finalize", b"finish_sum"):
        a, c = C.c_double(0.0), C.c_int64(0)
        lib.nf_prof_read(ctx.ptr, name, C.byref(a), C.byref(c))
        if c.value:
            print(f"   {name.decode():13s} {1e3 * a.value:8.1f} us")
    lib.nf_prof_enable(ctx.ptr, 0)
    # density evaluation / forward-KL side: logpdf(flow, ys) on the flow's own samples, then one forward-KL training step
    ys = nf.rand(flow, N, nf.PhiloxRNG(5))
    torch.cuda.synchronize()
    for name, fn in (("rand(flow, n)", lambda: nf.rand(flow, N, nf.PhiloxRNG(5))),
                     ("loglikelihood", lambda: nf.loglikelihood(None, flow, ys)),
                     ("forward-KL step", lambda: nf.loglikelihood_value_and_gradient(flow, ys))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        el2 = (time.perf_counter() - t0) / 5
        lib.nf_prof_enable(ctx.ptr, 2)
        fn()
        torch.cuda.synchronize()
        parts = []
        for kn in (b"base_sample", b"base_logpdf", b"simple_apply", b"target", b"sum2", b"simple_bwd", b"simple_finalize", b"finish_sum"):
            a, c = C.c_double(0.0), C.c_int64(0)
            lib.nf_prof_read(ctx.ptr, kn, C.byref(a), C.byref(c))
            if c.value:
                parts.append(f"{kn.decode()} {1e3 * a.value:.0f}us" + (f" x{c.value}" if c.value > 1 else ""))
        lib.nf_prof_enable(ctx.ptr, 0)
        print(f"   {name:16s} {1e3 * el2:8.3f} ms = {N / el2 / 1e6:7.1f} M samples/s   [{', '.join(parts)}]")
