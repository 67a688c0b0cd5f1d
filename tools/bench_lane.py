"""planar / radial ELBO step timing across targets and layer counts (Float32, d = 64, 1 M samples).
usage: python tools/bench_lane.py"""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
nf = load_package(); lib = nf.load_library(); dev = torch.device("cuda", 0)
vp = lambda t: C.c_void_p(t.data_ptr())
N, d = 1 << 20, 64
g = torch.Generator().manual_seed(1)
targets = {"diag": nf.DiagGaussTarget(torch.randn(d, generator=g).to(dev), (torch.rand(d, generator=g) + 0.5).to(dev)), "banana": nf.BananaTarget(d, 0.3, 4.0), "funnel": nf.FunnelTarget(d, -1.0, 1.5)}
for kind in ("planar", "radial"):
    for nl in (10, 16):
        flow = (nf.planarflow if kind == "planar" else nf.radialflow)(nf.MvNormal(d), nl, paramtype=torch.float32, device=dev, seed=3)
        with torch.no_grad(): flow.theta.mul_(0.1)
        ctx = nf.context_for(dev); out = torch.zeros(flow.P + 1, device=dev)
        for tn, tgt in targets.items():
            def step(i): nf._lib.check(lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), None, N, N, 1, 0, i, vp(out)))
            for i in range(3): step(i)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(10): step(3 + i)
            torch.cuda.synchronize()
            print(f"{kind} x{nl} {tn}: {1e3 * (time.perf_counter() - t0) / 10:.3f} ms/step")
