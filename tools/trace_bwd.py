"""Kernel-tuning aid: prints the in-kernel timeline (s_memtime deltas, block 0 / wave 0) of the
coupling reverse pass on the benchmark workload.  Usage (needs a library built with NF_KERNEL_TRACE=1 python __graft_entry__.py --force): python tools/trace_bwd.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

nf = load_package()
lib = nf.load_library()
dev = torch.device("cuda", 0)
D, N = 64, int(os.environ.get("N", 65536))
flow = nf.realnvp(nf.MvNormal(D), (64, 64), 4, paramtype=torch.float32, device=dev, seed=1)
tgt = nf.DiagGaussTarget(torch.randn(D, device=dev), torch.rand(D, device=dev) + 0.5)
ctx = nf.context_for(dev)
out = torch.zeros(flow.P + 1, device=dev)
vp = lambda t: C.c_void_p(t.data_ptr())
for i in range(3):
    lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), None, N, N, 1, 0, i, vp(out))
lib.nf_debug_trace(ctx.ptr, 1, None, 0)
lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), None, N, N, 1, 0, 9, vp(out))
buf = (C.c_int64 * 128)()
lib.nf_debug_trace(ctx.ptr, 0, buf, 128)
t = list(buf)
t0 = t[0]
names = ["start", "x2 loaded", "fwd L1-L3 (+stash x2,a1,a2)", "element-wise", "dX3 (+stash d3)", "dW3 (+d2 scale)",
             "dX2 (+stash d2)", "dW2 (+d1 scale)", "dX1 (+stash d1)", "dW1 (+x2bar stores)", "tail fence"]
print("kernel: staged phaseT @", t[1] - t0, " tiles done @", t[2] - t0, " folded+slab @", t[3] - t0)
print("        staged phaseS @", t[4] - t0, " tiles done @", t[5] - t0, " folded+slab @", t[6] - t0)
for ph in range(2):
    for ti in range(2):
        base = 8 + ph * 40 + ti * 12
        st = t[base:base + 11]
        if st[0] == 0:
            continue
        print(f"phase {'TS'[ph]} tile {ti}: start @{st[0]-t0}")
        for k in range(1, 11):
            print(f"   {names[k]:28s} +{st[k]-st[k-1]:7d}")
        print(f"   tile total {st[10]-st[0]}")
