"""Kernel-tuning aid: in-kernel timeline (clock64 deltas, block 0 / wave 0, first tile group) of the
weight-streaming reverse kernel k_wide_bwd at the cfg-4 shape.  Usage (needs a library built with NF_KERNEL_TRACE=1 python __graft_entry__.py --force): python tools/trace_wide.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

nf = load_package()
lib = nf.load_library()
dev = torch.device("cuda", 0)
D, N = 256, int(os.environ.get("N", 32768))
flow = nf.realnvp(nf.MvNormal(D), (256, 256), 1, paramtype=torch.float32, device=dev, seed=1)
tgt = nf.DiagGaussTarget(torch.randn(D, device=dev), torch.rand(D, device=dev) + 0.5)
ctx = nf.context_for(dev)
out = torch.zeros(flow.P + 1, device=dev)
vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
for i in range(3):
    lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), None, N, N, 1, 0, i, vp(out))
lib.nf_debug_trace(ctx.ptr, 1, None, 0)
lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), None, N, N, 1, 0, 9, vp(out))
buf = (C.c_int64 * 128)()
lib.nf_debug_trace(ctx.ptr, 0, buf, 128)
t = list(buf)
names = ["start", "x2 loaded", "L1 (4 chunks) + lrelu", "mask + stash a1", "L2 (8 chunks) + lrelu", "mask + stash a2",
         "L3 (8 chunks)", "element-wise + stash d3", "dX3 (8 chunks)", "scale d2", "stash d2", "dX2 (8 chunks)",
         "scale + stash d1", "dX1 (4 chunks) + x2bar stores"]
for ph in range(2):
    st = t[ph * 32: ph * 32 + 14]
    print("phase", "TS"[ph])
    for k in range(1, 14):
        print(f"   {names[k]:32s} +{st[k] - st[k - 1]:7d}")
    print(f"   group total {st[13] - st[0]}  (MFMA-ideal 262144 at 64 cycles per 32x32x2)")
