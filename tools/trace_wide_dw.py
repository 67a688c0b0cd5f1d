"""Kernel-tuning aid: in-kernel timeline (clock64 deltas, workgroup 0 / wave 0, its first eight sample tiles) of the weight-
gradient GEMM k_wide_dw_b6 at the cfg-4 shape.  Needs a library built with NF_KERNEL_TRACE=1 python __graft_entry__.py --force.
usage: python tools/trace_wide_dw.py"""
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
for i in range(20):
    lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), None, N, N, 1, 0, i, vp(out))
lib.nf_debug_trace(ctx.ptr, 1, None, 0)
lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), None, N, N, 1, 0, 99, vp(out))
buf = (C.c_int64 * 128)()
lib.nf_debug_trace(ctx.ptr, 0, buf, 128)
t = list(buf)
names = ["requests issued", "MFMAs, sample group 0", "MFMAs, sample group 1", "wait for the requested tile", "split + LDS stores", "barrier"]
print("k_wide_dw_b6, workgroup 0 / wave 0 (the LAST launch of the step overwrites the earlier ones): clocks per stage and tile")
for i in range(8):
    st = t[7 * i: 7 * i + 7]
    if st[6] <= st[0]:
        break
    print(f"tile {i}: " + "  ".join(f"{names[k][:28]} {st[k + 1] - st[k]:6d}" for k in range(6)) + f"   total {st[6] - st[0]}")
