"""Kernel-tuning aid: in-kernel timeline (clock64 deltas, workgroup 0 / wave 0) of k_deep_bwd's first phase (t net of the first
coupling processed): image staging, the wave's tiles stage by stage, fold, slab write.  Needs a library built with
NF_KERNEL_TRACE=1 python __graft_entry__.py --force.  usage: [HD=64 | HD=64,64,64] [N=65536] python tools/trace_deep_bwd.py"""
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
hd = tuple(int(h) for h in os.environ.get("HD", "64").split(","))
flow = nf.realnvp(nf.MvNormal(D), hd, 4, paramtype=torch.float32, device=dev, seed=1)
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
print(f"k_deep_bwd hidden {hd}, N = {N}: first phase of workgroup 0 / wave 0, clocks")
print(f"  image staged {t[1] - t[0]}, tiles {t[2] - t[1]}, fold {t[3] - t[2]}, slab write {t[4] - t[3]}   phase total {t[4] - t[0]}")
names = ["x2 arrives", "hidden layers forward", "output layer forward", "element-wise (waits for y1, ybar1)", "output layer dX + dW",
         "hidden layers dX + dW", "layer 0 dX (waits for ybar2) + dW + stores"]
for i in range(8):
    st = t[8 + 8 * i: 16 + 8 * i]
    if st[7] <= st[0]:
        break
    print(f"  tile {i}: " + "  ".join(f"{names[k]} {st[k + 1] - st[k]}" for k in range(7)) + f"   total {st[7] - st[0]}")
