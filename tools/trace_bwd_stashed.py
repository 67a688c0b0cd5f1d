"""Kernel-tuning aid: phase timeline (s_memtime ticks, block 0 / thread 0, coupling 0) of k_affine_bwd_stashed on the
benchmark workload.  Needs a library built with NF_KERNEL_TRACE=1 python __graft_entry__.py --force."""
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
for ph in range(2):
    b = 1 + ph * 4
    prev = t0 if ph == 0 else t[4]
    print(f"phase {'TS'[ph]}: stage+zero {t[b]-prev}  tiles {t[b+1]-t[b]}  wait-for-others {t[b+2]-t[b+1]}  fold+slab {t[b+3]-t[b+2]}")
names = ["loads issued", "element-wise (waits for its operands)", "dX3 (+stash d3, a1 loads)", "dW3 (+d2 scale)", "dX2 (+stash d2, x2 loads)",
         "dW2 (+d1 scale)", "dX1 (+stash d1, next tile's s / u loads)", "dW1 (+x2bar stores)"]
for ph in range(2):
    for ti in range(2):
        st = t[32 + ph * 24 + ti * 12: 32 + ph * 24 + ti * 12 + 9]
        if not st[0]:
            continue
        print(f"phase {'TS'[ph]} tile {ti}:")
        for i in range(1, 9):
            print(f"   {names[i-1]:44s} +{st[i]-st[i-1]:6d}")
print("coupling 0 total", t[8] - t0, "ticks")
