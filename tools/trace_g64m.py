"""Kernel-tuning aid: in-kernel timeline (clock64 deltas, workgroup 0 / wave 0) of the Float64 RealNVP coupling kernel
k_g64m_apply (the LAST coupling applied overwrites the earlier ones).  Needs a library built with NF_KERNEL_TRACE=1."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

nf = load_package()
lib = nf.load_library()
dev = torch.device("cuda", 0)
D, N = int(os.environ.get("D", 64)), int(os.environ.get("N", 65536))
flow = nf.realnvp(nf.MvNormal(D), (64, 64), 4, paramtype=torch.float64, device=dev, seed=1)
ctx = nf.context_for(dev)
xs = nf.device_specific_rand(nf.PhiloxRNG(1), flow.dist, N)
for i in range(3):
    nf.with_logabsdet_jacobian(flow.transform, xs)
lib.nf_debug_trace(ctx.ptr, 1, None, 0)
nf.with_logabsdet_jacobian(flow.transform, xs)
torch.cuda.synchronize()
buf = (C.c_int64 * 128)()
lib.nf_debug_trace(ctx.ptr, 0, buf, 128)
t = list(buf)
for p in range(2):
    b = t[16 * p: 16 * p + 16]
    print(f"pass {p}: net staged {b[1] - b[0]}, pass total {b[14] - b[0]}")
    for i in range(3):
        st = b[2 + 4 * i: 6 + 4 * i]
        if st[2] > st[0] > 0:
            print(f"   tile {i}: x load + net forward {st[1] - st[0]}, element-wise + stores {st[2] - st[1]}")

# ---- the reverse kernel k_g64m_bwd (one training step; the last coupling's launch remains)
tgt = nf.DiagGaussTarget(torch.randn(D, device=dev, dtype=torch.float64), torch.rand(D, device=dev, dtype=torch.float64) + 0.5)
out = torch.zeros(flow.P + 1, device=dev, dtype=torch.float64)
vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
for i in range(3):
    lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), None, N, N, 1, 0, i, vp(out))
lib.nf_debug_trace(ctx.ptr, 1, None, 0)
lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), None, N, N, 1, 0, 9, vp(out))
torch.cuda.synchronize()
lib.nf_debug_trace(ctx.ptr, 0, buf, 128)
t = list(buf)
names = ["forward + element-wise", "output layer dX + dW", "hidden layer dX + dW", "first layer dX + dW + stores"]
print("k_g64m_bwd")
for ph in range(2):
    b = t[32 + 24 * ph: 32 + 24 * ph + 24]
    print(f"phase {ph}: net staged {b[1] - b[0]}, groups {b[22] - b[1]}, slab stores {b[23] - b[22]}, phase total {b[23] - b[0]}")
    for g in range(4):
        st = b[2 + 5 * g: 7 + 5 * g]
        if st[4] > st[0] > 0:
            print(f"   group {g}: " + "  ".join(f"{names[k]} {st[k + 1] - st[k]}" for k in range(4)) + f"   total {st[4] - st[0]}")
