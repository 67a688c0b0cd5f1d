"""Kernel-tuning aid: in-kernel timeline (clock64 deltas, workgroup 0 / wave 0, its first two tile groups) of the fused neural-spline forward
k_rqs_chain<.., FUSED, B6> at the cfg-3 shape.  Needs a library built with NF_KERNEL_TRACE=1 python __graft_entry__.py --force."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

nf = load_package()
lib = nf.load_library()
dev = torch.device("cuda", 0)
D, N = 32, int(os.environ.get("N", 131072))
flow = nf.nsf(nf.MvNormal(D), (32, 32), 8, 5.0, 4, paramtype=torch.float32, device=dev, seed=1)
tgt = nf.DiagGaussTarget(torch.randn(D, device=dev), torch.rand(D, device=dev) + 0.5)
ctx = nf.context_for(dev)
out = torch.zeros(flow.P + 1, device=dev)
vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
for i in range(20):
    lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), None, N, N, 1, 0, i, vp(out))
lib.nf_debug_trace(ctx.ptr, 1, None, 0)
lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), None, N, N, 1, 0, 99, vp(out))
torch.cuda.synchronize()
buf = (C.c_int64 * 128)()
lib.nf_debug_trace(ctx.ptr, 0, buf, 128)
t = list(buf)
for g in range(2):
    b = t[64 + 28 * g: 64 + 28 * g + 28]
    if b[27] <= b[0]:
        break
    print(f"group {g}: draws {b[1] - b[0]}, eight couplings {b[25] - b[1]}, target + stores + sums {b[27] - b[25]}, total {b[27] - b[0]}")
    last = b[1]
    for p in range(8):
        print(f"   coupling {p}: step {b[2 + 3 * p] - last}, barrier {b[3 + 3 * p] - b[2 + 3 * p]}, request of the next triples {b[4 + 3 * p] - b[3 + 3 * p]}")
        last = b[4 + 3 * p]
