"""Kernel-tuning aid: in-kernel timeline (clock64 deltas, workgroup 0 / wave 0) of the neural-spline reverse kernel k_rqs_bwd_coop6 at the
cfg-3 shape (the LAST coupling's launch of a step remains).  Needs a library built with NF_KERNEL_TRACE=1 python __graft_entry__.py --force."""
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
print(f"k_rqs_bwd_coop6, N = {N}: prologue {t[1] - t[0]}, groups {t[60] - t[1]}, fold {t[61] - t[60]}, slab write {t[62] - t[61]}, launch {t[62] - t[0]} clocks")
prev_end = t[1]
for g in range(4):
    b = t[2 + 12 * g: 14 + 12 * g]
    if b[9] <= b[0] or b[0] == 0:
        break
    line = f"  group {g}: home forward (to B0) {b[0] - prev_end}"
    last = b[0]
    for k in range(4):
        line += f" | tile {k}: chunk phase {b[1 + 2 * k] - last}, d2 sum {b[2 + 2 * k] - b[1 + 2 * k]}"
        last = b[2 + 2 * k]
    line += f" | closing home phase {b[9] - last}   total {b[9] - prev_end}"
    prev_end = b[9]
    print(line)
