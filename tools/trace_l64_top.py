"""Kernel-tuning aid: in-kernel timeline (clock64 deltas, workgroup 0; wave 0 = a dW wave, wave 4 = a dX wave) of
k_l64_nsf_top_bwd, the fused reverse of a spline coupling's output layer (the last coupling's launch of one training step
remains in the buffer).  Needs a library built with NF_KERNEL_TRACE=1."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

nf = load_package()
lib = nf.load_library()
dev = torch.device("cuda", 0)
D, N = int(os.environ.get("D", 32)), int(os.environ.get("N", 131072))
flow = nf.nsf(nf.MvNormal(D), (64, 64), 8, 3.0, 6, paramtype=torch.float32, device=dev, seed=1)
ctx = nf.context_for(dev)
# ---- the forward kernel k_l64_nsf_top_fwd (workgroup 0 / wave 0; the last coupling applied remains)
xs = nf.device_specific_rand(nf.PhiloxRNG(1), flow.dist, N)
for i in range(3):
    nf.with_logabsdet_jacobian(flow.transform, xs)
lib.nf_debug_trace(ctx.ptr, 1, None, 0)
nf.with_logabsdet_jacobian(flow.transform, xs)
torch.cuda.synchronize()
fb = (C.c_int64 * 128)()
lib.nf_debug_trace(ctx.ptr, 0, fb, 128)
f = list(fb)
print(f"k_l64_nsf_top_fwd: prologue {f[1] - f[0]}")
for i in range(8):
    s = f[2 + 7 * i: 9 + 7 * i]
    if s[6] > s[0] > 0:
        print(f"   pair {i}: activations -> LDS + wait {s[1] - s[0]}  matrix stage {s[2] - s[1]}  stores {s[3] - s[2]}  wait {s[4] - s[3]}"
              f"  spline {s[5] - s[4]}  wait + ladj {s[6] - s[5]}   total {s[6] - s[0]}")
tgt = nf.DiagGaussTarget(torch.randn(D, device=dev), torch.rand(D, device=dev) + 0.5)
out = torch.zeros(flow.P + 1, device=dev, dtype=torch.float32)
vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
for i in range(3):
    lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), None, N, N, 1, 0, i, vp(out))
lib.nf_debug_trace(ctx.ptr, 1, None, 0)
lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), None, N, N, 1, 0, 9, vp(out))
torch.cuda.synchronize()
buf = (C.c_int64 * 128)()
lib.nf_debug_trace(ctx.ptr, 0, buf, 128)
t = list(buf)
for w, name in ((0, "dW wave"), (1, "dX wave")):
    b = t[64 * w: 64 * w + 64]
    print(f"{name}: prologue (layer through LDS, slice) {b[1] - b[0]}, intervals {b[62] - b[1]}, slab stores {b[63] - b[62]}, total {b[63] - b[0]}")
    print(f"   interval 1: the next tile's loads arrived {b[56] - b[9]} after the spline stage began")
    for i in range(9):
        s = b[2 + 6 * i: 8 + 6 * i]
        if s[5] > s[0] > 0:
            print(f"   interval {i}: dW {s[1] - s[0]}  spline of the next tile {s[2] - s[1]}  dX {s[3] - s[2]}  wait {s[4] - s[3]}"
                  f"  cotangent sum + store {s[5] - s[4]}   total {s[5] - s[0]}")

# ---- HIDDEN=1 (library built with NF_KERNEL_TRACE and -DNF_TRACE_HIDDEN): the buffer is k_l64_hidden_bwd's, workgroup 0 / wave 0
if os.environ.get("HIDDEN"):
    b = t
    print(f"k_l64_hidden_bwd: weights staged {b[1] - b[0]}, tiles {b[60] - b[1]}, wait {b[61] - b[60]}, folds + slab {b[62] - b[61]}, total {b[62] - b[0]}")
    for i in range(5):
        s = b[2 + 9 * i: 11 + 9 * i]
        if s[8] > s[0] > 0:
            print(f"   tile {i}: loads + mask {s[1] - s[0]}  scratch {s[2] - s[1]}  dW1 {s[3] - s[2]}  dX1 {s[4] - s[3]}  x load + mask {s[5] - s[4]}"
                  f"  scratch + dW0 {s[6] - s[5]}  dX0 {s[7] - s[6]}  gbar += {s[8] - s[7]}   total {s[8] - s[0]}")
