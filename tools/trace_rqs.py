"""Kernel-tuning aid: in-kernel timeline (clock64 deltas, block 0 / wave 0, first tile) of the
neural-spline reverse kernel k_rqs_bwd at the cfg-3 shape.  Usage (needs a library built with NF_KERNEL_TRACE=1 python __graft_entry__.py --force): python tools/trace_rqs.py"""
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
flow = nf.nsf(nf.MvNormal(D), (32, 32), 8, 5.0, 1, paramtype=torch.float32, device=dev, seed=1)
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
if os.environ.get("NF_RQS_BWD_PERWAVE") is None:
    # cooperative kernel (the default): block 0 / wave 0, first tile group
    print(f"home: loads + L1 + L2 + transposes   +{t[1] - t[0]:7d}   (MFMA-ideal 2048)")
    print(f"barrier B0                           +{t[2] - t[1]:7d}")
    prev = t[2]
    for k in range(4):
        b = 3 + 6 * k
        print(f"tile {k}: a2 reload + L3 chunk GEMM     +{t[b] - prev:7d}   (3072)")
        print(f"        spline inverse + reverse (x2)  +{t[b + 1] - t[b]:7d}")
        print(f"        dX3 chunk + slot write         +{t[b + 2] - t[b + 1]:7d}   (3072)")
        print(f"        dW3 (3 x transpose + GEMM)     +{t[b + 3] - t[b + 2]:7d}   (3072)")
        print(f"        barrier B1 + d2 sum            +{t[b + 4] - t[b + 3]:7d}")
        print(f"        barrier B2                     +{t[b + 5] - t[b + 4]:7d}")
        prev = t[b + 5]
    print(f"home: layers 2 and 1 + stores        +{t[27] - prev:7d}   (MFMA-ideal 4096)")
    print(f"group total {t[27] - t[0]}  (MFMA-ideal {672 * 64})")
    if t[103] > t[101]:
        print(f"all groups of workgroup 0: {t[102] - t[100]} shader clocks in {(t[103] - t[101]) / 100:.1f} us = {100.0 * (t[102] - t[100]) / (t[103] - t[101]):.0f} MHz while the kernel ran")
    sys.exit(0)

print(f"loads + L1 + L2 (+stash)        +{t[1] - t[0]:7d}   (MFMA-ideal 2048)")
for ch in range(4):
    b = 2 + 4 * ch
    prev = t[b - 1]
    print(f"chunk {ch}: L3 GEMM (3 blocks)     +{t[b] - prev:7d}   (3072)")
    print(f"         spline inverse+reverse  +{t[b + 1] - t[b]:7d}")
    print(f"         dX3 (3 blocks)          +{t[b + 2] - t[b + 1]:7d}   (3072)")
    print(f"         dW3 (stash+3 blocks)    +{t[b + 3] - t[b + 2]:7d}   (3072)")
print(f"stores, layers 2 and 1           +{t[18] - t[17]:7d}   (MFMA-ideal 4096)")
print(f"tile total {t[18] - t[0]}  (MFMA-ideal {41 * 1024})")
