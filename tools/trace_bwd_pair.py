"""Kernel-tuning aid: who waits for whom in k_affine_bwd_pair (clock stamps of block 0, pair 0, coupling 0: the producer's
lane 0 and the consumer's lane 0 around the three barriers of a tile).  Needs a library built with
NF_KERNEL_TRACE=1 python __graft_entry__.py --force."""
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
vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
for i in range(3):
    lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), None, N, N, 1, 0, i, vp(out))
lib.nf_debug_trace(ctx.ptr, 1, None, 0)
lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), None, N, N, 1, 0, 9, vp(out))
buf = (C.c_int64 * 128)()
lib.nf_debug_trace(ctx.ptr, 0, buf, 128)
t = list(buf)
t0 = t[0]
names = ["prologue (operands, d3 -> LDS | a2 requested)", "wait at B1", "dX3, d2 -> LDS | dW3, a1 requested", "wait at B2",
         "dX2, d1 -> LDS | dW2, x2 requested", "wait at B3", "dX1, x2bar | dW1"]
for ph in range(2):
    for ti in range(2):
        a = t[ph * 16 + ti * 8: ph * 16 + ti * 8 + 8]
        b = t[64 + ph * 16 + ti * 8: 64 + ph * 16 + ti * 8 + 8]
        if not a[0]:
            continue
        print(f"phase {'TS'[ph]} tile {ti}: starts at +{a[0]-t0} (producer) / +{b[0]-t0} (consumer)")
        for i in range(7):
            print(f"   {names[i]:50s} producer +{a[i+1]-a[i]:6d}   consumer +{b[i+1]-b[i]:6d}")
bn = ["stage the image (own share)", "wait: image staged", "TILES", "wait: all tiles done", "fold (consumer) | s, u requested; barrier",
      "slab write (own share)", "wait: slab written"]
for ph in range(2):
    a, b = t[32 + ph * 8: 40 + ph * 8], t[96 + ph * 8: 104 + ph * 8]
    if not a[0]:
        continue
    print(f"phase {'TS'[ph]} boundary stamps: phase starts at +{a[0]-t0} (producer) / +{b[0]-t0} (consumer)")
    for i in range(7):
        print(f"   {bn[i]:50s} producer +{a[i+1]-a[i]:6d}   consumer +{b[i+1]-b[i]:6d}")
    print(f"   {'phase, first stamp -> last':50s} producer  {a[7]-a[0]:6d}   consumer  {b[7]-b[0]:6d}")
print("first tile start -> last stamp:", max(t) - t0, "ticks")
