"""Kernel-tuning aid: where a wavefront's time goes in the fused ELBO forward k_affine_chain<FUSED, STASH> on the benchmark
workload (clock stamps of block 0, waves 0 and 4 -- the two waves of one SIMD -- around the s net, the t net, the combine +
stash stores and the barrier of every coupling).  Needs a library built with NF_KERNEL_TRACE=1 python __graft_entry__.py --force."""
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
names = ["s net", "t net", "combine + s/u stores", "barrier"]
print("position:              " + "".join(f"{p:>8d}" for p in range(8)))
for w in range(2):
    a = t[32 + w * 64: 32 + w * 64 + 32]  # [4 pos + i]: after the s net, the t net, the combine, the barrier
    print(f"wave {4 * w}")
    for i in range(4):
        row = []
        for p in range(8):
            prev = a[4 * p + i - 1] if 4 * p + i else None
            row.append(f"{a[4 * p + i] - prev:8d}" if prev else "       -")
        print(f"  {names[i]:22s}" + "".join(row))
    print(f"  first stamp -> last stamp: {a[31] - a[0]} ticks")
