"""Kernel-tuning aid: where a wavefront's time goes in the six-term (bf16) chain WITHOUT a stash -- k_affine_chain<INVERSE, ..., B6>,
the kernel behind loglikelihood / logpdf / elbo_batch (BASELINE cfg 5) -- clock stamps of workgroup 0, waves 0 and 4 (the two waves of
one SIMD), second tile group, per coupling: s net | hand-over barrier + request | t net | tanh / exp / combine.
Needs a library built with NF_KERNEL_TRACE=1 python __graft_entry__.py --force."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

nf = load_package()
lib = nf.load_library()
dev = torch.device("cuda", 0)
D, N = 64, int(os.environ.get("N", 1 << 20))
flow = nf.realnvp(nf.MvNormal(D), (64, 64), 4, paramtype=torch.float32, device=dev, seed=1)
ctx = nf.context_for(dev)
ys = nf.device_specific_rand(nf.PhiloxRNG(1), flow.dist, N)
for i in range(3):
    nf.loglikelihood(None, flow, ys)
lib.nf_debug_trace(ctx.ptr, 1, None, 0)
nf.loglikelihood(None, flow, ys)
torch.cuda.synchronize()
buf = (C.c_int64 * 128)()
lib.nf_debug_trace(ctx.ptr, 0, buf, 128)
t = list(buf)
names = ["s net", "barrier + request", "t net", "tanh / exp / combine", "-> next coupling's first stamp (phase barrier + request)"]
print("position:                                              " + "".join(f"{p:>8d}" for p in range(8)))
for w in range(2):
    a = t[32 + w * 48: 32 + w * 48 + 40]
    print(f"wave {4 * w}")
    for i in range(5):
        row = []
        for p in range(8):
            lo = a[5 * p + i]
            hi = a[5 * p + i + 1] if i < 4 else (a[5 * (p + 1)] if p < 7 else 0)
            row.append(f"{hi - lo:8d}" if lo and hi else "       -")
        print(f"  {names[i]:52s}" + "".join(row))
    print(f"  eight couplings: {a[39] - a[0]} clocks")
