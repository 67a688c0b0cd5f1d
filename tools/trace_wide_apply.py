"""Kernel-tuning aid: in-kernel timeline (clock64 deltas, block 0 / wave 0, first tile group) of the weight-streaming
forward kernel k_wide_apply at the cfg-4 shape (TRAIN=1: the stashing form inside the training step).  Needs a library built with
NF_KERNEL_TRACE=1 python __graft_entry__.py --force.  NF_WIDE_FP32=1 traces the fp32-MFMA form."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

nf = load_package()
lib = nf.load_library()
dev = torch.device("cuda", 0)
D, N = 256, int(os.environ.get("N", 32768))
flow = nf.realnvp(nf.MvNormal(D), (256, 256), 1, paramtype=torch.float32, device=dev, seed=1)
ctx = nf.context_for(dev)
x = torch.randn(D, N, device=dev)
y = torch.empty_like(x)
ladj = torch.empty(N, device=dev)
vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
train = os.environ.get("TRAIN") is not None  # TRAIN=1: the stashing form inside the training step
tgt = nf.DiagGaussTarget(torch.randn(D, device=dev), torch.rand(D, device=dev) + 0.5)
out = torch.zeros(flow.P + 1, device=dev)


def run(i):
    if train:
        nf._lib.check(lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), None, N, N, 1, 0, i, vp(out)))
    else:
        nf._lib.check(lib.nf_flow_fwd(ctx.ptr, C.byref(flow.desc), vp(flow.theta), vp(x), N, vp(y), vp(ladj)))


for i in range(3):
    run(i)
lib.nf_debug_trace(ctx.ptr, 1, None, 0)
run(9)
torch.cuda.synchronize()
buf = (C.c_int64 * 128)()
lib.nf_debug_trace(ctx.ptr, 0, buf, 128)
t = list(buf)
names = {1: "x loads", 2: "layer 1 (4 chunks) + lrelu", 3: "masks", 4: "layer 2 (8 chunks) + lrelu", 5: "masks", 6: "layer 3 (8 chunks)"}
for base, net in ((0, "s net"), (16, "t net")):
    print(net)
    prev = t[base] if base == 0 else t[6]
    for k in range(1, 7):
        print(f"   {names[k]:30s} +{t[base + k] - prev:7d}")
        prev = t[base + k]
print(f"   element-wise, stores            +{t[23] - t[22]:7d}")
if t[32]:
    print("inside layer 2 of the t net's... last traced net, chunks 0-3: DMA issue + stash stores | GEMM (96 bf16 MFMAs = 3072) | barrier wait")
    for ib in range(4):
        b = 32 + 4 * ib
        print(f"   chunk {ib}: +{t[b + 1] - t[b]:6d} | +{t[b + 2] - t[b + 1]:6d} | +{t[b + 3] - t[b + 2]:6d}")
print(f"group total {t[23] - t[0]} clocks; bf16 MFMAs alone 2 x 1536 x 32 = 98304, fp32 MFMAs alone 2 x 2048 x 64 = 262144")
