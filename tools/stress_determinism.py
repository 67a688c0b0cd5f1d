"""Race / hazard stress: the same training step repeated must give the same BITS (every reduction in the library has a
fixed order), at sizes where all waves of the chip are busy.  The two hardware hazards of round 3
both showed up as run-to-run differences first.  usage: python tools/stress_determinism.py [repeats]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

nf = load_package()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = "cuda"
cases = []
g = torch.Generator().manual_seed(0)


def diag(d):
    return nf.DiagGaussTarget(torch.randn(d, generator=g).to(dev), (torch.rand(d, generator=g) + 0.5).to(dev))


for d, hd, nl, n in ((64, [64, 64], 4, 65536), (64, [64, 64], 4, 65536 + 17), (63, [40, 64], 2, 50001), (64, [32, 32], 4, 65536), (20, [32, 32], 2, 9999)):
    cases.append((f"realnvp d{d} h{hd} x{nl} n{n}", nf.realnvp(nf.MvNormal(d), hd, nl, paramtype=torch.float32, seed=1), diag(d), n))
for d, K, nl, n in ((32, 8, 4, 131072), (32, 10, 2, 40000), (5, 10, 2, 7777)):
    cases.append((f"nsf d{d} K{K} x{nl} n{n}", nf.nsf(nf.MvNormal(d), [32, 32], K, 5.0, nl, paramtype=torch.float32, seed=2), diag(d), n))
for kind, d, nl, n in (("planar", 64, 10, 1 << 20), ("radial", 64, 10, 1 << 20), ("planar", 33, 16, 100001), ("radial", 9, 3, 100001)):
    f = (nf.planarflow if kind == "planar" else nf.radialflow)(nf.MvNormal(d), nl, paramtype=torch.float32, seed=3)
    cases.append((f"{kind} d{d} x{nl} n{n}", f.with_theta(f.theta * 0.3), diag(d), n))
cases.append(("realnvp wide d256 h256 x2 n32768", nf.realnvp(nf.MvNormal(256), [256, 256], 2, paramtype=torch.float32, seed=4).with_theta(
    nf.realnvp(nf.MvNormal(256), [256, 256], 2, paramtype=torch.float32, seed=4).theta * 0.5), diag(256), 32768))
bad = 0
for name, flow, tgt, n in cases:
    ref = None
    for r in range(reps):
        loss, grad = nf.value_and_gradient(nf.elbo_batch, flow, tgt, n, rng=nf.PhiloxRNG(9))
        cur = torch.cat([grad, torch.tensor([loss], device=dev, dtype=grad.dtype)])
        if ref is None:
            ref = cur.clone()
        elif not torch.equal(ref, cur):
            bad += 1
            print(f"NOT REPRODUCIBLE  {name}: run {r} differs from run 0 by {float((ref - cur).abs().max()):.3e} (|g|inf {float(ref.abs().max()):.3e})")
            break
    else:
        print(f"ok  {name}: {reps} runs, same bits (loss {float(ref[-1]):.6f})")
    # forward-KL as well
    ys = nf.rand(flow, min(n, 200000), nf.PhiloxRNG(5))
    a = nf.loglikelihood_value_and_gradient(flow, ys)[1].clone()
    for r in range(2):
        if not torch.equal(a, nf.loglikelihood_value_and_gradient(flow, ys)[1]):
            bad += 1
            print(f"NOT REPRODUCIBLE  {name}: forward-KL gradient")
            break
print(f"{bad} irreproducible cases")
sys.exit(1 if bad else 0)
