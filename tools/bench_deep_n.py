"""Deep RealNVP (1 / 3 hidden layers of 64, d = 64, 4 layers) training step at several batch sizes: where the per-phase fixed
work of k_deep_bwd (image staging, fold, slab write) stops mattering.  usage: python tools/bench_deep_n.py [N ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_configs as bc  # noqa: E402

nf, dev = bc.nf, bc.dev
g = torch.Generator().manual_seed(1)
tgt = nf.DiagGaussTarget(torch.randn(64, generator=g).to(dev), (torch.rand(64, generator=g) + 1e-3).to(dev))
for n in [int(a) for a in sys.argv[1:]] or [65536, 262144, 1048576]:
    for hd in ((64,), (64, 64, 64)):
        flow = nf.realnvp(nf.MvNormal(64), hd, 4, paramtype=torch.float32, device=dev, seed=123)
        r = bc.time_step(flow, tgt, n, 10, warmup=3)
        print(f"hidden {hd} N={n}: {r['ms_per_step']:.4f} ms/step  {r['kernels_us']}", flush=True)
