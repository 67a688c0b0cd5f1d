import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from __graft_entry__ import load_package
import nf_oracle as o
nf = load_package()
for d, K, B, hd, nl in ((5, 10, 30.0, (32, 32), 2), (32, 10, 30.0, (32, 32), 2), (32, 10, 5.0, (32, 32), 2), (5, 10, 5.0, (32, 32), 2), (32, 10, 30.0, (32, 32), 1)):
    flow = nf.nsf(nf.MvNormal(d), list(hd), K, B, nl, paramtype=torch.float64, seed=3)
    gen = torch.Generator().manual_seed(d)
    flow = flow.with_theta(flow.theta + 0.05 * torch.randn(flow.P, generator=gen, dtype=torch.float64).to("cuda"))
    xs = nf.device_specific_rand(nf.PhiloxRNG(5), flow.dist, 333, dtype=torch.float64)
    ys, ladj = nf.with_logabsdet_jacobian(flow.transform, xs)
    xr, lb = nf.with_logabsdet_jacobian(nf.inverse(flow.transform), ys)
    spec = o.FlowSpec("nsf", d, nl, hd, K=K, B=B)
    th = flow.theta.cpu().numpy()
    yo, lo = o.flow_fwd(spec, th, xs.cpu().numpy())
    xo, lio = o.flow_inv(spec, th, yo)
    e = (xr - xs).abs().max().item()
    print(f"d={d} K={K} B={B} nl={nl}: device round trip max abs {e:.3e}; fwd vs oracle {np.abs(ys.cpu().numpy()-yo).max():.3e}; oracle round trip {np.abs(xo - xs.cpu().numpy()).max():.3e}; device inv vs oracle inv (on oracle ys) {np.abs(nf.with_logabsdet_jacobian(nf.inverse(flow.transform), torch.tensor(yo, device='cuda'))[0].cpu().numpy() - xo).max():.3e}")
