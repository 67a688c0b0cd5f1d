import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0,'/root/repo/oracle')
from __graft_entry__ import load_package
import nf_oracle as o
nf = load_package()
for d in (5, 6, 64):
    for n in (10, 32):
        spec = o.FlowSpec("realnvp", d, 2, (32, 32))
        rng = np.random.default_rng(1)
        th = (o.init_params(spec, rng) + 0.05*rng.standard_normal(o.param_count(spec))).astype(np.float32)
        flow = nf.Flow("realnvp", nf.MvNormal(d), 2, (32,32), dtype=torch.float32, device="cuda", theta=torch.tensor(th, device="cuda"))
        mu = rng.standard_normal(d).astype(np.float32); var=(rng.uniform(size=d)+0.5).astype(np.float32)
        tgt = nf.DiagGaussTarget(torch.tensor(mu,device="cuda"), torch.tensor(var,device="cuda"))
        xs = rng.standard_normal((d,n)).astype(np.float32)
        xt = torch.tensor(xs.T.copy(), device="cuda").t()
        lref, gref = o.neg_elbo_value_and_grad(spec, th.astype(np.float64), ("diaggauss", mu.astype(np.float64), var.astype(np.float64)), xs.astype(np.float64))
        eb = nf.elbo_batch(flow, tgt, xt)
        loss, g = nf.value_and_gradient(nf.elbo_batch, flow, tgt, xt)
        loss2, g2 = nf.value_and_gradient(nf.elbo_batch, flow, tgt, xt)
        print(d, n, "ref", lref, "elbo_batch", -eb, "vag", loss, loss2, "gerr", float(np.abs(g.cpu().numpy()-gref).max()/np.abs(gref).max()))
