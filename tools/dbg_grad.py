import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
from __graft_entry__ import load_package
import nf_oracle as o
nf = load_package()
for (d, h, nl) in ((64, 32, 1), (64, 32, 2), (5, 32, 2), (64, 64, 1)):
    for n in (10, 32, 33, 64, 128, 160):
        spec = o.FlowSpec("realnvp", d, nl, (h, h))
        rng = np.random.default_rng(1)
        th = (o.init_params(spec, rng) + 0.05*rng.standard_normal(o.param_count(spec))).astype(np.float32)
        flow = nf.Flow("realnvp", nf.MvNormal(d), nl, (h,h), dtype=torch.float32, device="cuda", theta=torch.tensor(th, device="cuda"))
        mu = rng.standard_normal(d).astype(np.float32); var=(rng.uniform(size=d)+0.5).astype(np.float32)
        tgt = nf.DiagGaussTarget(torch.tensor(mu,device="cuda"), torch.tensor(var,device="cuda"))
        xs = rng.standard_normal((d,n)).astype(np.float32)
        xt = torch.tensor(xs.T.copy(), device="cuda").t()
        lref, gref = o.neg_elbo_value_and_grad(spec, th.astype(np.float64), ("diaggauss", mu.astype(np.float64), var.astype(np.float64)), xs.astype(np.float64))
        loss, g = nf.value_and_gradient(nf.elbo_batch, flow, tgt, xt)
        ge = np.abs(g.cpu().numpy()-gref)
        print(d, h, nl, n, "loss err", abs(loss-lref)/abs(lref), "gerr", float(ge.max()/np.abs(gref).max()), "argmax", int(ge.argmax()), "P", len(gref))

print("---- per-segment error, d=64 h=32 nl=1 n=32")
d,h,nl,n=64,32,1,32
spec = o.FlowSpec("realnvp", d, nl, (h, h))
rng = np.random.default_rng(1)
th = (o.init_params(spec, rng) + 0.05*rng.standard_normal(o.param_count(spec))).astype(np.float32)
flow = nf.Flow("realnvp", nf.MvNormal(d), nl, (h,h), dtype=torch.float32, device="cuda", theta=torch.tensor(th, device="cuda"))
mu = rng.standard_normal(d).astype(np.float32); var=(rng.uniform(size=d)+0.5).astype(np.float32)
tgt = nf.DiagGaussTarget(torch.tensor(mu,device="cuda"), torch.tensor(var,device="cuda"))
xs = rng.standard_normal((d,n)).astype(np.float32)
xt = torch.tensor(xs.T.copy(), device="cuda").t()
lref, gref = o.neg_elbo_value_and_grad(spec, th.astype(np.float64), ("diaggauss", mu.astype(np.float64), var.astype(np.float64)), xs.astype(np.float64))
loss, g = nf.value_and_gradient(nf.elbo_batch, flow, tgt, xt)
g = g.cpu().numpy()
off = 0
for k in range(2):
    for net in "st":
        for nm, sz in (("W1", 32*32), ("b1", 32), ("W2", 32*32), ("b2", 32), ("W3", 32*32), ("b3", 32)):
            e = np.abs(g[off:off+sz]-gref[off:off+sz]).max()/max(1e-12, np.abs(gref[off:off+sz]).max())
            print(f"coupling {k} {net}-net {nm}: rel err {e:.2e}")
            off += sz
