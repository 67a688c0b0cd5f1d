import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from __graft_entry__ import load_package
import nf_oracle as o
nf = load_package()
d, nl, n = 32, int(os.environ.get("NL", 1)), int(os.environ.get("N", 100))
spec = o.FlowSpec("nsf", d, nl, (32, 32), 10, 5.0)
rng = np.random.default_rng(d)
th = (o.init_params(spec, rng) + 0.05 * rng.standard_normal(o.param_count(spec))).astype(np.float32)
flow = nf.Flow("nsf", nf.MvNormal(d), nl, (32, 32), 10, 5.0, dtype=torch.float32, device="cuda", theta=torch.tensor(th, device="cuda"))
xs = (rng.standard_normal((d, n)) * 1.5).astype(np.float32)
mu, var = rng.standard_normal(d).astype(np.float32), (rng.uniform(size=d) + 0.5).astype(np.float32)
tgt = nf.DiagGaussTarget(torch.tensor(mu, device="cuda"), torch.tensor(var, device="cuda"))
otgt = ("diaggauss", mu.astype(np.float64), var.astype(np.float64))
x_t = torch.tensor(np.ascontiguousarray(xs.T), device="cuda").t()
if os.environ.get("PRE"):
    ys, ladj = nf.with_logabsdet_jacobian(flow.transform, x_t)
    if os.environ.get("PRE") == "2":
        xr, lb = nf.with_logabsdet_jacobian(nf.inverse(flow.transform), ys)
loss, g = nf.value_and_gradient(nf.elbo_batch, flow, tgt, x_t)
lr, gr = o.neg_elbo_value_and_grad(spec, th.astype(np.float64), otgt, xs.astype(np.float64))
g = g.cpu().numpy().astype(np.float64)
print("loss", loss, lr, "gmax", np.abs(gr).max())
for li in o.layers_flat_order(spec):
    sl = slice(li.offset, li.offset + li.nparams)
    e = np.abs(g[sl] - gr[sl])
    print("layer", li.kind, li.offset, li.nparams, "maxerr", e.max(), "at", int(e.argmax()), "of", np.abs(gr[sl]).max())
    # blocks of the net: W1 (m x h1) b1 W2 b2 W3 b3
    m, h = 16, 32
    sizes = [m * h, h, h * h, h, h * 29 * 16, 29 * 16]
    off = 0
    for nm, sz in zip(("W1", "b1", "W2", "b2", "W3", "b3"), sizes):
        ee = e[off:off + sz]
        print("   ", nm, ee.max(), int(ee.argmax()))
        off += sz
