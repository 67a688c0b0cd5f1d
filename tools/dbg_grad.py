"""Per-parameter-block gradient error of a RealNVP flow against the oracle (kernel debugging aid).
Usage: python tools/dbg_grad.py d h1 h2 nlayers n"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from __graft_entry__ import load_package
import nf_oracle as o
nf = load_package()
d, h1, h2, nl, n = (int(v) for v in sys.argv[1:6])
spec = o.FlowSpec("realnvp", d, nl, (h1, h2))
rng = np.random.default_rng(1)
th = (o.init_params(spec, rng) + 0.05 * rng.standard_normal(o.param_count(spec))).astype(np.float32)
flow = nf.Flow("realnvp", nf.MvNormal(d), nl, (h1, h2), dtype=torch.float32, device="cuda", theta=torch.tensor(th, device="cuda"))
mu = rng.standard_normal(d).astype(np.float32); var = (rng.uniform(size=d) + 0.5).astype(np.float32)
tgt = nf.DiagGaussTarget(torch.tensor(mu, device="cuda"), torch.tensor(var, device="cuda"))
xs = rng.standard_normal((d, n)).astype(np.float32)
xt = torch.tensor(xs.T.copy(), device="cuda").t()
lref, gref = o.neg_elbo_value_and_grad(spec, th.astype(np.float64), ("diaggauss", mu.astype(np.float64), var.astype(np.float64)), xs.astype(np.float64))
loss, g = nf.value_and_gradient(nf.elbo_batch, flow, tgt, xt)
g = g.cpu().numpy()
print("loss", loss, lref)
off = 0
for k in range(2 * nl):
    c = (d + 1) // 2 if k % 2 == 0 else d // 2
    m = d - c
    for net in "st":
        for nm, sz in (("W1", m * h1), ("b1", h1), ("W2", h1 * h2), ("b2", h2), ("W3", h2 * c), ("b3", c)):
            a, b = g[off:off + sz], gref[off:off + sz]
            e = np.abs(a - b).max() / max(1e-12, np.abs(b).max())
            print(f"coupling {k} {net}-net {nm}: rel err {e:.2e}   gpu[:3] {a[:3]}  ref[:3] {b[:3]}")
            off += sz
if len(sys.argv) > 6:
    c = (d + 1) // 2; m = d - c
    o3 = m * h1 + h1 + h1 * h2 + h2 + h2 * c
    a, b = g[o3:o3 + c], gref[o3:o3 + c]
    np.set_printoptions(precision=4, linewidth=200)
    print("b3 gpu", a[:40]); print("b3 ref", b[:40]); print("ref mean over blocks of 32:", b.reshape(-1, 32).mean(1))
    W3g, W3r = g[o3 - h2 * c:o3].reshape(h2, c), gref[o3 - h2 * c:o3].reshape(h2, c)
    print("W3 gpu rows0-1", W3g[:2, :8], "\nW3 ref rows0-1", W3r[:2, :8]); print("W3 ref row0 block mean", W3r[0].reshape(-1, 32).mean(1), "gpu col0", W3g[:8, 0], "ref col0", W3r[:8, 0])
