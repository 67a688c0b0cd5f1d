import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
nf = load_package(); lib = nf.load_library()
d, hd, nl, n = 64, (64, 64), 4, 4096 + 5
flow = nf.realnvp(nf.MvNormal(d), hd, nl, paramtype=torch.float32, seed=3)
rng = np.random.default_rng(0)
tgt = nf.DiagGaussTarget(torch.tensor(rng.standard_normal(d), dtype=torch.float32, device="cuda"), torch.tensor(rng.uniform(size=d) + 0.5, dtype=torch.float32, device="cuda"))
stream = torch.cuda.current_stream().cuda_stream
ca, cb = nf.Context(0, stream), nf.Context(0, stream)
vp = lambda t: C.c_void_p(t.data_ptr())
tha, ma, va = flow.theta.clone(), torch.zeros_like(flow.theta), torch.zeros_like(flow.theta)
thb, mb, vb = flow.theta.clone(), torch.zeros_like(flow.theta), torch.zeros_like(flow.theta)
out, gn = torch.empty(flow.P + 1, device="cuda"), torch.empty(1, device="cuda")
for step in range(4):
    nf._lib.check(lib.nf_elbo_step(ca.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(tha), vp(ma), vp(va), n, 77, step, 1e-3, 0.9, 0.999, 1e-8, None, None))
    nf._lib.check(lib.nf_elbo_value_and_grad(cb.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(thb), None, n, n, 77, 0, step, vp(out)))
    nf._lib.check(lib.nf_adam_update(cb.ptr, 0, vp(thb), vp(out), vp(mb), vp(vb), flow.P, 1e-3, 0.9, 0.999, 1e-8, step + 1, vp(gn)))
    torch.cuda.synchronize()
    for nm, a, b in (("th", tha, thb), ("m", ma, mb), ("v", va, vb)):
        diff = (a - b).abs()
        print(step, nm, "ndiff", int((diff > 0).sum()), "max", float(diff.max()), "idx", int(diff.argmax()))
