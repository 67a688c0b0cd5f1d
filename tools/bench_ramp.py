"""Per-step device time of the cfg-2 training step from the first step of a fresh process: how long a cold MI355X takes to reach its
sustained clocks (profiles/r3n_step_ramp.txt: ~60 steps / 40 ms).  bench.py pre-warms for that reason (--prewarm)."""
import ctypes as C, os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
nf = load_package(); lib = nf.load_library(); dev = torch.device("cuda", 0)
D, N = 64, 65536
flow = nf.realnvp(nf.MvNormal(D), (64, 64), 4, paramtype=torch.float32, device=dev, seed=123)
tgt = nf.DiagGaussTarget(torch.randn(D, device=dev), torch.rand(D, device=dev) + 0.5)
ctx = nf.context_for(dev)
theta = flow.theta.clone(); m = torch.zeros_like(theta); v = torch.zeros_like(theta)
vp = lambda t: C.c_void_p(t.data_ptr())
K = 400
marks = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
torch.cuda.synchronize()
marks[0].record()
for i in range(K):
    nf._lib.check(lib.nf_elbo_step(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(theta), vp(m), vp(v), N, 123, i, 1e-3, 0.9, 0.999, 1e-8, None, None))
    marks[i + 1].record()
torch.cuda.synchronize()
ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(K)]
for a in range(0, K, 20):
    blk = ms[a:a + 20]
    print(f"steps {a:3d}-{a+19:3d}: mean {sum(blk)/len(blk):.4f}  min {min(blk):.4f}  max {max(blk):.4f}")
