import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import parity as P
orig = P.gradient
def patched(key, got, ref, rtol=P.GRAD_RTOL, floor=None):
    g, r = P._np(got), P._np(ref)
    e = np.abs(g - r)
    print(key, "maxerr", e.max(), "at", int(e.argmax()), "n>1e-3:", int((e > 1e-3 * np.abs(r).max()).sum()), "of", e.size)
    idx = np.argsort(-e)[:12]
    print("  worst idx", idx.tolist())
    print("  got", g[idx][:6], "ref", r[idx][:6])
    return orig(key, got, ref, rtol, floor)
P.gradient = patched
import test_gpu_parity as T
from __graft_entry__ import load_package
T.test_nsf_reference_default_k10_up_to_d32(load_package(), 32, 2, 77)
