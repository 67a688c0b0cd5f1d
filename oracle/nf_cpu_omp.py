"""Loader / timer of oracle/nf_cpu_step.cpp, the C++ / OpenMP restatement of the reference's RealNVP reverse-KL step.

TEST / MEASUREMENT INFRASTRUCTURE ONLY: imported by tests/ (pinned against nf_oracle.py) and by bench.py's
`cpu_baseline` leg.  The shared object is compiled at run time ON THE BOX WHOSE CORES ARE TIMED (`g++ -O3 -march=native
-fno-math-errno -fopenmp`), into a temporary directory -- a binary built in the (different) build container could use
instructions the GPU box's host lacks, or miss its AVX-512.  Compiled with -ffast-math (vectorised exp / tanh / log), LINKED
without it (the link step is what would pull in crtfastmath.o and switch the importing process to flush-to-zero).
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "nf_cpu_step.cpp")
_lib = None


def load():
    """Compile (once per process) and dlopen.  Returns None when no C++ compiler is available."""
    global _lib
    if _lib is not None:
        return _lib
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        return None
    tmp = tempfile.mkdtemp(prefix="nfcpu_")
    obj, out = os.path.join(tmp, "nf_cpu_step.o"), os.path.join(tmp, "libnfcpu.so")
    # compile with -ffast-math (vectorised exp / tanh / log through libmvec), LINK without it: it is the link step
    # that pulls in crtfastmath.o, whose constructor would switch the importing process to flush-to-zero
    for cmd in ([cxx, "-O3", "-march=native", "-ffast-math", "-fopenmp", "-fPIC", "-c", SRC, "-o", obj],
                [cxx, "-shared", "-fopenmp", "-o", out, obj, "-lm"]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("nf_cpu_step.cpp failed to build:\n" + r.stderr[-2000:])
    lib = C.CDLL(out)
    lib.nfcpu_param_count.restype = C.c_long
    lib.nfcpu_param_count.argtypes = [C.c_int] * 4
    F = C.POINTER(C.c_float)
    lib.nfcpu_realnvp_step.restype = C.c_int
    lib.nfcpu_realnvp_step.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, F, F, F, F, F, F, C.c_long, C.c_uint64, C.c_int, C.c_float,
                                       F, F, F, C.c_int]
    _lib = lib
    return lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def value_and_grad(d, hdims, nlayers, theta, mu, var, xs, nthreads=0):
    """(loss, grad) of -elbo_batch on supplied xs (d, N) -- for the parity pin against nf_oracle."""
    lib = load()
    th = np.ascontiguousarray(theta, dtype=np.float32).copy()
    x = np.ascontiguousarray(xs.T, dtype=np.float32)
    n = x.shape[0]
    g = np.zeros_like(th)
    loss, gn = C.c_float(0), C.c_float(0)
    rc = lib.nfcpu_realnvp_step(d, hdims[0], hdims[1], nlayers, _fp(th), None, None, _fp(np.ascontiguousarray(mu, dtype=np.float32)),
                                _fp(np.ascontiguousarray(var, dtype=np.float32)), _fp(x), n, 0, 1, 0.0, C.byref(loss), C.byref(gn), _fp(g), nthreads)
    assert rc == 0, rc
    return float(loss.value), g


def time_training_steps(d, hdims, nlayers, batch, threads, seconds_budget=15.0, max_steps=50):
    """Full training steps (draws, forward, reverse pass, Adam, norm) at `batch`, `threads` OpenMP threads."""
    import nf_oracle as orc

    lib = load()
    if lib is None:
        return None
    spec = orc.FlowSpec("realnvp", d, nlayers, tuple(hdims))
    rng = np.random.default_rng(123)
    theta = orc.init_params(spec, rng, dtype=np.float32)
    assert lib.nfcpu_param_count(d, hdims[0], hdims[1], nlayers) == theta.size
    mu = rng.standard_normal(d).astype(np.float32)
    var = (rng.uniform(size=d) + 1e-3).astype(np.float32)
    m, v = np.zeros_like(theta), np.zeros_like(theta)
    loss, gn = C.c_float(0), C.c_float(0)

    def step(i):
        rc = lib.nfcpu_realnvp_step(d, hdims[0], hdims[1], nlayers, _fp(theta), _fp(m), _fp(v), _fp(mu), _fp(var), None, batch, 123 + i,
                                    i + 1, 1e-3, C.byref(loss), C.byref(gn), None, threads)
        assert rc == 0, rc

    step(0)
    times, t_all, i = [], time.perf_counter(), 1
    while len(times) < max_steps and (time.perf_counter() - t_all < seconds_budget or len(times) < 1):
        t0 = time.perf_counter()
        step(i)
        times.append(time.perf_counter() - t0)
        i += 1
    med = float(np.median(times))
    assert np.isfinite(loss.value) and np.isfinite(gn.value)
    return {"value": batch / med, "ms_per_step": 1e3 * med, "steps": len(times), "threads": threads, "loss": float(loss.value)}
