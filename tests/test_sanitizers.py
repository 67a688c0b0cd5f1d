"""CPU-side sanitizer jobs.  No GPU sanitizer exists on this pool, so:

  * oracle/nf_cpu_step.cpp (the C++/OpenMP restatement behind bench.py's cpu_baseline) is built with
    -fsanitize=address,undefined into a small driver (tests/host/nf_cpu_step_check.cpp) and its outputs are compared with
    oracle/nf_oracle.py -- the same pin tests/test_oracle.py applies to the optimised build;
  * the HOST side of every translation unit of libnfhip.so is built with `hipcc --offload-host-only
    -fsanitize=address,undefined` and tests/host/nf_api_host_check.hip drives nf_api.hip's descriptor validation, workspace
    sizing and arena carving through it (no device, no launch).
Any ASan / UBSan report makes the driver exit non-zero (halt_on_error, -fno-sanitize-recover).
"""
import os
import shutil
import struct
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import nf_oracle as o
from __graft_entry__ import CSRC, HIPCC, ROOT, SOURCES

SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="halt_on_error=1:detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
           OMP_NUM_THREADS="4")


def test_cpu_step_restatement_under_asan_ubsan(tmp_path):
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "nf_cpu_step_check")
    r = subprocess.run([cxx, *SAN, "-std=c++17", "-fopenmp", os.path.join(ROOT, "tests", "host", "nf_cpu_step_check.cpp"), "-o", exe, "-lm"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    d, hd, nl, n = 11, (24, 17), 2, 192  # odd d and widths; the restatement takes whole 64-sample tiles only
    spec = o.FlowSpec("realnvp", d, nl, hd)
    rng = np.random.default_rng(5)
    th = (o.init_params(spec, rng) + 0.05 * rng.standard_normal(o.param_count(spec))).astype(np.float32)
    mu, var = rng.standard_normal(d).astype(np.float32), (rng.uniform(size=d) + 0.5).astype(np.float32)
    xs = rng.standard_normal((d, n)).astype(np.float32)
    inp, out = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("5i", d, hd[0], hd[1], nl, n))
        for a in (th, mu, var, np.ascontiguousarray(xs.T)):
            f.write(a.astype(np.float32).tobytes())
    r = subprocess.run([exe, str(inp), str(out)], capture_output=True, text=True, env=ENV, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    raw = np.fromfile(out, dtype=np.float32)
    P = th.size
    assert raw.size == 2 * (P + 2)
    loss, grad = float(raw[0]), raw[2 : 2 + P].astype(np.float64)
    lo, go = o.neg_elbo_value_and_grad(spec, th.astype(np.float64), ("diaggauss", mu.astype(np.float64), var.astype(np.float64)),
                                       xs.astype(np.float64))
    assert abs(loss - lo) <= 2e-5 * abs(lo)
    assert np.abs(grad - go).max() <= 2e-4 * np.abs(go).max()
    assert abs(float(raw[1]) - np.linalg.norm(go)) <= 2e-4 * np.linalg.norm(go)
    th1 = raw[P + 4 :].astype(np.float64)  # theta after one Adam step on in-library draws: finite and moved by about lr
    assert np.isfinite(raw[P + 2]) and np.isfinite(raw[P + 3]) and np.all(np.isfinite(th1))
    step = np.abs(th1 - th.astype(np.float64))
    assert 0 < step.max() <= 1.01e-3


def test_api_host_side_under_asan_ubsan(tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    flags = ["--offload-host-only", *SAN, "-std=c++17", "-Wno-unused-value", "-Wno-comment", "-Wno-unused-result"]

    def cc(src, obj):
        r = subprocess.run([HIPCC, *flags, "-c", src, "-o", obj], capture_output=True, text=True)
        assert r.returncode == 0, (src, r.stderr[-3000:])
        return obj

    jobs = [(os.path.join(CSRC, s), str(tmp_path / s.replace(".hip", ".o"))) for s in SOURCES if s != "nf_api.hip"]
    jobs.append((os.path.join(ROOT, "tests", "host", "nf_api_host_check.hip"), str(tmp_path / "check.o")))  # includes nf_api.hip
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda j: cc(*j), jobs))
    # No HIP runtime is linked: the check must not need one.  Every HIP symbol the host objects reference gets a stand-in
    # that reports hipErrorNoDevice (100) -- a code path that reaches one fails loudly instead of touching a driver --
    # and the device binaries (__hip_fatbin_*) are empty.
    und = subprocess.run(["nm", "-u", *objs], capture_output=True, text=True).stdout.split()
    syms = sorted({t for t in und if t.startswith("hip") or t.startswith("__hip")})
    stub = ['#include <cstdio>', 'extern "C" {']
    for t in syms:
        if t.startswith("__hip_fatbin_"):
            stub.append(f"char {t}[64] = {{0}};")
        elif t == "__hipRegisterFatBinary":
            stub.append("void **__hipRegisterFatBinary(void *) { static void *h[1]; return h; }")
        elif t in ("__hipRegisterFunction", "__hipUnregisterFatBinary", "__hipRegisterVar", "__hipRegisterManagedVar"):
            stub.append(f"void {t}(...) {{}}")
        elif t == "hipGetErrorString":
            stub.append('const char *hipGetErrorString(int) { return "hip stand-in: no device"; }')
        else:
            stub.append(f"int {t}(...) {{ return 100; }}")
    stub.append("}")
    stub_src = tmp_path / "hip_standins.cpp"
    stub_src.write_text("\n".join(stub) + "\n")
    stub_obj = str(tmp_path / "hip_standins.o")
    r = subprocess.run(["g++", "-c", str(stub_src), "-o", stub_obj], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    exe = str(tmp_path / "nf_api_host_check")
    clangxx = os.path.join(os.path.dirname(os.path.realpath(HIPCC)), "..", "lib", "llvm", "bin", "clang++")
    r = subprocess.run([clangxx, *SAN, "-o", exe, *objs, stub_obj, "-ldl", "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "nf_api host check: ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
