"""Per-step device time of the cfg-2 training step from the first step of a fresh process, WITH the shader clock read next to
every step (tools/probe/clock_probe.hip: core-clock counter against the constant 100 MHz counter inside a 10 us kernel), then the
same again after an idle pause in the same process.  Tells the two explanations of the cold-start ramp apart
(VERDICT r3 weak 4): if the second ramp (memory already touched, TLBs and caches warm, clocks fallen back) looks like the first,
it is the clocks / power management; if it is flat, it is first touch.
usage: python tools/bench_ramp.py [steps_per_phase] [idle_seconds]"""
import ctypes as C, os, subprocess, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
so = os.path.join(ROOT, "tools", "probe", "libclock_probe.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                    os.path.join(ROOT, "tools", "probe", "clock_probe.hip"), "-o", so], check=True)
probe = C.CDLL(so)
probe.clock_probe_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
nf = load_package(); lib = nf.load_library(); dev = torch.device("cuda", 0)
D, N = 64, 65536
K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
IDLE = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
flow = nf.realnvp(nf.MvNormal(D), (64, 64), 4, paramtype=torch.float32, device=dev, seed=123)
tgt = nf.DiagGaussTarget(torch.randn(D, device=dev), torch.rand(D, device=dev) + 0.5)
ctx = nf.context_for(dev)
theta = flow.theta.clone(); m = torch.zeros_like(theta); v = torch.zeros_like(theta)
vp = lambda t: C.c_void_p(t.data_ptr())
stream = torch.cuda.current_stream(dev).cuda_stream
clk = torch.zeros(2 * (K + 1), dtype=torch.int64, device=dev)


def smi(tag):
    for cmd in (["rocm-smi", "--showclocks", "--showpower"], ["amd-smi", "metric", "-c", "-p"]):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
            lines = [l for l in r.stdout.splitlines() if any(w in l.lower() for w in ("sclk", "mclk", "power", "gfx", "mem"))]
            print(f"# {tag}: {' '.join(cmd)} rc={r.returncode}")
            for l in lines[:12]:
                print("#   " + l.strip())
            return
        except Exception as e:  # noqa: BLE001
            print(f"# {tag}: {cmd[0]} unavailable ({e})")


def phase(name, first_step):
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(2 * K)]
    clk.zero_()
    torch.cuda.synchronize()
    probe.clock_probe_launch(stream, clk.data_ptr(), K, 1000)
    for i in range(K):
        marks[2 * i].record()
        nf._lib.check(lib.nf_elbo_step(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(theta), vp(m), vp(v), N, 123, first_step + i,
                                       1e-3, 0.9, 0.999, 1e-8, None, None))
        marks[2 * i + 1].record()
        probe.clock_probe_launch(stream, clk.data_ptr(), i, 1000)
    torch.cuda.synchronize()
    ms = [marks[2 * i].elapsed_time(marks[2 * i + 1]) for i in range(K)]
    c = clk.cpu().view(-1, 2).double()
    mhz = (100.0 * c[:, 0] / c[:, 1].clamp(min=1)).tolist()
    print(f"## {name}: sclk before the first step {mhz[K]:.0f} MHz")
    blk = 10
    for a in range(0, K, blk):
        b = ms[a:a + blk]; f = mhz[a:a + blk]
        print(f"steps {a:3d}-{a+blk-1:3d}: mean {sum(b)/len(b):.4f} ms  min {min(b):.4f}  max {max(b):.4f}   sclk after step: mean {sum(f)/len(f):6.0f} MHz  min {min(f):6.0f}  max {max(f):6.0f}"
              f"   ms x GHz {sum(x * y for x, y in zip(b, f)) / len(b) / 1e3:.4f}")


smi("idle, before anything ran")
phase("phase A: fresh process, cold GPU", 0)
smi("straight after phase A")
time.sleep(IDLE)
phase(f"phase B: same process after {IDLE:.1f} s idle (memory touched, TLB / caches warm)", K)
time.sleep(0.05)
phase("phase C: after 50 ms idle", 2 * K)
