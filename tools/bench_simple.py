"""HBM-bound kernels at scale: ELBO step of planar / radial flows, d = 64, 10 layers, 1 M samples (fp32), with the
bytes each kernel must move and the resulting GB/s.  usage: python tools/bench_simple.py [N]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

nf = load_package()
lib = nf.load_library()


def issue_slots(kernel_key):
    """Vector-issue slots of one pass through a kernel's code, from the built object's ISA (tools/isa_stats.py's extraction):
    every v_* instruction one slot (2 clocks of a 32-lane SIMD per wave64 instruction; a packed-f32 instruction is one slot for
    two values), quarter-rate transcendentals (exp, log, rcp, sqrt, rsq, sin, cos) four slots, fp32 matrix instructions their
    pipe time in slots.  The lane-per-sample step kernels are one unrolled tile loop (every layer of
    the instantiation written out, executed when nl = NL as here), so the static count is the count per 32-sample tile plus
    a few hundred instructions of prologue / epilogue -- an over-estimate of the per-tile work by < 5 %.  None without
    the LLVM binutils (this is a tuning aid, not a product path)."""
    import re
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin/"
    obj = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "normalizingflows.jl_amd", "build", "nf_simple.o")
    try:
        tmp = "/tmp/nfhip_isa_simple"
        os.makedirs(tmp, exist_ok=True)
        fb, co = os.path.join(tmp, "x.fatbin"), os.path.join(tmp, "x.co")
        subprocess.run([llvm + "llvm-objcopy", "--dump-section", f".hip_fatbin={fb}", obj], check=True, capture_output=True)
        subprocess.run([llvm + "clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        dis = subprocess.run([llvm + "llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
    except Exception:  # noqa: BLE001
        return None
    syms = re.findall(r"^[0-9a-f]+ <(\S+)>:$", dis, flags=re.M)
    dem = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True).stdout.split("\n")
    for m, dn in zip(syms, dem):
        if kernel_key in dn:
            body = dis.split(f"<{m}>:\n", 1)[1].split("\n\n", 1)[0].splitlines()
            ops = [l.split()[0] for l in body if l.strip()]
            trans = sum(o.startswith(("v_exp", "v_log", "v_rcp", "v_sqrt", "v_rsq", "v_sin", "v_cos")) for o in ops)
            valu = sum(o.startswith("v_") for o in ops) - trans
            # fp32 matrix instructions in 2-clock slots of their pipe time (they share the SIMD with the vector stream): 32x32x2 64 clocks, 16x16x4 32
            mf = 32 * sum(o.startswith("v_mfma_f32_32x32x2") for o in ops) + 16 * sum(o.startswith("v_mfma_f32_16x16x4") for o in ops)
            nm = sum(o.startswith("v_mfma") for o in ops)
            return valu - nm + 4 * trans + mf, valu - nm, trans, mf
    return None
dev = torch.device("cuda", 0)
vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
d, nl = 64, 10
g = torch.Generator().manual_seed(1)
tgt = nf.DiagGaussTarget(torch.randn(d, generator=g).to(dev), (torch.rand(d, generator=g) + 0.5).to(dev))
for kind in ("planar", "radial"):
    flow = (nf.planarflow if kind == "planar" else nf.radialflow)(nf.MvNormal(d), nl, paramtype=torch.float32, device=dev, seed=3)
    with torch.no_grad():
        flow.theta.mul_(0.1)
    ctx = nf.context_for(dev)
    out = torch.zeros(flow.P + 1, device=dev)

    def step(i):
        nf._lib.check(lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(flow.theta), None, N, N, 1, 0, i, vp(out)))

    # the event pool first (its creation idles the GPU for milliseconds), then enough steps for the clocks to reach their
    # sustained state (a cold MI355X needs ~40 ms of load: profiles/r3n_step_ramp.txt), then the timed steps
    lib.nf_prof_enable(ctx.ptr, 2)
    lib.nf_prof_enable(ctx.ptr, 0)
    for i in range(300):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(100):
        step(300 + i)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 100
    lib.nf_prof_enable(ctx.ptr, 2)
    for i in range(20):
        step(400 + i)
    torch.cuda.synchronize()
    row = N * d * 4
    lpp = 4  # layers per pass of the reverse kernel (float, d = 64)
    # fused forward: the per-layer inputs (stash) and ybar are written, nothing is read;
    # reverse: every layer's input once, gbar read + written once per pass of `lpp` layers
    need = {"simple_apply": row * (nl + 1), "simple_bwd": row * (nl + 2 * -(-nl // lpp))}
    print(f"{kind} d={d} layers={nl} N={N}: {1e3 * el:.3f} ms/step = {N / el / 1e6:.1f} M samples/s, loss {float(out[-1]):.4f}")
    a, c = C.c_double(0.0), C.c_int64(0)
    kname = "simple_step"
    lib.nf_prof_read(ctx.ptr, b"simple_step", C.byref(a), C.byref(c))
    if not c.value:  # planar, d <= 64, <= 16 layers: the matrix-pipe step (k_planar_step); radial: k_radial_step
        kname = "radial_step"
        lib.nf_prof_read(ctx.ptr, b"radial_step", C.byref(a), C.byref(c))
    if not c.value:
        kname = "planar_step"
        lib.nf_prof_read(ctx.ptr, b"planar_step", C.byref(a), C.byref(c))
    if c.value:  # the stash-free step: one launch, no activation traffic; SURVEY 8(d) algorithmic bytes: 8d + 4 per sample
        alg = N * (8 * d + 4)
        print(f"   {kname}: {1e3 * a.value:.1f} us; algorithmic {alg / 1e6:.0f} MB -> {alg / (a.value * 1e-3) / 1e12:.2f} TB/s "
              f"= {100 * alg / (a.value * 1e-3) / 8e12:.1f} % of the 8 TB/s HBM roofline (the kernel moves only parameter slabs: issue-bound)")
        # the second roofline: vector issue.  MI355X_MICROARCH.md: a wave64 fp32 vector instruction passes a SIMD
        # in 2 clocks (32 lanes per clock: 157.3 TFLOP/s = 256 CUs x 4 SIMDs x 32 lanes x 2 flop x 2.4 GHz), a quarter-rate
        # transcendental in 8, and ONE wave issues at most every ~4 clocks -- so with the two waves per SIMD these kernels run
        # the floor of a launch is (tiles per SIMD) x (slots per tile) x 2 clocks.  (Rounds 3-5 priced a slot at 4 clocks and
        # called these kernels "issue-bound at 0.88": that was the ONE-wave rate.  They run at half their issue roofline.)
        key = {"radial_step": f"k_radial_step<RadialGeo<{(d + 31) // 32}, {10 if nl <= 10 else 16}>, true>",
               "planar_step": f"k_planar_step<PlanarGeo<{(d + 31) // 32}, {6 if nl <= 10 else 8}>, true>"}.get(kname)
        sl = issue_slots(key) if key else None
        if sl:
            tiles_per_simd = (N + 31) // 32 / (256 * 4)
            for ghz in (2.4, 2.1):
                t_issue = tiles_per_simd * sl[0] * 2 / (ghz * 1e9)
                print(f"   {kname}: vector-issue floor at {ghz} GHz: ({sl[1]} + 4 x {sl[2]} + {sl[3]} matrix) slots per 32-sample tile x {tiles_per_simd:.0f} tiles per SIMD x 2 clk = "
                      f"{1e6 * t_issue:.1f} us -> the launch runs at {t_issue / (a.value * 1e-3):.2f} of its vector-issue roofline")
    for name, b in need.items():
        a, c = C.c_double(0.0), C.c_int64(0)
        lib.nf_prof_read(ctx.ptr, name.encode(), C.byref(a), C.byref(c))
        if not c.value:
            continue
        if c.value:
            print(f"   {name:13s} {1e3 * a.value:8.1f} us  {b / 1e6:8.1f} MB  -> {b / (a.value * 1e-3) / 1e12:.2f} TB/s ({b / (a.value * 1e-3) / 8e12:.0%} of 8 TB/s)")
    for name in (b"simple_finalize", b"finish_sum"):
        a, c = C.c_double(0.0), C.c_int64(0)
        lib.nf_prof_read(ctx.ptr, name, C.byref(a), C.byref(c))
        if c.value:
            print(f"   {name.decode():13s} {1e3 * a.value:8.1f} us")
    lib.nf_prof_enable(ctx.ptr, 0)
    # density evaluation / forward-KL side: logpdf(flow, ys) on the flow's own samples, then one forward-KL training step
    ys = nf.rand(flow, N, nf.PhiloxRNG(5))
    torch.cuda.synchronize()
    for name, fn in (("rand(flow, n)", lambda: nf.rand(flow, N, nf.PhiloxRNG(5))),
                     ("loglikelihood", lambda: nf.loglikelihood(None, flow, ys)),
                     ("forward-KL step", lambda: nf.loglikelihood_value_and_gradient(flow, ys))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        el2 = (time.perf_counter() - t0) / 5
        lib.nf_prof_enable(ctx.ptr, 2)
        fn()
        torch.cuda.synchronize()
        parts = []
        for kn in (b"base_sample", b"base_logpdf", b"simple_apply", b"target", b"sum2", b"simple_bwd", b"simple_finalize", b"finish_sum"):
            a, c = C.c_double(0.0), C.c_int64(0)
            lib.nf_prof_read(ctx.ptr, kn, C.byref(a), C.byref(c))
            if c.value:
                parts.append(f"{kn.decode()} {1e3 * a.value:.0f}us" + (f" x{c.value}" if c.value > 1 else ""))
        lib.nf_prof_enable(ctx.ptr, 0)
        print(f"   {name:16s} {1e3 * el2:8.3f} ms = {N / el2 / 1e6:7.1f} M samples/s   [{', '.join(parts)}]")
