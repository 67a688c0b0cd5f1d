#!/usr/bin/env python3
"""Benchmark of the reverse-KL (ELBO) training step on MI355X.

One "step" = what one iteration of the reference's `optimize` loop does around
`_value_and_gradient` (src/optimize.jl:85-99) for loss = -elbo_batch(rng, flow, logp, n):
  draw n base samples (Philox, in-kernel) + log q0, flow forward + sum log|det J|, target logp,
  mean, full reverse pass to grad theta, [all-reduce of [grad ; loss] when >1 GPU], Adam update
  and gradient norm.

Workload (BASELINE.json configs[1]): RealNVP, d = 64, 8 affine couplings (= realnvp(q0,
[64, 64], 4)), 2-hidden-layer conditioners of width 64, batch 65 536 PER GPU (weak scaling),
fp32, diag-Gaussian target, synthetic Glorot weights.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the coupling reverse
pass, fp32 MFMA bound); `cpu_baseline` is the reference's step under torch-CPU autograd on all host
cores at the SAME batch (oracle/nf_torch_cpu.py; the Julia reference cannot run on this box).

Timing: W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize: `value`, `ms_per_step` and
`roofline.frac` are those K steps -- the driver's protocol, nothing inserted.  On an MI355X that has been idle (even for 50 ms)
the shader clock DROPS from 2.4 GHz to 1.6-2.0 GHz when the load arrives and takes ~40 cfg-2 steps (25 ms) to come back
(tools/bench_ramp.py reads the clock next to every step: profiles/r4a_step_ramp_clocks.txt; step time x clock is constant), so
with `--steps 20 --warmup 5` the timed region rides that ramp.  What a training run of thousands of steps sees is reported
NEXT to it, never as `value`: after the timed region `--prewarm` more untimed steps and K more timed ones ->
`value_sustained_clock`, `ms_per_step_sustained_clock`, `roofline.frac_sustained_clock` (`--prewarm 0` skips that).

Multi-GPU: one process per GPU.  Under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`
the ranks come from the environment; a bare `python bench.py --gpus N` starts the N ranks itself (the parent
never touches a GPU, a failed rank fails the run).  `n_gpus` in the output is the world size the collective
actually ran on, and the run fails if that differs from --gpus.
"""
from __future__ import annotations

import argparse
import ctypes as C
import datetime
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

D, HDIMS, NLAYERS, BATCH = 64, (64, 64), 4, 65536
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16: 16 384 MACs per 32 clocks per CU-SIMD)
# How the dominant kernel's GEMMs are executed since round 4 (DESIGN section 4 "B6"): fp32 operands as exact bf16 triples,
# six bf16 x bf16 MFMA products per fp32 product.  `roofline.achieved / peak / frac` stay ALGORITHMIC fp32 flops against the
# fp32-MFMA peak (the figure rounds 1-3 reported, and the only fp32-grade matrix instruction the chip has); the executed bf16
# flops against the bf16 peak are reported beside it.  cfg 3's kernel still issues fp32 MFMAs.
# (the environment switches that select another kernel or form turn the label off: NF_STASH_SLIM forces the fp32 pair kernel too)
BF16X6 = {"cfg2": all(os.environ.get(k) is None for k in ("NF_BWD_FP32", "NF_BWD_NO_PAIR", "NF_BWD_DW_FP32", "NF_AFFINE_NO_STASH", "NF_STASH_SLIM")),
          "cfg4": os.environ.get("NF_WIDE_FP32") is None,
          # round 5: the NSF reverse kernel's three output-layer GEMMs (576 of its 664 MFMAs per tile group) are six-term bf16
          # products, layers 1-2 (88) stay fp32 MFMAs: the output layer is 11 776 of the net's 13 312 MACs per sample
          "cfg3": os.environ.get("NF_RQS_BWD_FP32") is None and os.environ.get("NF_RQS_BWD_PERWAVE") is None}
BF16_SHARE = {"cfg2": 1.0, "cfg4": 1.0, "cfg3": 11776.0 / 13312.0}  # share of the dominant kernel's algorithmic flops executed as bf16x6
# algorithmic flops of ONE coupling's reverse pass per sample (step 786 432 =
# fwd 262 144 + dX 262 144 + dW 262 144 over 8 couplings; recompute is not counted)
MACS_NET = 32 * 64 + 64 * 64 + 64 * 32
FLOPS_BWD_PER_SAMPLE_PER_COUPLING = 2 * (2 * MACS_NET) * 2  # 2 nets x (dX + dW) x 2 flop/MAC = 65 536
COUPLINGS_PER_LAUNCH = 2 * NLAYERS  # the reverse kernel walks all 8 couplings in one launch
FLOPS_STEP_PER_SAMPLE = 786432
WORKLOAD_TEXT = ("reverse-KL ELBO step: RealNVP d=64, 8 affine couplings, conditioner 32-64-64-32 "
                 "(hdims [64,64]), diag-Gaussian target, Philox base draws, Adam")
STASHED = not os.environ.get("NF_AFFINE_NO_STASH")  # the library's A/B switch back to the recompute kernel
PAIR = STASHED and not os.environ.get("NF_BWD_NO_PAIR")  # the library's A/B switch back to one wavefront per tile
DOMINANT = (b"affine_bwd", ("k_affine_bwd_pair" if PAIR else "k_affine_bwd_stashed") +
            " (reverse pass of all 8 couplings in one launch from the forward's activation stash: dX + dW, nothing recomputed" +
            ("; a producer and a consumer wavefront per tile)" if PAIR else ")") if STASHED else
            "k_affine_bwd_all (reverse pass of all 8 couplings in one launch: recompute + dX + dW)")
KERNEL_NAMES = (b"base_sample", b"pack_weights", b"affine_chain", b"target", b"affine_bwd", b"reduce_slabs", b"adam")


def select_cfg4(world: int):
    """BASELINE.json configs[3] (--workload cfg4; NOT the default bench line): RealNVP d=256, 16 couplings,
    hidden [256,256], 262 144 samples in total, sharded over the ranks (strong scaling).  Dominant kernel:
    k_wide_bwd, one launch = recompute + dX chain of ONE net (its dW GEMM is k_wide_dw)."""
    global D, HDIMS, NLAYERS, BATCH, MACS_NET, FLOPS_BWD_PER_SAMPLE_PER_COUPLING, FLOPS_STEP_PER_SAMPLE, COUPLINGS_PER_LAUNCH
    global WORKLOAD_TEXT, DOMINANT, KERNEL_NAMES
    D, HDIMS, NLAYERS, BATCH = 256, (256, 256), 8, 262144 // world
    MACS_NET = 128 * 256 + 256 * 256 + 256 * 128
    FLOPS_BWD_PER_SAMPLE_PER_COUPLING = 2 * MACS_NET  # per launch: dX of one net
    COUPLINGS_PER_LAUNCH = 1
    FLOPS_STEP_PER_SAMPLE = 16 * 2 * 3 * 2 * MACS_NET  # 16 couplings x 2 nets x (fwd + dX + dW)
    WORKLOAD_TEXT = ("reverse-KL ELBO step: RealNVP d=256, 16 affine couplings, conditioner 128-256-256-128 "
                     "(hdims [256,256]), diag-Gaussian target, Philox base draws, Adam; 262144 samples in total")
    DOMINANT = (b"wide_bwd", "k_wide_bwd_stashed_b6 / k_wide_bwd_stashed (reverse pass of one conditioner net from the forward's stash: "
                             "element-wise stage + dX chain on streamed weights; its dW GEMM is k_wide_dw)")
    KERNEL_NAMES = (b"base_sample", b"pack_weights", b"wide_apply", b"target", b"wide_bwd", b"wide_dw", b"reduce_slabs", b"adam")


def select_cfg3(world: int):
    """BASELINE.json configs[2] (--workload cfg3): NSF d=32, 8 RQ-spline couplings, K=8, B=5, hidden [32,32], batch
    131 072 per GPU.  Dominant kernel: the spline-coupling reverse pass (conditioner GEMM flops only:
    fwd 212 992 flop/sample over 8 couplings, step 638 976; the spline's VALU work is not counted)."""
    global D, HDIMS, NLAYERS, BATCH, MACS_NET, FLOPS_BWD_PER_SAMPLE_PER_COUPLING, FLOPS_STEP_PER_SAMPLE, COUPLINGS_PER_LAUNCH
    global WORKLOAD_TEXT, DOMINANT, KERNEL_NAMES, FLOW_KIND
    D, HDIMS, NLAYERS, BATCH = 32, (32, 32), 4, 131072
    FLOW_KIND = "nsf"
    MACS_NET = 16 * 32 + 32 * 32 + 32 * 368
    FLOPS_BWD_PER_SAMPLE_PER_COUPLING = 2 * 2 * MACS_NET  # (dX + dW) x 2 flop/MAC = 53 248
    COUPLINGS_PER_LAUNCH = None  # read from the library: launches per step of the dominant kernel
    FLOPS_STEP_PER_SAMPLE = 638976
    WORKLOAD_TEXT = ("reverse-KL ELBO step: NSF d=32, 8 rational-quadratic spline couplings (K=8, B=5), conditioner 16-32-32-368 "
                     "(hdims [32,32]), diag-Gaussian target, Philox base draws, Adam")
    DOMINANT = (b"rqs_bwd", "k_rqs_bwd_coop (reverse pass of one spline coupling per launch: conditioner recompute, spline reverse "
                            "pass at the forward's taped bins, dX + dW)")
    KERNEL_NAMES = (b"base_sample", b"pack_weights", b"rqs_chain", b"target", b"rqs_bwd", b"reduce_slabs", b"adam")


FLOW_KIND = "realnvp"


def pmc_traffic(kernel_substr: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 counter passes
    (profiles/*_pmc_summary.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this
    same command by tools/collect_profiles.sh; rocprofv3 reports both in KiB).  Correction
    (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE counts 64 B per 128-B request, i.e. half
    the bytes read.  Calibrated on this library's own access pattern (4 B/lane tile loads and stores):
    k_target_tiled reads 17.3 MB and reports 8 478 KiB = 8.68 MB (factor 2.0); k_base_sample_tiled
    writes 17.04 MB and reports WRITE_SIZE 16 640 KiB = 17.04 MB (factor 1.0) -- profiles/r1c_pmc_calibration.json.
    So traffic = 2 * FETCH_SIZE + WRITE_SIZE.  None if no profile is committed."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    for path in reversed(files):
        with open(path) as f:
            summ = json.load(f)
        for name, counters in summ.items():
            if kernel_substr in name and "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
                kib = 2.0 * counters["FETCH_SIZE"]["avg_per_launch"] + counters["WRITE_SIZE"]["avg_per_launch"]
                return kib * 1024.0, os.path.relpath(path, ROOT)
    return None, None


def cpu_baseline(seconds_budget: float = 24.0):
    """CPU restatements of the reference's training step at the FULL batch of the GPU workload, both timed on the box's
    usable host cpus (cgroup quota respected), the FASTER one reported:
      (i)  torch-CPU autograd of the reference's graph (MKL / oneDNN GEMMs) -- oracle/nf_torch_cpu.py;
      (ii) a fused C++ / OpenMP implementation (tile-resident activations, register-blocked small GEMMs, hand-derived
           reverse pass), compiled on this box with -march=native -- oracle/nf_cpu_step.cpp.
    Both are pinned against the oracle by tests/test_oracle.py.  The Julia reference cannot run here."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nf_torch_cpu as tc

    threads = int(os.environ.get("NF_CPU_THREADS", "0")) or tc.usable_cpus()
    a = tc.time_training_steps(D, HDIMS, NLAYERS, BATCH, seconds_budget=0.5 * seconds_budget)
    best = dict(a)
    best["candidates"] = {"torch_cpu_autograd": {"samples_per_s": a["value"], "ms_per_step": a["ms_per_step"]}}
    try:
        import nf_cpu_omp as co

        b = co.time_training_steps(D, HDIMS, NLAYERS, BATCH, threads, seconds_budget=0.5 * seconds_budget)
    except Exception as e:  # no compiler on the box, or the build failed: the torch number stands
        b = None
        best["candidates"]["cpp_openmp"] = f"unavailable: {e}"
    if b is not None:
        best["candidates"]["cpp_openmp"] = {"samples_per_s": b["value"], "ms_per_step": b["ms_per_step"]}
        if b["value"] > a["value"]:
            best.update(value=b["value"], ms_per_step=b["ms_per_step"],
                        sample=(f"{b['steps']} full training steps at batch {BATCH} (the GPU workload's batch, flow, target and dtype; median "
                                f"step {b['ms_per_step']:.1f} ms), fused C++/OpenMP implementation built on this box (-O3 -march=native), "
                                f"{threads} threads of {os.cpu_count()} host cpus; faster than torch-CPU autograd ({a['ms_per_step']:.1f} ms): "
                                "CPU restatement of the reference algorithm (the Julia reference cannot run on this box)"))
    return best


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU) and wait.  This parent
    process never initialises a GPU (no HIP call, no torch.cuda.is_available()); it only counts devices.  Returns the
    exit code for the whole run: non-zero as soon as any rank fails (the others are then terminated)."""
    one_device = os.environ.get("NF_BENCH_ONE_DEVICE") == "1"
    have = torch.cuda.device_count()
    if have < n and not one_device:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in list(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in alive:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


def run_once(args, world, world_observed, rank, dev, dist, damp, state):
    """One timed run of the selected workload; rank 0 prints one JSON line."""
    nf = load_package()
    lib = nf.load_library()
    n_local, n_global = args.batch, args.batch * world
    if FLOW_KIND == "nsf":
        flow = nf.nsf(nf.MvNormal(D), HDIMS, 8, 5.0, NLAYERS, paramtype=torch.float32, device=dev, seed=123)
    else:
        flow = nf.realnvp(nf.MvNormal(D), HDIMS, NLAYERS, paramtype=torch.float32, device=dev, seed=123)
    g0 = torch.Generator().manual_seed(123)
    mu = torch.randn(D, generator=g0).to(dev)
    var = (torch.rand(D, generator=g0) + 1e-3).to(dev)
    target = nf.DiagGaussTarget(mu, var)
    ctx = nf.context_for(dev)
    # the timed loop owns theta between steps, exactly like train_flow's (objectives._optimize_fused), and opts in the same way
    nf._lib.check(lib.nf_ctx_set_weight_cache(ctx.ptr, 1))
    P = flow.P
    theta = flow.theta.clone()
    m, v = torch.zeros_like(theta), torch.zeros_like(theta)
    out = torch.zeros(P + 1, dtype=torch.float32, device=dev)
    gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
    vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    desc, tgt = C.byref(flow.desc), C.byref(target.c)

    if damp is not None:
        theta.mul_(damp)
    lib_comm = state.get("lib_comm", False)
    if (dist is not None and args.collective == "nfhip" and os.environ.get("NF_BENCH_ONE_DEVICE") != "1" and not lib_comm
            and not state.get("lib_comm_failed")):
        # the library's own RCCL communicator: rank 0's unique id travels through the process group
        idbuf = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            raw = (C.c_char * 128)()
            nf._lib.check(lib.nf_comm_get_unique_id(raw))
            idbuf = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
        idbuf = idbuf.to(dev)
        dist.broadcast(idbuf, 0)
        raw = (C.c_char * 128).from_buffer_copy(idbuf.cpu().numpy().tobytes())
        # every rank must take the same route: if the library's communicator does not come up on ANY rank (it has never met
        # a multi-GPU node), all ranks keep the process group's RCCL all-reduce and the output says so
        ok = torch.ones(1, dtype=torch.int32, device=dev)
        # the init runs in a watchdog thread (ctypes releases the GIL): ncclCommInitRank is collective, so a rank that cannot
        # join would otherwise block every other rank inside it for ever.  A rank whose init has not returned after
        # NF_COMM_INIT_TIMEOUT seconds votes "failed" below and abandons the thread; the run then ends with os._exit.
        import threading

        res = {}

        def _init():
            try:
                nf._lib.check(lib.nf_comm_init_rank(ctx.ptr, raw, world, rank))
                res["ok"] = True
            except Exception as e:  # noqa: BLE001
                res["err"] = str(e)

        th = threading.Thread(target=_init, daemon=True)
        th.start()
        th.join(float(os.environ.get("NF_COMM_INIT_TIMEOUT", "120")))
        if th.is_alive():
            res["err"] = "nf_comm_init_rank did not return (timeout)"
            state["lib_comm_hung"] = True
        if "err" in res:
            ok.zero_()
            state["lib_comm_error"] = f"rank {rank}: {res['err']}"
            print(f"[bench] nf_comm_init_rank failed on rank {rank}: {res['err']}", file=sys.stderr, flush=True)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        lib_comm = bool(int(ok.item()))
        state["lib_comm"] = lib_comm
        if not lib_comm:
            state["lib_comm_failed"] = True
            if not state.get("lib_comm_hung") and int(lib.nf_comm_size(ctx.ptr)) > 1:
                lib.nf_comm_destroy(ctx.ptr)
    # (after a hung set-up the abandoned thread may still be writing the context's communicator: it is not read again -- ADVICE r4)
    comm_size = 1 if state.get("lib_comm_hung") else int(lib.nf_comm_size(ctx.ptr))
    if lib_comm and comm_size != world:
        raise SystemExit(f"library communicator has {comm_size} ranks, the process group {world}")
    # nf_elbo_step runs the whole iteration inside the library (for cfg 2 as three launches; with a communicator on the
    # context the all-reduce is the library's own).  The torch collective and --split-calls keep the separate calls.
    fused_step = not args.split_calls and (dist is None or lib_comm)
    use_graph = args.graph and dist is None and args.workload == "cfg2" and fused_step
    step_counter = torch.zeros(1, dtype=torch.int32, device=dev)
    stat_dev = torch.zeros(2, dtype=torch.float32, device=dev)

    stat_host = [C.c_double(0.0), C.c_double(0.0)]

    def step(i: int, read_stats: bool = False):
        if fused_step:
            # asynchronous unless the stat tuple (loss, norm(g)) is asked for -- the LAST timed step asks, which costs the
            # synchronisation the closing barrier performs anyway
            nf._lib.check(lib.nf_elbo_step(ctx.ptr, desc, tgt, vp(theta), vp(m), vp(v), n_local, 123, i, 1e-3, 0.9, 0.999, 1e-8,
                                           C.byref(stat_host[0]) if read_stats else None, C.byref(stat_host[1]) if read_stats else None))
            return
        nf._lib.check(lib.nf_elbo_value_and_grad(ctx.ptr, desc, tgt, vp(theta), None, n_local, n_global, 123,
                                                 rank * n_local, i, vp(out)))
        if lib_comm:
            nf._lib.check(lib.nf_allreduce_grad_loss(ctx.ptr, 0, vp(out), P + 1))
        elif dist is not None:
            dist.all_reduce(out)  # one RCCL all-reduce of [grad ; loss] (P + 1 floats)
        nf._lib.check(lib.nf_adam_update(ctx.ptr, 0, vp(theta), vp(out), vp(m), vp(v), P, 1e-3, 0.9, 0.999, 1e-8,
                                         i + 1, vp(gnorm)))

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if not args.no_kernel_events:
        # create the library's event pool NOW: doing it between the pre-warm and the timed region leaves the GPU idle for
        # milliseconds, long enough for its clocks to fall back
        nf._lib.check(lib.nf_prof_enable(ctx.ptr, 1))
        nf._lib.check(lib.nf_prof_enable(ctx.ptr, 0))
    # (created BEFORE the warm-up: whatever the host does between the warm-up and the timed region is GPU idle time, and the
    # shader clock falls back during idle time -- tools/bench_ramp.py)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(2 * (args.steps + 1))]
    graph = None
    if use_graph:
        # warm-up eagerly through the device-counter form (sizes the workspace, sets kernel attributes, packs the weights),
        # then capture ONE step and replay it: every launch argument is constant from step to step
        side = torch.cuda.Stream(dev)
        gctx = nf.Context(dev.index or 0, side.cuda_stream)

        def enqueue():
            nf._lib.check(lib.nf_elbo_step_enqueue(gctx.ptr, desc, tgt, vp(theta), vp(m), vp(v), n_local, 123, vp(step_counter),
                                                   1e-3, 0.9, 0.999, 1e-8, vp(stat_dev)))

        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for i in range(args.warmup):
                enqueue()
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            enqueue()
        _eager_step = step

        def step(i: int, read_stats: bool = False):  # noqa: F811
            graph.replay()
    else:
        for i in range(args.warmup):
            step(i)
    barrier()
    first = args.warmup  # index of the first timed step

    def timed_region(first_step: int, marks_off: int):
        """EXACTLY K steps between barrier + synchronize (max over ranks), with the dominant kernel bracketed by HIP events on
        the launch stream (every 4th launch: an event pair between two kernels of one stream drains the queue, ~30 us per
        bracketed cfg-2 step, measured A/B with --no-kernel-events).  Returns (elapsed s, sorted per-step ms, kernel avg ms, n)."""
        if not args.no_kernel_events and not use_graph:
            nf._lib.check(lib.nf_prof_enable(ctx.ptr, 1 if os.environ.get("NF_BENCH_EVENTS_EVERY_STEP") else 3))
        barrier()
        t0 = time.perf_counter()
        marks[marks_off].record()
        for k, i in enumerate(range(first_step, first_step + args.steps)):
            step(i, read_stats=(k == args.steps - 1))
            marks[marks_off + k + 1].record()  # same stream as the library's launches: per-step device time for the median
        barrier()
        el = time.perf_counter() - t0
        per_step = sorted(marks[marks_off + k].elapsed_time(marks[marks_off + k + 1]) for k in range(args.steps))
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t[0])
        a_ms, c_n = C.c_double(0.0), C.c_int64(0)
        nf._lib.check(lib.nf_prof_read(ctx.ptr, DOMINANT[0], C.byref(a_ms), C.byref(c_n)))
        return el, per_step, a_ms.value, c_n.value

    elapsed, step_ms, dom_ms, dom_cnt = timed_region(first, 0)
    first += args.steps
    sustained = None
    if args.prewarm > 0:
        for i in range(args.prewarm):
            step(first + i)
        first += args.prewarm
        sustained = timed_region(first, args.steps + 1)
        first += args.steps
    avg_ms, cnt = C.c_double(dom_ms), C.c_int64(dom_cnt)
    if use_graph:
        loss, gnorm_v = float(stat_dev[0]), float(stat_dev[1])
        step = _eager_step  # the per-kernel breakdown below runs eagerly
        lib.nf_ctx_weights_changed(ctx.ptr)
    elif fused_step:
        loss, gnorm_v = stat_host[0].value, stat_host[1].value  # the last timed step's stat tuple
    else:
        loss, gnorm_v = float(out[P]), float(gnorm)
    assert np.isfinite(loss) and np.isfinite(gnorm_v), "non-finite loss / gradient norm"

    # per-kernel breakdown: a few extra, UNTIMED steps with every kernel bracketed
    kernel_ms = {}
    nbreak = 5
    nf._lib.check(lib.nf_prof_enable(ctx.ptr, 2))
    for i in range(nbreak):
        step(first + i)
    torch.cuda.synchronize(dev)
    for name in KERNEL_NAMES:
        a, c = C.c_double(0.0), C.c_int64(0)
        lib.nf_prof_read(ctx.ptr, name, C.byref(a), C.byref(c))
        if c.value:  # kernels fused away in this configuration are not listed
            kernel_ms[name.decode()] = {"avg_ms": round(a.value, 5), "launches_per_step": c.value / nbreak}
    nf._lib.check(lib.nf_prof_enable(ctx.ptr, 0))
    events_in_timed_region = cnt.value > 0
    if not events_in_timed_region and DOMINANT[0].decode() in kernel_ms:
        # graph replay (no event brackets inside a captured graph) or --no-kernel-events: the dominant kernel's duration
        # comes from the bracketed, untimed steps after the timed region
        avg_ms = C.c_double(kernel_ms[DOMINANT[0].decode()]["avg_ms"])
        cnt = C.c_int64(int(kernel_ms[DOMINANT[0].decode()]["launches_per_step"] * nbreak))

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = n_global * args.steps / elapsed
        couplings_per_launch = COUPLINGS_PER_LAUNCH
        if couplings_per_launch is None:  # cfg3: 8 couplings over however many launches the library used per step
            per_step = kernel_ms.get(DOMINANT[0].decode(), {}).get("launches_per_step", 8.0)
            couplings_per_launch = 2 * NLAYERS / max(per_step, 1.0)
        flop_per_launch = FLOPS_BWD_PER_SAMPLE_PER_COUPLING * couplings_per_launch * n_local
        achieved = flop_per_launch / (avg_ms.value * 1e-3) / 1e12 if avg_ms.value > 0 else 0.0
        traffic, traffic_src = (None, None)
        if n_local == BATCH and args.workload in ("cfg2", "cfg3"):
            traffic, traffic_src = pmc_traffic(("k_affine_bwd_pair" if PAIR else "k_affine_bwd_stashed" if STASHED else "k_affine_bwd_all") if args.workload == "cfg2" else "k_rqs_bwd")
        rec = {
            "metric": "elbo_samples_per_sec",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world_observed,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_step_median": step_ms[len(step_ms) // 2],
            "value_sustained_clock": None if sustained is None else n_global * args.steps / sustained[0],
            "ms_per_step_sustained_clock": None if sustained is None else 1e3 * sustained[0] / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if args.workload == "cfg4" else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": WORKLOAD_TEXT,
                "batch_per_gpu": n_local,
                "global_batch": n_global,
                "params": P,
                "parallelism": f"dp{world_observed} (sample-sharded, one all-reduce of P+1 floats per step"
                               + (", issued by libnfhip's RCCL communicator)" if lib_comm
                                  else ", torch.distributed RCCL; libnfhip's communicator failed to initialise)" if state.get("lib_comm_failed")
                                  else ", torch.distributed RCCL)" if dist is not None else ")"),
                "nf_comm_size": comm_size,
                "lib_comm_hung": bool(state.get("lib_comm_hung")),  # nf_comm_init_rank did not return within NF_COMM_INIT_TIMEOUT: the run fell back to torch's collective and EXITS NON-ZERO
                # messages the step's one logical all-reduce of [grad ; loss] travels as (nf_comm_bucket_count: buckets of whole
                # couplings on a second stream for cfg 4's 16.9 MB, one message for cfg 2's 0.5 MB; 0 = no communicator)
                "all_reduce_messages_per_step": int(lib.nf_comm_bucket_count(ctx.ptr, desc)) if lib_comm else (1 if dist is not None else 0),
                "step_form": ("hipGraph replay of nf_elbo_step_enqueue" if use_graph else "nf_elbo_step (whole iteration inside the library)"
                              if fused_step else "nf_elbo_value_and_grad + nf_adam_update (split calls)"),
                "init": "Glorot-uniform weights, zero biases (Flux default)" + (f", theta scaled by {damp}" if damp is not None else ""),
                "final_loss": loss,
                "final_gradient_norm": gnorm_v,
                "timing": (f"value / ms_per_step / roofline.frac: the {args.steps} steps straight after the {args.warmup} warm-ups "
                           "(the command's protocol; after any idle an MI355X drops its shader clock to 1.6-2.0 GHz when load arrives "
                           "and needs ~40 cfg-2 steps to return to 2.4 GHz: profiles/r4a_step_ramp_clocks.txt).  "
                           + ("" if sustained is None else
                              f"*_sustained_clock: {args.steps} more steps timed the same way after {args.prewarm} further untimed "
                              "steps, i.e. what step 200 of a training run costs")),
                "untimed_steps_before_value": args.warmup,
                "untimed_steps_before_value_sustained_clock": None if sustained is None else args.warmup + args.steps + args.prewarm,
                "weight_cache": "nf_ctx_set_weight_cache(ctx, 1), as train_flow's loop sets it" if fused_step else "not used",
            },
            "roofline": {
                "kernel": DOMINANT[1],
                "bound": "mfma",
                "achieved": achieved,
                "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                "frac_basis": ("ALGORITHMIC fp32 flops per launch / launch time / the fp32-MFMA peak (SURVEY 8(d)); where the GEMMs execute "
                               "as bf16x6 this can exceed 1 -- the fp32 instruction is then not the pipe the kernel runs on; see "
                               "executed_frac_of_bf16_peak for the executed form"),
                "mfma_form": (("bf16x6: every fp32 product as six bf16 MFMA products of exactly split operands, split by rounding to nearest "
                               "(fp32-grade: tools/probe/split_bias_probe.hip, tools/parity_ab.py)"
                               + ("" if BF16_SHARE[args.workload] == 1.0 else f"; {BF16_SHARE[args.workload]:.3f} of the kernel's algorithmic flops, the rest fp32 MFMAs"))
                              if BF16X6[args.workload] else "fp32 MFMA (v_mfma_f32_32x32x2_f32)"),
                "executed_bf16_tflops": 6.0 * BF16_SHARE[args.workload] * achieved if BF16X6[args.workload] else None,
                "executed_frac_of_bf16_peak": 6.0 * BF16_SHARE[args.workload] * achieved / PEAK_BF16_MFMA_TFLOPS if BF16X6[args.workload] else None,
                "traffic": traffic,
                "traffic_unit": "bytes per launch (HBM: 2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC, gfx950 correction)",
                "traffic_source": traffic_src,
                # the implementation's HBM bytes over the launch time: what the kernel actually pulls (the activation stash), next to
                # what a plain copy / read / write loop reaches on an MI355X of this pool (tools/probe/peaks_probe.hip)
                "traffic_rate_TBps": None if not traffic or avg_ms.value <= 0 else traffic / (avg_ms.value * 1e-3) / 1e12,
                "reference_hbm_TBps": {"copy": 4.66, "read": 6.29, "write": 3.91, "source": "profiles/r4w_peaks_probe.txt",
                                       "note": "constants measured once on an MI355X of this pool (tools/probe/peaks_probe.hip), NOT in this run"},
                # avg_launch_ms: the dominant kernel's launches INSIDE the timed region (the command's K steps, on the clock ramp when
                # the GPU was idle a moment ago); avg_launch_ms_sustained_clock: the same kernel in the sustained-clock measurement
                # afterwards -- that one, and the `kernels` table below, are what profiles/*_kernel_stats_*.csv reproduce
                "avg_launch_ms": avg_ms.value,
                "launches_timed": cnt.value,
                "timed_inside_the_timed_region": events_in_timed_region,
                "algorithmic_flop_per_launch": flop_per_launch,
                "whole_step_tflops": FLOPS_STEP_PER_SAMPLE * n_local / (ms_per_step * 1e-3) / 1e12,
                "frac_sustained_clock": None if sustained is None or sustained[2] <= 0 else
                                        flop_per_launch / (sustained[2] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "avg_launch_ms_sustained_clock": None if sustained is None else sustained[2],
            },
            "kernels": kernel_ms,
            "kernels_note": "per-kernel HIP-event averages of 5 extra steps AFTER the timed region and the sustained-clock measurement (post-run, sustained clock) -- compare with roofline.avg_launch_ms_sustained_clock, not with roofline.avg_launch_ms",
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "cfg2":
            rec["cpu_baseline"] = cpu_baseline()
            rec["gpu_over_cpu"] = value / rec["cpu_baseline"]["value"]
        print(json.dumps(rec))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--prewarm", type=int, default=None,
                    help="untimed steps AFTER the timed K, followed by K more timed steps reported as *_sustained_clock (the GPU's "
                         "shader clock needs ~40 cfg-2 steps under load to return to 2.4 GHz, profiles/r4a_step_ramp_clocks.txt); "
                         "default per workload (cfg2 120, cfg3 60, cfg4 4), 0 = no sustained-clock figure.  Never affects `value`")
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU per step")
    ap.add_argument("--workload", choices=("cfg2", "cfg3", "cfg4"), default="cfg2",
                    help="cfg2 = the headline line (default); cfg3 = NSF d=32 K=8, 131072 per GPU; "
                         "cfg4 = d=256 / 16 couplings / h=256, 262144 samples sharded (strong scaling)")
    ap.add_argument("--collective", choices=("torch", "nfhip"), default="nfhip",
                    help="who issues the one all-reduce per step: the library's own RCCL communicator (nf_allreduce_grad_loss on "
                         "the context stream, inside nf_elbo_step; default) or torch.distributed's RCCL process group")
    ap.add_argument("--split-calls", action="store_true",
                    help="time nf_elbo_value_and_grad + nf_adam_update (six launches per cfg-2 step) instead of nf_elbo_step "
                         "(three: fused forward, reverse pass, fused epilogue)")
    ap.add_argument("--graph", action="store_true",
                    help="1 GPU, cfg2: capture nf_elbo_step_enqueue into a hipGraph once and time K replays")
    ap.add_argument("--damp", type=float, default=None,
                    help="scale the Glorot-initialised theta (cfg4: --damp 0.5 is the contraction the parity tests judge; the "
                         "default is the Flux initialisation)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket kernels with HIP events in the timed region (roofline object is then empty)")
    args = ap.parse_args()
    if args.prewarm is None:
        args.prewarm = {"cfg2": 120, "cfg3": 60, "cfg4": 4}[args.workload]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))  # the parent only supervises; ranks are fresh processes
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # NF_BENCH_ONE_DEVICE=1 (logic check of the multi-rank path on a 1-GPU box): every rank on cuda:0, gloo
        # collectives -- RCCL refuses two ranks on one device.  Never set by the driver; numbers are meaningless.
        one_device = os.environ.get("NF_BENCH_ONE_DEVICE") == "1"
        if one_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if one_device:
            # gloo picks its interface by resolving the host name, which a container's may not do (or only after a long
            # stall: one 600 s hang of this path on a reused test box in round 5); the ranks are on this host by construction
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device("cuda", local_rank)

    if dist is not None and dist.get_world_size() != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
    world_observed = dist.get_world_size() if dist is not None else 1
    if args.workload == "cfg4":
        select_cfg4(world)
    elif args.workload == "cfg3":
        select_cfg3(world)
    if args.batch is None:
        args.batch = BATCH
    state = {}
    run_once(args, world, world_observed, rank, dev, dist, args.damp, state)  # ONE JSON line per invocation
    if dist is not None:
        dist.barrier()
        if state.get("lib_comm_hung"):  # a thread of this process is still inside RCCL's init: do not wait for it at exit --
            sys.stdout.flush()          # and do not report success: the line above carries lib_comm_hung, the exit code says so too
            sys.stderr.flush()
            os._exit(3)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
