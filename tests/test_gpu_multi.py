"""GPU-side checks of the multi-GPU plumbing that a 1-GPU box can run: bench.py's own rank launcher (two ranks on
one device, gloo collectives -- RCCL refuses two ranks on a device) and the library's RCCL entry points at world
size 1.  The 2-rank arithmetic (shards, one all-reduce, identical replicas) is covered on CPU by
tests/test_multigpu_cpu.py; real N-GPU RCCL runs are the driver's."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from __graft_entry__ import ROOT, load_package

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _bench(args, env_extra=None):
    env = dict(os.environ, **(env_extra or {}))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher and no environment must start 2 ranks, report the world size the
    collective actually ran on, and shard the same global batch a single rank would draw (Philox counters are global
    sample indices): the 2-rank loss equals the 1-rank loss at the same global batch."""
    common = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-events"]
    two = _bench(["--gpus", "2", "--batch", "2048"] + common, {"NF_BENCH_ONE_DEVICE": "1"})
    assert two["n_gpus"] == 2 and two["config"]["global_batch"] == 4096 and two["config"]["batch_per_gpu"] == 2048
    one = _bench(["--gpus", "1", "--batch", "4096"] + common)
    assert one["n_gpus"] == 1 and one["config"]["global_batch"] == 4096
    assert two["config"]["final_loss"] == pytest.approx(one["config"]["final_loss"], rel=1e-5)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs on the node (the pool's test boxes have one)")
def test_two_real_gpus_run_the_librarys_own_rccl_all_reduce():
    """On a node with >= 2 GPUs: `bench.py --gpus 2` must bring up libnfhip's communicator over RCCL (not fall back to the
    process group's all-reduce), run the whole iteration inside nf_elbo_step, and reproduce the 1-rank loss at the same
    global batch (the shards' draws are the same global Philox columns; the all-reduce only changes the summation order)."""
    common = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-events"]
    two = _bench(["--gpus", "2", "--batch", "2048"] + common)
    assert two["n_gpus"] == 2 and two["config"]["nf_comm_size"] == 2
    assert "libnfhip's RCCL communicator)" in two["config"]["parallelism"], two["config"]["parallelism"]
    assert two["config"]["step_form"].startswith("nf_elbo_step")
    one = _bench(["--gpus", "1", "--batch", "4096"] + common)
    assert two["config"]["final_loss"] == pytest.approx(one["config"]["final_loss"], rel=1e-5)


def test_bench_refuses_a_world_size_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_library_rccl_entry_points_world_size_one():
    """nf_comm_get_unique_id / nf_comm_init_rank / nf_allreduce_grad_loss / nf_comm_destroy over librccl (bound at run
    time) with one rank: the in-place sum all-reduce of [grad ; loss] on the context stream leaves the buffer as is."""
    nf = load_package()
    lib = nf.load_library()
    dev = torch.device("cuda", 0)
    ctx = nf._lib.Context(0, torch.cuda.current_stream(0).cuda_stream)  # a private context: the shared one stays comm-free
    try:
        assert lib.nf_comm_size(ctx.ptr) == 1
        raw = (C.c_char * 128)()
        nf._lib.check(lib.nf_comm_get_unique_id(raw))
        assert any(b != 0 for b in raw.raw)
        nf._lib.check(lib.nf_comm_init_rank(ctx.ptr, raw, 1, 0))
        assert lib.nf_comm_size(ctx.ptr) == 1
        assert lib.nf_comm_init_rank(ctx.ptr, raw, 1, 0) == -1  # a context holds one communicator
        for dt, code in ((torch.float32, 0), (torch.float64, 1)):
            buf = torch.randn(133633, dtype=dt, device=dev)  # P + 1 of the headline flow
            ref = buf.clone()
            nf._lib.check(lib.nf_allreduce_grad_loss(ctx.ptr, code, C.c_void_p(buf.data_ptr()), buf.numel()))
            torch.cuda.synchronize()
            assert torch.equal(buf, ref)
        nf._lib.check(lib.nf_comm_destroy(ctx.ptr))
        assert lib.nf_allreduce_grad_loss(ctx.ptr, 0, C.c_void_p(buf.data_ptr()), 4) == -1  # no communicator any more
        # the single-process form (one host thread, G contexts): G = 1 here
        arr = (C.c_void_p * 1)(ctx.ptr)
        nf._lib.check(lib.nf_comm_init_all(arr, 1))
        bufs = (C.c_void_p * 1)(buf.data_ptr())
        nf._lib.check(lib.nf_allreduce_grad_loss_all(arr, 1, 1, bufs, buf.numel()))
        torch.cuda.synchronize()
        assert torch.equal(buf, ref)
    finally:
        ctx.close()


def test_bucketed_all_reduce_equals_one_message_under_a_communicator():
    """nf_elbo_step under a communicator, weight-streaming RealNVP (the cfg-4 family): the gradient leaves in buckets of
    whole couplings on the context's second stream (nf_comm_bucket_count > 1) and the optimiser update waits for the
    join.  With one rank every all-reduce is the identity, so theta / m / v after three steps must equal, bit for bit,
    both the single-message form (bucket bytes 0) and a context without a communicator -- what the test pins is the
    schedule: bucket boundaries on coupling boundaries, the loss in the last bucket, events and the join in order."""
    nf = load_package()
    lib = nf.load_library()
    flow = nf.realnvp(nf.MvNormal(256), [256, 256], 3, paramtype=torch.float32, seed=4)  # 6 couplings, P = 1.58 M (6.3 MB)
    rng = np.random.default_rng(0)
    tgt = nf.DiagGaussTarget(torch.tensor(rng.standard_normal(256), dtype=torch.float32, device="cuda"),
                             torch.tensor(rng.uniform(size=256) + 0.5, dtype=torch.float32, device="cuda"))
    vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    n = 1000
    stream = torch.cuda.current_stream(0).cuda_stream

    def run(comm, bucket_bytes):
        ctx = nf._lib.Context(0, stream)
        try:
            if comm:
                raw = (C.c_char * 128)()
                nf._lib.check(lib.nf_comm_get_unique_id(raw))
                nf._lib.check(lib.nf_comm_init_rank(ctx.ptr, raw, 1, 0))
                nf._lib.check(lib.nf_ctx_set_comm_bucket_bytes(ctx.ptr, bucket_bytes))
            nb = lib.nf_comm_bucket_count(ctx.ptr, C.byref(flow.desc))
            th, m, v = flow.theta.clone(), torch.zeros_like(flow.theta), torch.zeros_like(flow.theta)
            stats = []
            for step in range(3):
                loss, gn = C.c_double(0), C.c_double(0)
                nf._lib.check(lib.nf_elbo_step(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), vp(th), vp(m), vp(v), n, 7, step, 1e-3, 0.9,
                                               0.999, 1e-8, C.byref(loss), C.byref(gn)))
                stats.append((loss.value, gn.value))
            torch.cuda.synchronize()
            return nb, th, m, v, stats
        finally:
            ctx.close()

    nb0, th0, m0, v0, st0 = run(False, 0)
    nb1, th1, m1, v1, st1 = run(True, 0)            # one message
    nb2, th2, m2, v2, st2 = run(True, 2 << 20)      # 2 MiB buckets: two couplings (2.1 MB) each -> 3 buckets
    nb3, th3, m3, v3, st3 = run(True, 1 << 20)      # 1 MiB: one coupling per bucket -> 6
    nb4, _, _, _, _ = run(True, -1)                 # automatic (4 MiB): 6.3 MB < two buckets -> one message
    assert (nb0, nb1, nb2, nb3, nb4) == (0, 1, 3, 6, 1)
    for th, m, v, st in ((th1, m1, v1, st1), (th2, m2, v2, st2), (th3, m3, v3, st3)):
        assert torch.equal(th, th0) and torch.equal(m, m0) and torch.equal(v, v0) and st == st0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs on the node (the pool's test boxes have one)")
def test_one_process_two_devices_all_reduce():
    """The single-process form of the collective (one host thread, G contexts: nf_comm_init_all +
    nf_allreduce_grad_loss_all) over two real devices: each context computes its shard of one global batch, the grouped
    all-reduce leaves the same [grad ; loss] on both, and it equals the one-device gradient of the whole batch."""
    nf = load_package()
    lib = nf.load_library()
    G = 2
    flows, tgts, ctxs, outs = [], [], [], []
    rng = np.random.default_rng(0)
    mu, var = rng.standard_normal(64).astype(np.float32), (rng.uniform(size=64) + 0.5).astype(np.float32)
    n = 2048
    for g in range(G):
        dev = torch.device("cuda", g)
        with torch.cuda.device(dev):
            flows.append(nf.realnvp(nf.MvNormal(64), [64, 64], 2, paramtype=torch.float32, device=dev, seed=3))
            tgts.append(nf.DiagGaussTarget(torch.tensor(mu, device=dev), torch.tensor(var, device=dev)))
            ctxs.append(nf._lib.Context(g, torch.cuda.current_stream(g).cuda_stream))
            outs.append(torch.zeros(flows[g].P + 1, dtype=torch.float32, device=dev))
    try:
        arr = (C.c_void_p * G)(*[c.ptr for c in ctxs])
        nf._lib.check(lib.nf_comm_init_all(arr, G))
        assert all(lib.nf_comm_size(c.ptr) == G for c in ctxs)
        for g in range(G):
            nf._lib.check(lib.nf_elbo_value_and_grad(ctxs[g].ptr, C.byref(flows[g].desc), C.byref(tgts[g].c), C.c_void_p(flows[g].theta.data_ptr()),
                                                     None, n, G * n, 11, g * n, 0, C.c_void_p(outs[g].data_ptr())))
        bufs = (C.c_void_p * G)(*[o.data_ptr() for o in outs])
        nf._lib.check(lib.nf_allreduce_grad_loss_all(arr, G, 0, bufs, outs[0].numel()))
        for g in range(G):
            torch.cuda.synchronize(g)
        assert torch.equal(outs[0].cpu(), outs[1].cpu())
        ref = torch.zeros_like(outs[0])
        ctx1 = nf._lib.Context(0, torch.cuda.current_stream(0).cuda_stream)
        nf._lib.check(lib.nf_elbo_value_and_grad(ctx1.ptr, C.byref(flows[0].desc), C.byref(tgts[0].c), C.c_void_p(flows[0].theta.data_ptr()),
                                                 None, G * n, G * n, 11, 0, 0, C.c_void_p(ref.data_ptr())))
        torch.cuda.synchronize(0)
        ctx1.close()
        scale = float(ref[:-1].abs().max())
        assert float((outs[0] - ref)[:-1].abs().max()) <= 1e-5 * scale and float(outs[0][-1]) == pytest.approx(float(ref[-1]), rel=1e-5)
    finally:
        for c in ctxs:
            c.close()


ARENA_FLOWS = {
    "realnvp_resident": lambda nf: nf.realnvp(nf.MvNormal(8), [32, 32], 2, paramtype=torch.float32, seed=1),
    "realnvp_wide": lambda nf: nf.realnvp(nf.MvNormal(100), [96, 130], 1, paramtype=torch.float32, seed=2),
    "realnvp_f64": lambda nf: nf.realnvp(nf.MvNormal(5), [16], 1, paramtype=torch.float64, seed=3),
    "nsf_mfma": lambda nf: nf.nsf(nf.MvNormal(5), [32, 32], 10, 5.0, 2, paramtype=torch.float32, seed=4),
    "nsf_general": lambda nf: nf.nsf(nf.MvNormal(6), [24, 16, 8], 8, 5.0, 1, paramtype=torch.float32, seed=5),
    "planar": lambda nf: nf.planarflow(nf.MvNormal(5), 10, paramtype=torch.float32, seed=6),
    "planar_many_layers": lambda nf: nf.planarflow(nf.MvNormal(40), 20, paramtype=torch.float32, seed=6),  # beyond k_simple_step
    "radial_f64": lambda nf: nf.radialflow(nf.MvNormal(7), 4, paramtype=torch.float64, seed=7),
    "meanfield": lambda nf: nf.meanfield(nf.MvNormal(4), paramtype=torch.float32),
    "realnvp_three_hidden": lambda nf: nf.realnvp(nf.MvNormal(10), [40, 33, 17], 2, paramtype=torch.float32, seed=8),  # MFMA layer kernels (l64)
    "nsf_hidden64": lambda nf: nf.nsf(nf.MvNormal(12), [64, 64], 8, 3.0, 1, paramtype=torch.float32, seed=9),
    "composite_nsf_wide_general_segments": lambda nf: nf.create_flow(
        [nf.nsf(nf.MvNormal(8), [32, 32], 8, 4.0, 1, paramtype=torch.float32, seed=1), nf.realnvp(nf.MvNormal(8), [96, 70], 1, paramtype=torch.float32, seed=2),
         nf.nsf(nf.MvNormal(8), [24, 16, 8], 6, 4.0, 1, paramtype=torch.float32, seed=3), nf.realnvp(nf.MvNormal(8), [20], 1, paramtype=torch.float32, seed=4)],
        nf.MvNormal(torch.randn(8, device="cuda"), torch.rand(8, device="cuda") + 0.5)),
    "composite_f64_general_base": lambda nf: nf.create_flow(
        [nf.realnvp(nf.MvNormal(6), [16, 16], 1, paramtype=torch.float64, seed=2), nf.nsf(nf.MvNormal(6), [12, 12], 5, 4.0, 1, paramtype=torch.float64, seed=3),
         nf.radialflow(nf.MvNormal(6), 2, paramtype=torch.float64, seed=1)],
        nf.MvNormal(torch.randn(6, dtype=torch.float64, device="cuda"), torch.rand(6, dtype=torch.float64, device="cuda") + 0.5)),
    "composite_general_base": lambda nf: nf.create_flow(
        [nf.radialflow(nf.MvNormal(6), 2, paramtype=torch.float32, seed=1), nf.realnvp(nf.MvNormal(6), [16, 16], 1, paramtype=torch.float32, seed=2),
         nf.planarflow(nf.MvNormal(6), 2, paramtype=torch.float32, seed=3)],
        nf.MvNormal(torch.randn(6, device="cuda"), torch.rand(6, device="cuda") + 0.5)),
}


@pytest.mark.parametrize("name", list(ARENA_FLOWS))
def test_caller_provided_arena_covers_every_entry_point(name):
    """nf_workspace_bytes + nf_ctx_set_arena ("allocates nothing persistent except the ctx; arena sized by
    nf_workspace_bytes"): with the arena set, every compute entry point runs inside it -- same results as the owned
    mode -- and an arena that is too small is refused with NF_ERR_WORKSPACE instead of allocating."""
    nf = load_package()
    lib = nf.load_library()
    flow = ARENA_FLOWS[name](nf)
    if name.startswith("planar") or name.startswith("radial") or name == "composite_general_base":
        flow = flow.with_theta(flow.theta * 0.3)
    dt, d, n = flow.theta.dtype, flow.dist.d, 333
    ctx = flow.ctx
    tgt = nf.DiagGaussTarget(torch.randn(d, dtype=dt, device="cuda"), torch.rand(d, dtype=dt, device="cuda") + 0.5)

    def run_everything():
        out = {}
        xs = nf.device_specific_rand(nf.PhiloxRNG(3), flow.dist, n, dtype=dt, device="cuda")
        ys, ladj = nf.with_logabsdet_jacobian(flow.transform, xs)
        xr, lb = nf.with_logabsdet_jacobian(nf.inverse(flow.transform), ys)
        y1, l1 = nf.with_logabsdet_jacobian(nf.layer(flow, 0), xs)
        out["fwd"], out["inv"], out["layer"] = ys, xr, y1
        out["rand"] = nf.rand(flow, n, nf.PhiloxRNG(4))
        out["elbo_xs"] = torch.tensor(nf.elbo_batch(flow, tgt, xs))
        out["elbo_rng"] = torch.tensor(nf.elbo_batch(nf.PhiloxRNG(5), flow, tgt, n))
        out["elbos"] = nf.batched_elbos(flow, tgt, xs)
        out["loglik"] = torch.tensor(nf.loglikelihood(None, flow, ys))
        l, g = nf.value_and_gradient(nf.elbo_batch, flow, tgt, n, rng=nf.PhiloxRNG(6))
        out["step_rng"] = torch.cat([g, torch.tensor([l], dtype=dt, device="cuda")])
        l, g = nf.value_and_gradient(nf.elbo_batch, flow, tgt, xs)
        out["step_xs"] = torch.cat([g, torch.tensor([l], dtype=dt, device="cuda")])
        l, g = nf.value_and_gradient(nf.elbo_batch, flow, lambda y: -(y * y).sum(0), xs)  # generic closure: nf_flow_bwd
        out["pullback"] = g
        l, g = nf.loglikelihood_value_and_gradient(flow, ys)
        out["fkl"] = torch.cat([g, torch.tensor([l], dtype=dt, device="cuda")])
        th = flow.theta.clone()
        st = nf.setup(nf.Adam(1e-3), th)
        out["gnorm"] = nf.adam_update(nf.Adam(1e-3), st, th, g).clone()
        out["adam"] = th
        th2, m2, v2 = flow.theta.clone(), torch.zeros_like(flow.theta), torch.zeros_like(flow.theta)
        loss, gn = C.c_double(0), C.c_double(0)
        nf._lib.check(lib.nf_elbo_step(ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), C.c_void_p(th2.data_ptr()), C.c_void_p(m2.data_ptr()),
                                       C.c_void_p(v2.data_ptr()), n, 9, 0, 1e-3, 0.9, 0.999, 1e-8, C.byref(loss), C.byref(gn)))
        out["elbo_step"] = th2
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in out.items()}

    ref = run_everything()
    need = int(lib.nf_workspace_bytes(ctx.ptr, C.byref(flow.desc), n))
    assert need > 0
    arena = torch.empty(need + 256, dtype=torch.uint8, device="cuda")
    base = (arena.data_ptr() + 255) // 256 * 256
    try:
        nf._lib.check(lib.nf_ctx_set_arena(ctx.ptr, C.c_void_p(base), need))
        got = run_everything()
        # every path sums parameter gradients in a fixed order (round 3: the general one-thread-per-sample kernels too,
        # wave sums into workgroup slabs): the arena run must reproduce the owned-memory run bit for bit

        def same(a, b):
            return torch.equal(a, b)

        for k in ref:
            assert same(ref[k], got[k]), (name, k)
        # an arena that is too small is refused, never silently replaced by an allocation
        nf._lib.check(lib.nf_ctx_set_arena(ctx.ptr, C.c_void_p(base), 4096))
        with pytest.raises(nf.NFHipError, match="arena is too small"):
            nf.value_and_gradient(nf.elbo_batch, flow, tgt, 4096, rng=nf.PhiloxRNG(6))
    finally:
        nf._lib.check(lib.nf_ctx_set_arena(ctx.ptr, None, 0))  # back to the owned, grow-only mode
    again = run_everything()
    for k in ref:
        assert same(ref[k], again[k]), (name, k, "after leaving arena mode")


def test_elbo_step_reports_non_finite_loss():
    """NF_ERR_NONFINITE: the reference's tests require finite ELBOs (test/flow.jl:58-60); the fused step says so."""
    nf = load_package()
    lib = nf.load_library()
    flow = nf.realnvp(nf.MvNormal(6), [16, 16], 1, paramtype=torch.float32, seed=1)
    tgt = nf.DiagGaussTarget(torch.zeros(6, device="cuda"), torch.ones(6, device="cuda"))
    th = flow.theta.clone()
    th[3] = float("nan")
    m, v = torch.zeros_like(th), torch.zeros_like(th)
    loss, gn = C.c_double(0), C.c_double(0)
    code = lib.nf_elbo_step(flow.ctx.ptr, C.byref(flow.desc), C.byref(tgt.c), C.c_void_p(th.data_ptr()), C.c_void_p(m.data_ptr()),
                            C.c_void_p(v.data_ptr()), 256, 1, 0, 1e-3, 0.9, 0.999, 1e-8, C.byref(loss), C.byref(gn))
    assert code == -4 and b"non-finite" in lib.nf_strerror(code)
    assert not (np.isfinite(loss.value) and np.isfinite(gn.value))  # here the clamp inside tanh keeps the loss finite, the gradient is NaN
