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
