"""world_size-2 test of the sample-sharded data-parallel path on CPU (gloo).

The per-rank compute is stood in for by the CPU oracle (the HIP path needs a GPU); what is under
test is the host logic that the multi-GPU bench and `optimize(all_reduce=...)` use: shard
ranges, global-sample-index Philox draws (the union of shards equals the single-rank batch), the
single all-reduce of [grad ; loss], and replicas staying identical through Adam updates."""
import os
import socket

import numpy as np
import pytest

import nf_oracle as o
from __graft_entry__ import ROOT, load_package

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

SPEC = o.FlowSpec("realnvp", 6, 1, (8, 8))
N_GLOBAL, SEED, STEPS = 50, 7, 3


def _theta0():
    rng = np.random.default_rng(0)
    return o.init_params(SPEC, rng) + 0.05 * rng.standard_normal(o.param_count(SPEC))


def _target():
    rng = np.random.default_rng(1)
    return ("diaggauss", rng.standard_normal(SPEC.d), rng.uniform(size=SPEC.d) + 0.5)


def _oracle_local_step(theta, offset, count, n_global, step):
    xs = o.base_sample(SPEC.d, count, seed=SEED, sample_offset=offset, stream=step)
    th = theta.numpy()
    ys, ladj, states = o.flow_fwd(SPEC, th, xs, keep=True)
    tgt = _target()
    elbos = o.target_logp(tgt, ys) - o.std_normal_logpdf(xs) + ladj
    ybar = -o.target_grad(tgt, ys) / n_global
    lbar = np.full(count, -1.0 / n_global)
    _, grad = o.flow_bwd(SPEC, th, states, ybar, lbar)
    return torch.tensor(np.concatenate([grad, [-elbos.sum() / n_global]]))


def _coupling_offsets():
    """theta offset of every flat coupling of SPEC (the oracle's own parameter walk)."""
    P = o.param_count(SPEC)
    nc = 2 * SPEC.nlayers
    per_pair = P // SPEC.nlayers
    odd = o.param_count(o.FlowSpec("realnvp", SPEC.d, 1, SPEC.hdims)) // 2  # d even: both couplings of a layer are the same size
    return [(k >> 1) * per_pair + (odd if (k & 1) else 0) for k in range(nc)]


def _worker(rank, world, port, ret, bucketed=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nf = load_package()
    buckets = nf.bucket_bounds(_coupling_offsets(), o.param_count(SPEC), 1) if bucketed else None
    obj = nf.ShardedObjective(_oracle_local_step, N_GLOBAL, rank, world, buckets=buckets)
    theta = torch.tensor(_theta0())
    m, v = np.zeros(theta.numel()), np.zeros(theta.numel())
    losses = []
    for t in range(1, STEPS + 1):
        loss, g = obj(theta)
        losses.append(loss)
        th = theta.numpy()
        o.adam_update(th, g.numpy(), m, v, t, lr=1e-2)
    ret[rank] = (losses, theta.numpy().copy())
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_covers_batch():
    nf = load_package()
    for n, w in [(50, 2), (65536, 8), (7, 3), (2, 4)]:
        spans = [nf.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (o0, c0), (o1, _) in zip(spans, spans[1:]):
            assert o0 + c0 == o1


def test_two_rank_gloo_matches_single_rank():
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        res = dict(ret)
    # single-process reference on the full batch
    theta = _theta0()
    m, v = np.zeros_like(theta), np.zeros_like(theta)
    ref_losses = []
    for t in range(1, STEPS + 1):
        buf = _oracle_local_step(torch.tensor(theta), 0, N_GLOBAL, N_GLOBAL, t - 1).numpy()
        ref_losses.append(buf[-1])
        o.adam_update(theta, buf[:-1], m, v, t, lr=1e-2)
    for r in range(world):
        np.testing.assert_allclose(res[r][0], ref_losses, rtol=1e-12)
        np.testing.assert_allclose(res[r][1], theta, rtol=1e-10, atol=1e-12)
    np.testing.assert_array_equal(res[0][1], res[1][1])  # replicas bit-identical


def test_bucketed_all_reduce_equals_the_single_message():
    """The bucketed schedule (one all-reduce per coupling's theta range on its way, the loss in the last bucket, all
    joined before the update -- what nf_elbo_step does for large gradients under a communicator) through
    ShardedObjective with gloo at world size 2: same losses and parameters as the one-message run on every rank, and
    the replicas identical to each other."""
    nf = load_package()
    P = o.param_count(SPEC)
    offs = _coupling_offsets()
    bounds = nf.bucket_bounds(offs, P, 1)
    assert bounds[0][0] == 0 and bounds[-1][1] == P + 1 and len(bounds) == 2 * SPEC.nlayers
    assert all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
    assert nf.bucket_bounds(offs, P, 2) == [(0, P + 1)]
    world = 2
    res = {}
    for bucketed in (False, True):
        with mp.Manager() as mgr:
            ret = mgr.dict()
            mp.spawn(_worker, args=(world, _free_port(), ret, bucketed), nprocs=world, join=True)
            res[bucketed] = dict(ret)
    for r in range(world):
        assert res[True][r][0] == res[False][r][0]
        np.testing.assert_array_equal(res[True][r][1], res[False][r][1])
    np.testing.assert_array_equal(res[True][0][1], res[True][1][1])


# ---- forward KL: the data set (not the base draws) is what is sharded ---------------------------------
def _fkl_data():
    return 0.9 * np.random.default_rng(3).standard_normal((SPEC.d, N_GLOBAL))


def _oracle_fkl_local_step(theta, offset, count, n_global, step):
    ys = _fkl_data()[:, offset:offset + count]
    loss, grad = o.neg_loglik_value_and_grad(SPEC, theta.numpy(), ys, n_global=n_global)
    return torch.tensor(np.concatenate([grad, [loss]]))


def _fkl_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nf = load_package()
    obj = nf.ShardedObjective(_oracle_fkl_local_step, N_GLOBAL, rank, world)
    loss, g = obj(torch.tensor(_theta0()))
    ret[rank] = (loss, g.numpy().copy())
    dist.destroy_process_group()


def test_two_rank_forward_kl_matches_single_rank():
    """train_flow(loglikelihood, ...) data-parallel: column shards of the data set, one all-reduce of
    [grad ; loss] (same contract as nf_loglikelihood_value_and_grad with N_global)."""
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_fkl_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        res = dict(ret)
    lref, gref = o.neg_loglik_value_and_grad(SPEC, _theta0(), _fkl_data())
    for r in range(world):
        assert res[r][0] == pytest.approx(lref, rel=1e-12)
        np.testing.assert_allclose(res[r][1], gref, rtol=1e-10, atol=1e-13)


def test_bench_launcher_fails_loudly_without_enough_gpus():
    """`python bench.py --gpus 2` on a box with fewer than 2 GPUs must exit non-zero with a message -- never fall back
    to one rank and print a single-GPU number under an 8-GPU flag."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NF_BENCH_ONE_DEVICE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "GPU(s) visible" in p.stderr
    assert not any(ln.startswith("{") for ln in p.stdout.splitlines())
