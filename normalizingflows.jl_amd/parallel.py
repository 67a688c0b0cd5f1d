"""Sample-sharded data parallelism for the ELBO step.

The Monte-Carlo ELBO is a mean over independent base draws (src/objectives/elbo.jl:68,96), so a
global batch of N samples is split over the ranks of one node: rank r processes the global
samples [offset_r, offset_r + n_r).  Base draws come from Philox keyed by the GLOBAL sample index
(PhiloxRNG.sample_offset), so the union of the shards is the same batch for every world size.
Each rank produces sum_j d(-elbo_j / N)/dtheta and sum_j (-elbo_j / N) over its shard
(nf_elbo_value_and_grad); ONE all-reduce (RCCL over xGMI through torch.distributed's "nccl"
backend; "gloo" in the CPU tests) of the packed [grad ; loss] buffer gives every rank the full
gradient and loss, and every rank applies the same Adam update, so replicas stay bit-identical
without a broadcast.  Parameters, Adam state and layer descriptors are replicated.
"""
from __future__ import annotations


def shard_range(n_global: int, rank: int, world: int):
    """(offset, count) of rank's contiguous shard; the first n_global % world ranks get one extra."""
    if not (0 <= rank < world) or n_global < 0:
        raise ValueError("bad shard arguments")
    base, rem = divmod(n_global, world)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def allreduce_grad_loss(buf, group=None):
    """In-place sum of the packed [grad(P) ; loss] buffer over all ranks (one collective)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


def bucket_bounds(coupling_offsets, n_params: int, couplings_per_bucket: int):
    """Bucket boundaries of the packed [grad ; loss] buffer for a bucketed all-reduce: `coupling_offsets[k]` is the
    theta offset of flat coupling k (ascending: flat coupling order is Optimisers.destructure order), buckets hold
    `couplings_per_bucket` whole couplings, the last one also the loss at index n_params.  Returns [(lo, hi), ...]
    covering [0, n_params + 1) -- the schedule nf_elbo_step uses under a communicator (nf_comm.hip)."""
    offs = list(coupling_offsets)
    if not offs or offs[0] != 0 or any(b <= a for a, b in zip(offs, offs[1:])) or offs[-1] >= n_params or couplings_per_bucket < 1:
        raise ValueError("bad coupling offsets")
    starts = offs[::couplings_per_bucket]
    return [(lo, hi) for lo, hi in zip(starts, starts[1:] + [n_params + 1])]


def allreduce_grad_loss_bucketed(buf, bounds, group=None):
    """The same logical collective as allreduce_grad_loss, sent as one message per bucket (asynchronously, joined before
    returning): every rank issues the same buckets in the same order and receives the same reduced bits."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return buf
    if bounds[0][0] != 0 or bounds[-1][1] != buf.numel() or any(a[1] != b[0] for a, b in zip(bounds, bounds[1:])):
        raise ValueError("buckets must tile the buffer")
    works = [dist.all_reduce(buf[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True) for lo, hi in bounds]
    for w in works:
        w.wait()
    return buf


class ShardedObjective:
    """loss_and_grad(theta) for `optimize`: evaluates this rank's shard with `local_step` and
    all-reduces.  `local_step(theta, offset, count, n_global, step) -> tensor[P + 1]` is
    nf_elbo_value_and_grad on the GPU path (see make_gpu_local_step) or any function with the same
    contract (the CPU tests use the oracle)."""

    def __init__(self, local_step, n_global: int, rank: int, world: int, group=None, buckets=None):
        self.local_step, self.n_global, self.rank, self.world, self.group = local_step, n_global, rank, world, group
        self.offset, self.count = shard_range(n_global, rank, world)
        self.step = 0
        self.buckets = buckets  # None: one message; else [(lo, hi), ...] from bucket_bounds

    def __call__(self, theta):
        buf = self.local_step(theta, self.offset, self.count, self.n_global, self.step)
        self.step += 1
        if self.buckets:
            allreduce_grad_loss_bucketed(buf, self.buckets, self.group)
        else:
            allreduce_grad_loss(buf, self.group)
        return float(buf[-1]), buf[:-1]


def make_gpu_local_step(flow, target, seed: int):
    """local_step backed by the HIP library (device-resident [grad ; loss] buffer)."""
    import ctypes as C

    import torch

    from ._lib import check
    from .flows import _ptr

    out = torch.empty(flow.P + 1, dtype=flow.theta.dtype, device=flow.theta.device)

    def local_step(theta, offset, count, n_global, step):
        ctx = flow.ctx
        check(ctx.lib.nf_elbo_value_and_grad(ctx.ptr, C.byref(flow.desc), C.byref(target.c), _ptr(theta), _ptr(None),
                                             count, n_global, seed, offset, step, _ptr(out)))
        return out

    return local_step


def make_gpu_forward_kl_local_step(flow, xs):
    """local_step for forward-KL training (`train_flow(loglikelihood, flow, xs)`) on a data set sharded by
    column: rank r evaluates columns [offset, offset + count) of `xs` (d x N_global, on this rank's device)
    with nf_loglikelihood_value_and_grad; the same single all-reduce of [grad ; loss] follows."""
    import ctypes as C

    import torch

    from ._lib import check
    from .flows import _ptr, as_batch

    out = torch.empty(flow.P + 1, dtype=flow.theta.dtype, device=flow.theta.device)
    xm, _ = as_batch(xs.to(flow.theta.dtype))

    def local_step(theta, offset, count, n_global, step):
        ctx = flow.ctx
        shard = xm[:, offset:offset + count]  # columns are contiguous in the column-major batch
        check(ctx.lib.nf_loglikelihood_value_and_grad(ctx.ptr, C.byref(flow.desc), _ptr(theta), _ptr(shard), count,
                                                      n_global, _ptr(out)))
        return out

    return local_step
