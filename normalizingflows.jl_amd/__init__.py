"""normalizingflows.jl_amd -- MI355X-native ELBO / reverse-KL hot path of NormalizingFlows.jl.

Python host mirror of the reference's public names (src/NormalizingFlows.jl:17,138-141 and
docs/src/api.md) over the C ABI of libnfhip.so (include/nfhip.h).  There is no CPU path:
importing works anywhere, computing needs a gfx950 GPU and the built library.
"""
from ._lib import LIB_PATH, SYMBOLS, Context, NFHipError, context_for, load_library
from .flows import (BananaTarget, CompositeFlow, CrossTarget, DiagGaussTarget, FunnelTarget, WarpedGaussTarget, Flow, MvNormal, create_flow, PhiloxRNG, Transform, as_batch, base_logpdf,
                    device_specific_rand, hamiltonianflow, inverse, layer, logpdf, meanfield, new_batch, nsf, planarflow, radialflow,
                    rand, realnvp, rrule_with_logabsdet_jacobian, target_logp, transform, with_logabsdet_jacobian)
from .parallel import ShardedObjective, allreduce_grad_loss, allreduce_grad_loss_bucketed, bucket_bounds, make_gpu_forward_kl_local_step, make_gpu_local_step, shard_range
from .objectives import (Adam, AdamState, Descent, Momentum, SGDState, adam_update, setup, update, batched_elbos, elbo, elbo_batch, loglikelihood, loglikelihood_value_and_gradient, optimize,
                         train_flow, value_and_gradient)

_device_specific_rand = device_specific_rand  # the reference's (underscored) extension hook name

__all__ = [
    "train_flow", "elbo", "elbo_batch", "loglikelihood", "loglikelihood_value_and_gradient", "optimize",
    "planarflow", "radialflow", "realnvp", "nsf", "meanfield", "hamiltonianflow", "create_flow",
    "with_logabsdet_jacobian", "rrule_with_logabsdet_jacobian", "transform", "inverse", "logpdf", "rand", "layer",
    "MvNormal", "PhiloxRNG", "device_specific_rand", "_device_specific_rand",
    "DiagGaussTarget", "BananaTarget", "FunnelTarget", "WarpedGaussTarget", "CrossTarget", "Adam", "Descent", "Momentum", "value_and_gradient",
]
