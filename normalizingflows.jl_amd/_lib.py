"""ctypes binding of libnfhip.so (include/nfhip.h).

There is no CPU fallback: if the library is missing the import of any compute entry point
fails loudly, and every non-zero status from the library raises NFHipError.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnfhip.so")

NF_KIND = {"planar": 0, "radial": 1, "realnvp": 2, "nsf": 3, "meanfield": 4, "hamiltonian": 5, "composite": 6}
NF_DTYPE_F32, NF_DTYPE_F64 = 0, 1
NF_TARGET_DIAGGAUSS, NF_TARGET_BANANA, NF_TARGET_FUNNEL, NF_TARGET_WARPED, NF_TARGET_CROSS = 0, 1, 2, 3, 4
NF_MAX_HIDDEN = 4
NF_ERR_NONFINITE = -4


class NFHipError(RuntimeError):
    pass


class FlowDesc(C.Structure):
    """nf_flow_desc"""

    _fields_ = [
        ("kind", C.c_int32),
        ("dtype", C.c_int32),
        ("d", C.c_int32),
        ("nlayers", C.c_int32),
        ("n_hidden", C.c_int32),
        ("hdims", C.c_int32 * NF_MAX_HIDDEN),
        ("K", C.c_int32),
        ("B", C.c_float),
        ("score", C.c_void_p),  # const nf_target * (Hamiltonian flows), else NULL
        ("base", C.c_void_p),  # const nf_base * (general MvNormal(mu, Sigma) base), NULL = MvNormal(zeros, I)
        ("nsegments", C.c_int32),  # NF_KIND_COMPOSITE
        ("segments", C.c_void_p),  # const nf_flow_desc * [nsegments]
    ]


class Base(C.Structure):
    """nf_base"""

    _fields_ = [("kind", C.c_int32), ("mu", C.c_void_p), ("scale", C.c_void_p), ("logdet", C.c_double)]


NF_BASE_STANDARD, NF_BASE_DIAG, NF_BASE_DENSE = 0, 1, 2


class Target(C.Structure):
    """nf_target"""

    _fields_ = [
        ("kind", C.c_int32),
        ("p0", C.c_void_p),
        ("p1", C.c_void_p),
        ("s0", C.c_double),
        ("s1", C.c_double),
    ]


# every symbol include/nfhip.h declares: name -> (restype, argtypes)
_P, _I32, _I64, _U64, _U32, _D = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_uint32, C.c_double
_DESC, _TGT, _BASE = C.POINTER(FlowDesc), C.POINTER(Target), C.POINTER(Base)
_PD = C.POINTER(C.c_double)
SYMBOLS = {
    "nf_abi_version": (C.c_int, []),
    "nf_strerror": (C.c_char_p, [C.c_int]),
    "nf_ctx_create": (C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    "nf_ctx_destroy": (C.c_int, [_P]),
    "nf_ctx_set_stream": (C.c_int, [_P, _P]),
    "nf_ctx_synchronize": (C.c_int, [_P]),
    "nf_workspace_bytes": (_I64, [_P, _DESC, _I64]),
    "nf_ctx_set_arena": (C.c_int, [_P, _P, C.c_size_t]),
    "nf_ctx_set_stash_budget": (C.c_int, [_P, C.c_int64]),
    "nf_param_count": (_I64, [_DESC]),
    "nf_layer_count": (_I32, [_DESC]),
    "nf_base_sample_logpdf": (C.c_int, [_P, _I32, _I32, _I64, _U64, _U64, _U32, _P, _P]),
    "nf_base_logpdf": (C.c_int, [_P, _I32, _I32, _I64, _P, _P]),
    "nf_base_rand": (C.c_int, [_P, _I32, _BASE, _I32, _I64, _U64, _U64, _U32, _P, _P]),
    "nf_base_logpdf_general": (C.c_int, [_P, _I32, _BASE, _I32, _I64, _P, _P]),
    "nf_flow_fwd": (C.c_int, [_P, _DESC, _P, _P, _I64, _P, _P]),
    "nf_flow_inv": (C.c_int, [_P, _DESC, _P, _P, _I64, _P, _P]),
    "nf_flow_rand": (C.c_int, [_P, _DESC, _P, _I64, _U64, _U64, _U32, _P]),
    "nf_layer_apply": (C.c_int, [_P, _DESC, _I32, _I32, _P, _P, _I64, _P, _P]),
    "nf_flow_bwd": (C.c_int, [_P, _DESC, _P, _P, _P, _P, _P, _I64, _P, _P]),
    "nf_tape_bytes": (_I64, [_P, _DESC, _I64]),
    "nf_flow_fwd_keep": (C.c_int, [_P, _DESC, _P, _P, _I64, _P, _P, _P, C.c_size_t]),
    "nf_flow_bwd_kept": (C.c_int, [_P, _DESC, _P, _P, C.c_size_t, _P, _P, _I64, _P, _P]),
    "nf_target_logp": (C.c_int, [_P, _I32, _TGT, _I32, _I64, _P, _P, _P]),
    "nf_elbo_batch": (C.c_int, [_P, _DESC, _TGT, _P, _P, _I64, _P, _PD]),
    "nf_elbo_batch_rng": (C.c_int, [_P, _DESC, _TGT, _P, _I64, _U64, _U64, _U32, _PD]),
    "nf_loglikelihood": (C.c_int, [_P, _DESC, _P, _P, _I64, _P, _PD]),
    "nf_elbo_value_and_grad": (C.c_int, [_P, _DESC, _TGT, _P, _P, _I64, _I64, _U64, _U64, _U32, _P]),
    "nf_loglikelihood_value_and_grad": (C.c_int, [_P, _DESC, _P, _P, _I64, _I64, _P]),
    "nf_adam_update": (C.c_int, [_P, _I32, _P, _P, _P, _P, _I64, _D, _D, _D, _D, _I64, _P]),
    "nf_sgd_update": (C.c_int, [_P, _I32, _P, _P, _P, _I64, _D, _D, _P]),
    "nf_elbo_step": (C.c_int, [_P, _DESC, _TGT, _P, _P, _P, _I64, _U64, _U32, _D, _D, _D, _D, _PD, _PD]),
    "nf_ctx_weights_changed": (C.c_int, [_P]),
    "nf_ctx_set_weight_cache": (C.c_int, [_P, C.c_int32]),
    "nf_elbo_step_enqueue": (C.c_int, [_P, _DESC, _TGT, _P, _P, _P, _I64, _U64, _P, _D, _D, _D, _D, _P]),
    "nf_comm_get_unique_id": (C.c_int, [_P]),
    "nf_comm_init_rank": (C.c_int, [_P, _P, _I32, _I32]),
    "nf_comm_init_all": (C.c_int, [C.POINTER(_P), _I32]),
    "nf_comm_size": (C.c_int, [_P]),
    "nf_ctx_set_comm_bucket_bytes": (C.c_int, [_P, _I64]),
    "nf_comm_bucket_count": (C.c_int, [_P, _DESC]),
    "nf_allreduce_grad_loss": (C.c_int, [_P, _I32, _P, _I64]),
    "nf_allreduce_grad_loss_all": (C.c_int, [C.POINTER(_P), _I32, _I32, C.POINTER(_P), _I64]),
    "nf_comm_destroy": (C.c_int, [_P]),
    "nf_prof_enable": (C.c_int, [_P, _I32]),
    "nf_prof_read": (C.c_int, [_P, C.c_char_p, _PD, C.POINTER(C.c_int64)]),
    "nf_debug_trace": (C.c_int, [_P, _I32, C.POINTER(C.c_int64), _I32]),
}

_lib = None


def load_library():
    """dlopen libnfhip.so and bind every declared symbol.  Does not touch a GPU."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NFHipError(
            f"{LIB_PATH} is missing: run `python __graft_entry__.py` (hipcc --offload-arch=gfx950) first; "
            "there is no CPU fallback"
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.nf_abi_version() != 4:
        raise NFHipError("libnfhip.so ABI version mismatch")
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != 0:
        msg = load_library().nf_strerror(code)
        raise NFHipError(f"libnfhip status {code}: {msg.decode() if msg else '?'}")


class Context:
    """nf_ctx bound to one device and the torch current stream of that device."""

    def __init__(self, device_index: int = 0, stream_ptr: int = 0):
        self.lib = load_library()
        self.ptr = _P()
        check(self.lib.nf_ctx_create(device_index, _P(stream_ptr), C.byref(self.ptr)))
        self.device_index = device_index

    def set_stream(self, stream_ptr: int) -> None:
        check(self.lib.nf_ctx_set_stream(self.ptr, _P(stream_ptr)))

    def synchronize(self) -> None:
        check(self.lib.nf_ctx_synchronize(self.ptr))

    def close(self) -> None:
        if self.ptr:
            self.lib.nf_ctx_destroy(self.ptr)
            self.ptr = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_contexts = {}


def context_for(device) -> Context:
    """One context per (device, current torch stream)."""
    import torch

    idx = device.index if device.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(idx).cuda_stream
    ctx = _contexts.get(idx)
    if ctx is None:
        ctx = Context(idx, stream)
        _contexts[idx] = ctx
        ctx._stream = stream
    elif ctx._stream != stream:
        ctx.set_stream(stream)
        ctx._stream = stream
    return ctx
