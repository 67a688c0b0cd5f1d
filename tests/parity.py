"""Parity criteria of the GPU tests, in one place, with the measured errors kept for the record.

Tolerances are BASELINE.md's (fp32 device arithmetic vs the float64 oracle evaluated on the same
fp32-representable inputs) and the reference's own (test/flow.jl):

  per-sample y / ladj / elbo terms : ELEMENT-wise |got - ref| <= Y_ATOL + Y_RTOL * |ref|      (1e-6 + 1e-5 |ref|)
  ELBO mean / loss                  : relative 1e-5
  gradient                          : max |got - ref| <= 1e-4 * ||ref||_inf
  round trip x ~= inv(fwd(x)), lj_fwd ~= -lj_bwd : Julia isapprox (norm-wise) at the reference's rtol --
        1e-6 for RealNVP (test/flow.jl:30-38, Float32 included), 1e-4 for NSF / planar / radial (:97-105,163-171,229-237)
  Float64 paths                     : 1e-10 (y, ladj, loss), 1e-9 (gradient, round trip)

The fp32 floor.  The oracle is dtype-generic: fed float32 arrays it evaluates the same algorithm op by op in IEEE
float32 (numpy), which is what the reference's own Float32 CPU path does.  Its distance from the float64 evaluation
is the error ANY fp32 implementation of the algorithm makes on these inputs; deep / wide flows amplify round-off
(cfg 2 shape: 4x the element-wise tolerance above; the random-init cfg 4 flow, 16 couplings at d = 256 with outputs
up to 2e3: 5 000x).  Checks that pass `floor=` (the float32-oracle result) accept
    |got - ref| <= atol + rtol |ref| + CFLOOR * max|floor - ref|
and record both the device's and the float32 oracle's ratio to the plain tolerance, so the table shows where the
stated tolerance holds outright and where the bound is the arithmetic, not the kernel.

Every check records the measured error under a readable key; conftest.py dumps the table to
gpurun_out/parity_measured.json at the end of a `-m gpu` session (copied to profiles/ per round).
"""
import numpy as np

Y_RTOL, Y_ATOL = 1e-5, 1e-6
LOSS_RTOL = 1e-5
GRAD_RTOL = 1e-4
INV_RTOL = {"realnvp": 1e-6, "meanfield": 1e-6, "nsf": 1e-4, "planar": 1e-4, "radial": 1e-4, "hamiltonian": 1e-4}
F64_RTOL, F64_GRAD = 1e-10, 1e-9

CFLOOR = 3.0

MEASURED = {}


def f32(*arrays):
    """float32 copies (tuples, e.g. oracle targets, are converted element by element)."""
    out = []
    for a in arrays:
        if isinstance(a, tuple):
            out.append(tuple(np.asarray(x, dtype=np.float32) if isinstance(x, np.ndarray) else x for x in a))
        else:
            out.append(np.asarray(a, dtype=np.float32))
    return out[0] if len(out) == 1 else out


def _np(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a, dtype=np.float64)


def record(key, value):
    MEASURED[key] = float(value)


def elementwise(key, got, ref, rtol=Y_RTOL, atol=Y_ATOL, floor=None):
    """max over elements of |got-ref| / (atol + rtol |ref|) is recorded; passes when every element is within
    atol + rtol |ref| (+ CFLOOR * the float32 oracle's worst error on this array, when `floor` is given)."""
    got, ref = _np(got), _np(ref)
    assert got.shape == ref.shape, (key, got.shape, ref.shape)
    if got.size == 0:
        return 0.0
    base = atol + rtol * np.abs(ref)
    err = np.abs(got - ref)
    ratio = float((err / base).max())
    record(key + " [elementwise err / (atol + rtol|ref|)]", ratio)
    # (round 6) the same errors as an rms over the array: the maximum above is one element's figure -- of a few hundred samples the
    # one whose reference value happens to be nearest zero, where the relative tolerance vanishes -- and says little about whether
    # the device arithmetic is worse than IEEE float32; the rms of the device next to the rms of the float32 oracle does
    # (tools/r6_probe_wide_ladj.py)
    rms = float(np.sqrt(np.mean((err / base) ** 2)))
    record(key + " [rms of elementwise err / (atol + rtol|ref|)]", rms)
    extra = 0.0
    if floor is not None:
        ferr = np.abs(_np(floor) - ref)
        record(key + " [fp32-oracle floor / (atol + rtol|ref|)]", float((ferr / base).max()))
        frms = float(np.sqrt(np.mean((ferr / base) ** 2)))
        record(key + " [fp32-oracle floor, rms / (atol + rtol|ref|)]", frms)
        extra = CFLOOR * float(ferr.max())
        assert rms <= max(1.0, CFLOOR * frms), (f"{key}: rms error {rms:.3f}x the tolerance against {frms:.3f}x for the float32 oracle "
                                                f"on the same array")
    worst = float((err / (base + extra)).max())
    assert worst <= 1.0, (f"{key}: worst element is {ratio:.2f}x the plain tolerance (rtol {rtol}, atol {atol})"
                          + (f", {worst:.2f}x the fp32-floor-extended one" if floor is not None else ""))
    return ratio


def isapprox(key, a, b, rtol, floor_err=None):
    """Julia isapprox: norm(a-b) <= rtol * max(norm(a), norm(b)); records the measured ratio.  `floor_err`: the
    same quantity measured on the float32 oracle; the accepted rtol is max(rtol, CFLOOR * floor_err)."""
    a, b = _np(a), _np(b)
    den = max(np.linalg.norm(a), np.linalg.norm(b))
    err = float(np.linalg.norm(a - b) / den) if den > 0 else float(np.linalg.norm(a - b))
    record(key + " [norm-wise rel err]", err)
    if floor_err is not None:
        record(key + " [fp32-oracle floor, norm-wise rel err]", float(floor_err))
        rtol = max(rtol, CFLOOR * float(floor_err))
    assert err <= rtol, f"{key}: norm-wise relative error {err:.3e} > {rtol:.3e}"
    return err


def relerr(a, b):
    a, b = _np(a), _np(b)
    den = max(np.linalg.norm(a), np.linalg.norm(b))
    return float(np.linalg.norm(a - b) / den) if den > 0 else 0.0


def scalar(key, got, ref, rtol=LOSS_RTOL, atol=0.0):
    got, ref = float(got), float(ref)
    err = abs(got - ref) / max(abs(ref), 1e-300)
    record(key + " [rel err]", err)
    assert abs(got - ref) <= atol + rtol * abs(ref), f"{key}: {got} vs {ref} (rel {err:.3e} > {rtol})"
    return err


def gradient(key, got, ref, rtol=GRAD_RTOL, floor=None):
    """max |got-ref| / ||ref||_inf; with `floor` (the float32 oracle's gradient) the accepted error is
    max(rtol, CFLOOR * the float32 oracle's own error by the same measure)."""
    got, ref = _np(got), _np(ref)
    scale = max(float(np.abs(ref).max()), 1e-300)
    err = float(np.abs(got - ref).max() / scale)
    record(key + " [max abs err / |g|inf]", err)
    if floor is not None:
        ferr = float(np.abs(_np(floor) - ref).max() / scale)
        record(key + " [fp32-oracle floor, max abs err / |g|inf]", ferr)
        rtol = max(rtol, CFLOOR * ferr)
    assert err <= rtol, f"{key}: gradient error {err:.3e} of |g|inf > {rtol:.3e}"
    return err
