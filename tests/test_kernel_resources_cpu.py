"""Scratch (private segment) of the benchmarked kernel instantiations, read from the built objects' code-object metadata
exactly as tools/kernel_resources.py does -- runs in the build container, no GPU.

VERDICT r5 item 3: several shipped kernels that the documents called register-resident carried scratch (a dynamically
indexed local array in a streaming kernel, loop-invariant addresses hoisted and spilled, a thread id kept live through the
other role's loops), and nothing watched it.  ZERO lists the instantiations the benchmark configurations launch; a
non-zero private_segment_fixed_size there fails the suite.  KNOWN lists the ones that still spill, each with the reason and
a ceiling, so that they cannot get worse unnoticed either."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

H64 = "NetGeo<1, 2, 2, 1, 4>"
RQ8 = "RqsGeo<1, 1, 1, 8, 4, 2>"

# kernel-name prefixes (demangled, as c++filt prints them) that must not use scratch
ZERO = [
    # cfg 2: fused forward with the activation stash; reverse pass (FULL, forward direction, six-term products in both waves); epilogue
    f"void k_affine_chain<{H64}, false, true, true, false, false, 8>(",
    f"void k_affine_bwd_pair<{H64}, true, false, false, true, true>(",
    f"void k_affine_bwd_pair<{H64}, false, false, false, true, true>(",  # ragged batches / d < 64
    f"void k_affine_epilogue<{H64}, ",
    # cfg 5: the six-term inverse chain; forward-KL's reverse pass of the inverse chain
    f"void k_affine_chain<{H64}, true, false, false, false, true, 8>(",
    f"void k_affine_bwd_pair<{H64}, true, true, false, true, true>(",
    # cfg 3: fused forward chain (six-term output layer); the cooperative reverse kernel is in KNOWN
    f"void k_rqs_chain<{RQ8}, false, true, true>(",
    # cfg 4: the slab reduction (a pure streaming kernel), weight packing
    "void k_wide_reduce_all<",
    "void k_pack_net_images<",
    "void k_reduce_image_slabs<",
    # planar / radial d = 64 x 10 layers with the diagonal-Gaussian target (tools/bench_simple.py)
    "void k_planar_step<PlanarGeo<2, 6>, true>(",
    "void k_radial_step<RadialGeo<2, 10>, true>(",
    "void k_radial_step<RadialGeo<1, 10>, true>(",
    "void k_planar_step<PlanarGeo<1, 6>, true>(",
]

# (prefix, ceiling in bytes, why)
KNOWN = [
    (f"void k_rqs_bwd_coop6<{RQ8}, ", 24,
     "512 registers: the wave's 144 weight registers + accumulators + the spline's state.  Four or five accumulator dwords are stored once "
     "before the tile-group loop and re-loaded in its closing phase (not in the chunk loop).  Zero with the weights as inline-asm AGPR "
     "operands, which is not shipped (a matrix-instruction hazard hipcc does not cover inside asm: nf_rqs.hip, RQS6_MFMA_W)"),
    ("void k_deep_bwd<DeepGeo<3, 2>, ", 348,
     "three hidden layers of 64: 192 weight-gradient accumulators + the recompute's operands exceed 512 registers; the other DeepGeo shapes are scratch-free"),
    ("void k_g64m_bwd<G64M<1, 2, 6>, true>(", 1056,
     "Float64 spline coupling: the general kernels' run-time-K spline state (knot arrays, parameter cotangents) is indexed dynamically"),
    ("void k_g64m_nsf_apply<G64M<1, 2, 6> >(", 672, "the same spline state, forward"),
    ("void k_radial_step<RadialGeo<2, 10>, false>(", 28, "non-Gaussian targets: the target switch's extra state at 256 registers"),
    ("void k_radial_step<RadialGeo<2, 16>, true>(", 28, "sixteen layers at d = 64"),
    ("void k_radial_step<RadialGeo<2, 16>, false>(", 72, "sixteen layers at d = 64, non-Gaussian target"),
    ("void k_planar_step<PlanarGeo<2, 6>, false>(", 212, "non-Gaussian targets at d = 64"),
]


@pytest.fixture(scope="module")
def table():
    import kernel_resources

    bdir = os.path.join(ROOT, "normalizingflows.jl_amd", "build")
    if not os.path.isdir(bdir) or not any(f.endswith(".o") for f in os.listdir(bdir)):
        import __graft_entry__ as ge

        ge.build()
    rows = kernel_resources.kernel_table(bdir)
    assert len(rows) > 200, "the objects' metadata notes were not readable"
    return rows


def _match(rows, prefix):
    hit = [r for r in rows if r[0].startswith(prefix)]
    assert hit, f"no kernel named {prefix!r} in the built objects (renamed? update this list)"
    return hit


def test_benchmarked_kernels_use_no_scratch(table):
    bad = [(r[0][:110], r[4]) for p in ZERO for r in _match(table, p) if r[4] != 0]
    assert not bad, f"scratch in kernels that must be register-resident: {bad}"


def test_known_spills_do_not_grow(table):
    for prefix, cap, why in KNOWN:
        for r in _match(table, prefix):
            assert isinstance(r[4], int) and r[4] <= cap, f"{r[0][:100]}: {r[4]} B of scratch > {cap} ({why})"


def test_register_budgets_of_the_two_wave_kernels(table):
    """Kernels launched with two waves per SIMD (512 threads per workgroup, or 256 with two workgroups per CU) must fit 256
    registers INCLUDING the accumulation half -- otherwise the launch silently drops to one wave per SIMD."""
    for prefix in (f"void k_affine_bwd_pair<{H64}, ", f"void k_affine_chain<{H64}, ", f"void k_rqs_chain<{RQ8}, "):
        for r in _match(table, prefix):
            assert r[2] <= 256, (r[0][:100], r[1], r[2])
