"""Bounds on the UNPINNED decisions of the arithmetic contract (CPU only).

OpenCV 3.4.1 is not in the image, so the accumulation order inside cv::GaussianBlur (OpticalFlow.cpp:73-77)
and nvcc's contraction of Harris.cu:89-91 are decisions of this repository.  The oracle carries the most
likely alternatives as bounding variants (oracle.h: ORC_VAR_BLUR_CVCPU / _FUSED, ORC_HARRIS_GPU_FMAD); these
tests pin what is known about the distance between them and the contract, and tools/unpinned_bounds.py
writes the C2-sized numbers DESIGN.md section 3 quotes (profiles/r04/unpinned_bounds.json)."""
import json
import os
import sys

import numpy as np
import pytest

import _oracle as orc
from introtocomputervision_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import unpinned_bounds as ub  # noqa: E402


def test_cvcpu_filter_is_the_same_filter():
    """The bounding variant computes the same separable correlation: exact on integer images with dyadic
    taps, and its fused row pass IS the contract's row pass (a chain from +0 starts with a plain product)."""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (37, 53)).astype(np.float32)
    g5 = np.array([0.0625, 0.25, 0.375, 0.25, 0.0625], np.float32)
    for fused in (False, True):
        assert np.array_equal(orc.sep_filter_cvcpu(img, g5, g5, fused), orc.sep_filter(img, g5, g5))
    g = orc.gaussian_kernel(15, 5.0)
    one = np.array([1.0], np.float32)
    f = rng.standard_normal((40, 64)).astype(np.float32) * 100
    assert np.array_equal(orc.sep_filter_cvcpu(f, g, one, True), orc.sep_filter(f, g, one))  # row pass alone
    a, b = orc.sep_filter_cvcpu(f, g, g, False), orc.sep_filter(f, g, g)
    assert not np.array_equal(a, b) and np.allclose(a, b, rtol=2e-6, atol=1e-4)  # a different order, a few ulps apart


def test_single_level_lk_stays_within_1e4_under_the_other_orders():
    """One level of LK (lk::calcOpticalFlow) is within north_star's 1e-4 of the contract under OpenCV's CPU
    filter order, fused or not, at every pixel -- no pixel of this frame sits close enough to det = 0.1 to flip."""
    prev, nxt = synth.lk_pair(0x5EED0005, 270, 480, dx=3, dy=-2)
    u0, v0, d0 = orc.lk_flow_ex(prev, nxt, 15, 0, want_det=True)
    for var in (orc.VAR_BLUR_CVCPU, orc.VAR_BLUR_CVCPU | orc.VAR_BLUR_FUSED):
        u1, v1, d1 = orc.lk_flow_ex(prev, nxt, 15, var, want_det=True)
        r = ub.flow_distance((u0, v0), (u1, v1), d0, d1)
        assert r["pixels_that_changed_side_of_det_lt_tau"] == 0
        assert max(r["max_abs_du"], r["max_abs_dv"]) < 5e-5 and r["frac_beyond_1e-4"] == 0.0
        assert r["max_rel_change_of_det"] < 1e-5


def test_det_threshold_flips_a_pixel_under_the_other_order():
    """The discontinuity is real: scale a frame pair so that det(A) of some pixels lands within the filters'
    rounding distance of tau = 0.1 and the two orders put pixels on different sides of `det < tau`
    (OpticalFlow.cpp:82,95) -- there the flow differs by the whole flow, not by 1e-4."""
    prev, nxt = synth.lk_pair(0x5EED0005, 96, 128, dx=1, dy=0)
    _, _, d = orc.lk_flow_ex(prev, nxt, 15, 0, want_det=True)
    target = np.sort(d.ravel())[d.size // 2]            # a typical det; det scales with the 4th power of the image scale
    flips = 0
    for k in range(40):
        s = np.float32((0.1 * (1 + 1e-7 * (k - 20)) / target) ** 0.25)
        u0, v0, d0 = orc.lk_flow_ex(prev * s, nxt * s, 15, 0, want_det=True)
        u1, v1, d1 = orc.lk_flow_ex(prev * s, nxt * s, 15, orc.VAR_BLUR_CVCPU, want_det=True)
        f = (d0 < 0.1) != (d1 < 0.1)
        flips += int(f.sum())
        if f.any():
            assert np.all((u0[f] == 0) | (u1[f] == 0))  # one side returned (0, 0)
    assert flips > 0


def test_pyramidal_lk_distance_is_reported_and_bounded():
    """Through the pyramid the few-ulp differences are amplified by the 1/32-px quantisation of cv::remap's
    coordinates (a coarse flow 1e-6 apart can land in the next 1/32 cell): the other orders stay within 1e-4
    for >= 99 % of the pixels and within 1e-2 everywhere, on the bench's frame at a reduced size."""
    r = ub.lk_bounds(270, 480, 3, 15)
    for k in ("cvcpu_unfused (SSE2 baseline build)", "cvcpu_fused (AVX2/FMA3 build)"):
        v = r[k]
        assert v["frac_beyond_1e-4"] < 0.01 and max(v["max_abs_du"], v["max_abs_dv"]) < 1e-2
        assert v["pixels_that_changed_side_of_det_lt_tau"] == 0


def test_harris_corner_list_survives_the_other_arithmetic():
    """C1: the corner coordinates (north_star: bit-exact) are the same under nvcc's default contraction of
    Harris.cu:89-91 and under harris::cpu's arithmetic; R itself moves by < 1e-4 relative."""
    r = ub.harris_bounds()
    for k in ("gpu_with_nvcc_fmad", "harris_cpu_as_written"):
        assert r[k]["corner_list_identical"] and r[k]["max_rel_dR"] < 1e-4
    assert r["contract_corners"] > 50


def test_committed_c2_numbers_have_the_shape_design_md_quotes():
    d = json.load(open(os.path.join(ROOT, "profiles", "r04", "unpinned_bounds.json")))
    lk = d["lk"]
    assert lk["workload"].startswith("1080x1920") and "5 levels" in lk["workload"]
    for k in ("single_level_cvcpu_unfused", "single_level_cvcpu_fused"):
        assert max(lk[k]["max_abs_du"], lk[k]["max_abs_dv"]) < 1e-4
    for k in ("cvcpu_unfused (SSE2 baseline build)", "cvcpu_fused (AVX2/FMA3 build)"):
        assert lk[k]["frac_beyond_1e-4"] < 0.01
