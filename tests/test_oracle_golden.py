"""The CPU oracle against the committed golden vectors (tests/golden/golden_v1.npz, produced by
tests/golden/make_golden.py) and against the reference's only real data file."""
import os

import numpy as np
import pytest

import _oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(HERE, "golden", "golden_v1.npz"))


def test_golden_regenerates_bit_for_bit(G):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    new = mod.build()
    assert set(new) == set(G.files)
    for k in G.files:
        assert np.array_equal(new[k], G[k]), k


def test_gaussian_kernel_properties(G):
    g = G["gauss15"]
    assert g.dtype == np.float32 and abs(float(g.sum()) - 1.0) < 1e-6
    assert np.array_equal(g, g[::-1]) and g.argmax() == 7
    # cv::getGaussianKernel formula evaluated independently in float64
    x = np.arange(15) - 7.0
    w = np.exp(-x * x / 50.0).astype(np.float32).astype(np.float64)
    assert np.array_equal(g, (w * (1.0 / w.sum())).astype(np.float32))


def test_check_bmp_has_no_strict_maxima():
    """Resources/ProblemSet4/check.bmp is a perfect binary checkerboard: every crossing is a 2x2
    plateau of equal responses, and refineCorners keeps STRICT maxima only (Harris.cpp:128)."""
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(HERE, "golden", "check.bmp")).convert("L"), dtype=np.float32)
    assert img.shape == (120, 160) and set(np.unique(img)) == {0.0, 255.0}
    gx, gy = orc.sobel(img, 3, 1.0)
    R = orc.harris_response(gx, gy, 5, 1.5, 0.04)
    assert R.max() > 5e8
    corners, locs = orc.harris_refine(R, 5e8, 5)
    assert len(locs) == 0 and not corners.any()
    # the rotated board has generic corners
    rot = np.asarray(Image.open(os.path.join(HERE, "golden", "check_rot.bmp")).convert("L"), dtype=np.float32)
    gx, gy = orc.sobel(rot, 3, 1.0)
    R = orc.harris_response(gx, gy, 5, 1.5, 0.04)
    _, locs = orc.harris_refine(R, 5e8, 5)
    assert len(locs) > 10
