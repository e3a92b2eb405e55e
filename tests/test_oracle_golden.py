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
    for k in G.files:  # bytes, not values: the non-finite entries hold NaNs
        assert new[k].shape == G[k].shape and new[k].dtype == G[k].dtype and new[k].tobytes() == G[k].tobytes(), k


def test_golden_regenerates_under_sanitizers():
    """`make -C oracle asan` (-fsanitize=address,undefined) + the golden regeneration and one call of every
    other oracle translation unit under it: an out-of-bounds read, a signed overflow or a misaligned
    access in the checker would make every parity claim worthless.  Runs in a child process because the
    sanitizer runtime has to be preloaded."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan.so not installed")
    r = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, LD_PRELOAD=asan, ORACLE_SO=os.path.join(root, "oracle", "liboracle_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_golden.py"), "--check"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "regenerates bit for bit" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


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
