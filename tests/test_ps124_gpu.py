"""Parity of the HIP Harris / SIFT-keypoint / stereo / Hough kernels (through the C ABI)
against the CPU oracle.  Integer and index outputs (corner lists, disparities, vote counts,
peak lists) and the float Harris response are BIT-EXACT; atan2-based angles carry a stated
tolerance (device atan2f vs libm atan2f)."""
import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


@pytest.fixture(scope="module")
def M():
    from introtocomputervision_amd import harris, hough, stereo, synth
    return harris, stereo, hough, synth


# ------------------------------------------------------------------------------ ps4 ------

@pytest.mark.parametrize("ksize,scale", [(3, 1.0), (1, 1.0), (5, 1.0), (7, 1.0), (3, 1.0 / 9.0)])
def test_sobel(M, ksize, scale):
    harris, stereo, hough, synth = M
    img = synth.smooth_noise(21, 75, 101)
    ex, ey = orc.sobel(img, ksize, np.float32(scale))
    gx, gy = harris.getGradients(dev(img), ksize, np.float32(scale))
    assert np.array_equal(host(gx), ex) and np.array_equal(host(gy), ey)
    hx, hy = harris.getGradients(img, ksize, np.float32(scale))
    assert np.array_equal(hx, ex) and np.array_equal(hy, ey)


@pytest.mark.parametrize("rows,cols", [(120, 160), (61, 200), (33, 47)])
@pytest.mark.parametrize("win,sigma", [(5, 1.5), (3, 1.0), (9, 2.0)])
def test_harris_response(M, rows, cols, win, sigma):
    harris, stereo, hough, synth = M
    img = synth.checkerboard(rows, cols, square=20, seed=3)
    gx, gy = orc.sobel(img, 3, 1.0)
    exp = orc.harris_response(gx, gy, win, sigma, 0.04)
    got = harris.getCornerResponse(dev(gx), dev(gy), win, sigma, 0.04)
    assert np.array_equal(host(got), exp)
    assert np.array_equal(harris.getCornerResponse(gx, gy, win, sigma, 0.04), exp)


@pytest.mark.parametrize("rows,cols", [(480, 640), (61, 200), (17, 65), (130, 3)])
@pytest.mark.parametrize("win,sigma", [(5, 1.5), (3, 1.0), (7, 2.0), (9, 2.0)])
def test_harris_response_cpu_arithmetic(M, rows, cols, win, sigma):
    """MICV_HARRIS_CPU: harris::cpu::getCornerResponse as written (Harris.cpp:78-92) -- unfused
    `secondMoment + weight * gradVals`, cv::determinant in double, the difference rounded to float once --
    bit-exact against the oracle's twin on the tiled kernels (windows 3 / 5 / 7), the generic one (9), the
    device and the host entry point; it differs from the gpu:: arithmetic, and C1's corner list does not."""
    harris, stereo, hough, synth = M
    img = synth.checkerboard(rows, cols, square=40 if rows >= 400 else 12, seed=0x5EED0001)
    img[rows // 2, cols // 2] += 3e4  # a large value: products near 1e18, where double and float determinants part
    gx, gy = orc.sobel(img, 3, 1.0)
    exp = orc.harris_response_ex(gx, gy, win, sigma, 0.04, orc.HARRIS_CPU)
    got = harris.getCornerResponse(dev(gx), dev(gy), win, sigma, 0.04, cpu_arithmetic=True)
    assert host(got).tobytes() == exp.tobytes()
    assert harris.getCornerResponse(gx, gy, win, sigma, 0.04, cpu_arithmetic=True).tobytes() == exp.tobytes()
    gpu = orc.harris_response(gx, gy, win, sigma, 0.04)
    if rows * cols > 4000:
        assert not np.array_equal(exp, gpu)
    if (rows, cols, win) == (480, 640, 5):
        img = synth.checkerboard(rows, cols, square=40, seed=0x5EED0001)
        dgx, dgy = harris.getGradients(dev(img), 3)
        Rc = harris.getCornerResponse(dgx, dgy, 5, 1.5, 0.04, cpu_arithmetic=True)
        Rg = harris.getCornerResponse(dgx, dgy, 5, 1.5, 0.04)
        _, lc = harris.refineCorners(Rc, 5e8, 5)
        _, lg = harris.refineCorners(Rg, 5e8, 5)
        assert len(lc) == 165 and np.array_equal(host(lc), host(lg))


def test_harris_response_cpu_arithmetic_nonfinite_and_views(M):
    harris, stereo, hough, synth = M
    rng = np.random.default_rng(11)
    bx = (rng.standard_normal((75, 456)) * 300).astype(np.float32)
    by = (rng.standard_normal((75, 456)) * 300).astype(np.float32)
    bx[7, 130] = np.inf
    by[40, 70] = np.nan
    bx[50, 300] = 1e19  # squares overflow float, not double
    for xoff, cols in ((0, 448), (1, 330), (4, 260)):
        ex = orc.harris_response_ex(np.ascontiguousarray(bx[:, xoff:xoff + cols]), np.ascontiguousarray(by[:, xoff:xoff + cols]),
                                    5, 1.5, 0.04, orc.HARRIS_CPU)
        got = host(harris.getCornerResponse(dev(bx)[:, xoff:xoff + cols], dev(by)[:, xoff:xoff + cols], 5, 1.5, 0.04,
                                            cpu_arithmetic=True))
        assert got.tobytes() == ex.tobytes()


def test_harris_pipeline_c1(M):
    """C1: 480x640 checkerboard with config/ps4.yaml parameters (sobel 3, window 5, sigma 1.5,
    alpha 0.04, threshold 5e8, minDistance 5): corner list identical to the oracle's, and it is
    the lattice of checker crossings."""
    harris, stereo, hough, synth = M
    img = synth.checkerboard(480, 640, square=40, seed=0x5EED0001)
    gx, gy = harris.getGradients(dev(img), 3)
    R = harris.getCornerResponse(gx, gy, 5, 1.5, 0.04)
    corners, locs = harris.refineCorners(R, 5e8, 5)
    egx, egy = orc.sobel(img, 3, 1.0)
    eR = orc.harris_response(egx, egy, 5, 1.5, 0.04)
    ec, el = orc.harris_refine(eR, 5e8, 5)
    assert np.array_equal(host(R), eR)
    assert np.array_equal(host(corners), ec)
    assert np.array_equal(host(locs), el)
    assert len(el) > 50
    # every detected corner sits within 2 px of a checker crossing
    d = np.minimum(el % 40, 40 - el % 40)
    assert (d <= 2).all()
    # keypoints / angles
    kp = harris.getKeypoints(gx, gy, locs, 10)
    ekp = orc.sift_keypoints(egx, egy, el, 10)
    assert np.array_equal(host(kp)[:, :3], ekp[:, :3])
    assert np.allclose(host(kp)[:, 3], ekp[:, 3], rtol=0, atol=1e-3)  # degrees; atan2f differs by ulps
    ang = harris.getAnglesFromGradients(gx, gy)
    assert np.allclose(host(ang), orc.sift_angles(egx, egy), rtol=0, atol=1e-5)


@pytest.mark.parametrize("rows,cols", [(64, 64), (33, 129), (97, 61), (1, 70), (70, 1), (2, 2)])
@pytest.mark.parametrize("ksize", [3, 5, 7])
def test_sobel_fused_tiles_and_tiny_images(M, rows, cols, ksize):
    """The fused 64x32-tile Sobel on sizes around / below the tile and the reflect border."""
    harris, stereo, hough, synth = M
    img = synth.smooth_noise(5, rows, cols)
    for scale in (1.0, 1.0 / 9.0):
        ex, ey = orc.sobel(img, ksize, np.float32(scale))
        gx, gy = harris.getGradients(dev(img), ksize, np.float32(scale))
        assert np.array_equal(host(gx), ex) and np.array_equal(host(gy), ey)


@pytest.mark.parametrize("rows,cols", [(16, 64), (17, 65), (5, 300), (130, 3)])
@pytest.mark.parametrize("win,sigma", [(3, 0.8), (5, 1.5), (7, 2.0), (11, 2.5)])
def test_harris_response_tiled_and_generic(M, rows, cols, win, sigma):
    harris, stereo, hough, synth = M
    rng = np.random.default_rng(rows * 1000 + cols)
    gx = (rng.standard_normal((rows, cols)) * 300).astype(np.float32)
    gy = (rng.standard_normal((rows, cols)) * 300).astype(np.float32)
    exp = orc.harris_response(gx, gy, win, sigma, 0.04)
    assert np.array_equal(host(harris.getCornerResponse(dev(gx), dev(gy), win, sigma, 0.04)), exp)


@pytest.mark.parametrize("win,sigma", [(3, 0.8), (5, 1.5), (7, 2.0)])
@pytest.mark.parametrize("xoff,cols", [(0, 448), (0, 331), (1, 330), (4, 260), (2, 200)])
def test_harris_response_interior_tiles_and_views(M, win, sigma, xoff, cols):
    """Interior tiles stage with aligned float4 loads and store float4; column-offset views of a wider
    device image (base not 16-byte aligned) and odd widths must take the scalar path and agree."""
    import torch
    harris, stereo, hough, synth = M
    rows = 75
    rng = np.random.default_rng(win * 100 + cols + xoff)
    bx = (rng.standard_normal((rows, 456)) * 300).astype(np.float32)
    by = (rng.standard_normal((rows, 456)) * 300).astype(np.float32)
    bx[7, 130] = np.inf
    by[40, 70] = np.nan
    dx, dy = dev(bx), dev(by)
    gx, gy = dx[:, xoff:xoff + cols], dy[:, xoff:xoff + cols]
    exp = orc.harris_response(np.ascontiguousarray(bx[:, xoff:xoff + cols]), np.ascontiguousarray(by[:, xoff:xoff + cols]),
                              win, sigma, 0.04)
    got = host(harris.getCornerResponse(gx, gy, win, sigma, 0.04))
    assert got.tobytes() == exp.tobytes()


@pytest.mark.parametrize("dist", [0, 1, 3, 5, 16, 17])
def test_harris_refine_ties_nan_inf(M, dist):
    """Quantised responses (many exact ties), -0/+0, infinities and NaNs: the tiled (max,
    multiplicity) NMS (dist <= 16) and the scanning kernel (17) against the oracle's loop."""
    harris, stereo, hough, synth = M
    rng = np.random.default_rng(dist + 40)
    R = rng.integers(0, 6, (83, 150)).astype(np.float32)
    R[rng.random(R.shape) < 0.02] = np.nan
    R[rng.random(R.shape) < 0.01] = np.inf
    R[rng.random(R.shape) < 0.01] = -np.inf
    R[5, 5] = 9.0          # an isolated strict maximum
    R[40, 70] = -0.0
    R[0, 149] = 11.0       # corners of the image: clamped windows
    R[82, 0] = 12.0
    for thr in (3.0, 0.0, -np.inf):
        ec, el = orc.harris_refine(R, thr, dist)
        c, l = harris.refineCorners(dev(R), thr, dist)
        assert np.array_equal(host(l), el)
        assert np.array_equal(host(c), ec)


@pytest.mark.parametrize("thr,dist", [(0.5, 1), (0.0, 3), (-1.0, 2), (0.9, 0)])
def test_harris_refine_random(M, thr, dist):
    harris, stereo, hough, synth = M
    rng = np.random.default_rng(7)
    R = rng.random((97, 131)).astype(np.float32)
    R[10, 10:14] = 2.0  # ties: none of them is a strict maximum
    R[0, 0] = 5.0
    R[-1, -1] = 5.0
    ec, el = orc.harris_refine(R, thr, dist)
    c, l = harris.refineCorners(dev(R), thr, dist)
    assert np.array_equal(host(c), ec) and np.array_equal(host(l), el)
    c2, l2 = harris.refineCorners(R, thr, dist)
    assert np.array_equal(c2, ec) and np.array_equal(l2, el)
    # capacity smaller than the count: the first `cap` in row-major order
    c3, l3 = harris.refineCorners(dev(R), thr, dist, capacity=5)
    assert np.array_equal(host(l3), el[:5])


@pytest.mark.parametrize("rows,cols,density", [(2160, 3840, 0.02), (2160, 3840, 0.9), (480, 640, 0.3), (33, 31, 0.5), (1, 5000, 0.5),
                                               (300, 1100, 0.0), (300, 1100, 1.0)])
def test_ordered_list_one_launch_vs_three(M, rows, cols, density):
    """The ordered corner list comes from one launch (chained scan with look-back) and, with
    MICV_OPT_COMPACT_3PASS, from count / scan / emit: both must give numpy's row-major list -- at sizes where
    the look-back spans several 64-chunk steps (4K = 8100 chunks), with nearly empty and nearly full chunks,
    for repeated calls on one context (the state is left zeroed) and from two streams of one context."""
    import torch
    from introtocomputervision_amd import _capi
    harris, stereo, hough, synth = M
    rng = np.random.default_rng(rows + cols)
    # minDistance 0: every pixel >= threshold is kept (Harris.cpp:119-135 with an empty neighbourhood)
    R = rng.random((rows, cols)).astype(np.float32)
    thr = 1.0 - density if 0.0 < density < 1.0 else (2.0 if density == 0.0 else -1.0)
    ys, xs = np.nonzero(R.astype(np.float64) >= thr)
    exp = np.stack([ys, xs], axis=1).astype(np.int32)
    dR = dev(R)
    c1 = _capi.Context(0)
    c1.set_option(_capi.OPT_COMPACT_3PASS, -1)  # one launch at every size (the default switches at 1 M elements)
    c3 = _capi.Context(0)
    c3.set_option(_capi.OPT_COMPACT_3PASS, 1)
    c0 = _capi.Context(0)
    for ctx in (c1, c3, c1, c0, c1):
        _, locs = harris.refineCorners(dR, thr, 0, ctx=ctx)
        assert np.array_equal(host(locs), exp)
    _, locs = harris.refineCorners(dR, thr, 0, capacity=7, ctx=c1)
    assert np.array_equal(host(locs), exp[:7])
    # a second (non-blocking) stream gets its own state, zeroed on THAT stream before its first launch; a context's
    # scratch is shared, so calls on different streams are not overlapped (mi_cv.h)
    s2 = torch.cuda.Stream()
    for _ in range(2):
        with torch.cuda.stream(s2):
            dR2 = dR.clone()
            _, l2 = harris.refineCorners(dR2, thr, 0, ctx=c1)
        torch.cuda.synchronize()
        _, l1 = harris.refineCorners(dR, thr, 0, ctx=c1)
        torch.cuda.synchronize()
        assert np.array_equal(host(l1), exp) and np.array_equal(host(l2), exp)


def test_reference_check_bmp(M):
    """The only real image in the reference (Resources/ProblemSet4/check.bmp, 160x120 8-bit)."""
    harris, stereo, hough, synth = M
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "check.bmp")
    if not os.path.exists(path):
        pytest.skip("fixture not present")
    from PIL import Image
    img = np.asarray(Image.open(path).convert("L"), dtype=np.float32)
    gx, gy = harris.getGradients(dev(img), 3)
    R = harris.getCornerResponse(gx, gy, 5, 1.5, 0.04)
    egx, egy = orc.sobel(img, 3, 1.0)
    eR = orc.harris_response(egx, egy, 5, 1.5, 0.04)
    assert np.array_equal(host(R), eR)
    c, l = harris.refineCorners(R, 5e8, 5)
    ec, el = orc.harris_refine(eR, 5e8, 5)
    assert np.array_equal(host(l), el)


# ------------------------------------------------------------------------------ ps2 ------

@pytest.mark.parametrize("rad", [1, 3, 5, 7, 10, 12])
def test_ssd_integer_images(M, rad):
    harris, stereo, hough, synth = M
    left, right, d = synth.stereo_pair(0x5EED0002, 70, 150)
    exp = orc.disparity_ssd(left, right, rad, -40, 0)
    got = stereo.disparitySSD(dev(left), dev(right), rad, -40, 0)
    assert np.array_equal(host(got), exp)


@pytest.mark.parametrize("flags", [0, 1, 2, 3])
def test_ssd_float_images_and_flags(M, flags):
    harris, stereo, hough, synth = M
    rng = np.random.default_rng(5)
    left = (rng.random((45, 133)) * 255).astype(np.float32)
    right = np.roll(left, -6, axis=1) + (rng.random((45, 133)) * 3).astype(np.float32)
    exp = orc.disparity_ssd(left, right, 4, -20, 5, flags)
    got = stereo.disparitySSD(dev(left), dev(right), 4, -20, 5, flags)
    assert np.array_equal(host(got), exp)
    assert np.array_equal(stereo.disparitySSD(left, right, 4, -20, 5, flags), exp)


def test_ssd_min_ssd_threshold_leaves_minus_one(M):
    harris, stereo, hough, synth = M
    left = np.zeros((40, 90), np.float32)
    right = np.full((40, 90), 255, np.float32)  # 11x10 window of 255^2 > 5e6
    exp = orc.disparity_ssd(left, right, 5, 0, 8, 3)
    got = stereo.disparitySSD(dev(left), dev(right), 5, 0, 8, 3)
    assert np.array_equal(host(got), exp) and (exp == -1).all()


@pytest.mark.parametrize("rad,lo,hi", [(3, -30, 0), (5, 0, 25), (12, -10, 10)])
def test_ssd_serial_semantics(M, rad, lo, hi):
    harris, stereo, hough, synth = M
    left, right, d = synth.stereo_pair(99, 50, 120)
    exp = orc.disparity_ssd_serial(left, right, rad, lo, hi)
    got = stereo.disparitySSD(dev(left), dev(right), rad, lo, hi, stereo.STEREO_SERIAL)
    assert np.array_equal(host(got), exp)


def test_ssd_serial_equals_cuda_semantics_inside(M):
    """for integer-valued inputs the (2r+1)^2 CUDA-addressing cost equals serial::
    wherever serial:: searches the full range."""
    harris, stereo, hough, synth = M
    left, right, d = synth.stereo_pair(7, 60, 200)
    a = host(stereo.disparitySSD(dev(left), dev(right), 4, -30, 0))
    b = host(stereo.disparitySSD(dev(left), dev(right), 4, -30, 0, stereo.STEREO_SERIAL))
    assert np.array_equal(a[:, 40:-40], b[:, 40:-40])


@pytest.mark.parametrize("rad", [2, 5, 11])
@pytest.mark.parametrize("flags", [0, 1])
def test_ncorr(M, rad, flags):
    harris, stereo, hough, synth = M
    left, right, d = synth.stereo_pair(0x5EED0002, 48, 140)
    left = left + 1.0  # keep energies > 0
    right = right + 1.0
    exp = orc.disparity_ncorr(left, right, rad, -30, 0, flags)
    got = stereo.disparityNCorr(dev(left), dev(right), rad, -30, 0, flags)
    assert np.array_equal(host(got), exp)


def test_ssd_1080p_known_ramp(M):
    """C3 at full size: known disparity ramp d(y) = 8 + floor(96 y / H), r = 5, 128 candidates.
    Property check (no oracle at this size): interior pixels recover -d(y) exactly."""
    harris, stereo, hough, synth = M
    left, right, negd = synth.stereo_pair(0x5EED0002, 1080, 1920)
    got = host(stereo.disparitySSD(dev(left), dev(right), 5, -127, 0))
    # rows whose whole 11-row window lies inside one constant-disparity band
    ys = np.array([y for y in range(5, 1075) if negd[y - 5] == negd[y + 5]])
    assert len(ys) >= 90
    inner = got[ys][:, 140:-140]
    want = np.broadcast_to(negd[ys][:, None], inner.shape)
    assert (inner == want).mean() > 0.999


def _stereo_ctxs():
    """(exact-sum kernels for disparitySSD = the default, float kernels only): MICV_OPT_STEREO_EXACT = 0 / -1."""
    from introtocomputervision_amd._capi import Context, OPT_STEREO_EXACT
    exact, flt = Context(0), Context(0)
    exact.set_option(OPT_STEREO_EXACT, 0)
    flt.set_option(OPT_STEREO_EXACT, -1)
    return exact, flt


def _u8_pair(rng, rows, cols, kind):
    if kind == "noise":
        left = rng.integers(0, 256, (rows, cols)).astype(np.float32)
        right = np.roll(left, -5, axis=1)
        right[::3] = rng.integers(0, 256, right[::3].shape)
    elif kind == "levels":  # three grey levels: equal costs / scores everywhere
        left = (rng.integers(0, 3, (rows, cols)) * 100).astype(np.float32)
        right = (rng.integers(0, 3, (rows, cols)) * 100).astype(np.float32)
    elif kind == "flat":  # flat patches, a black corner, a saturated band
        left = rng.integers(0, 256, (rows, cols)).astype(np.float32)
        left[:, cols // 3: cols // 2] = 255
        left[rows // 2:, : cols // 4] = 0
        right = np.roll(left, -3, axis=1)
    else:  # "far": the largest costs the 8-bit range allows
        left = rng.integers(200, 256, (rows, cols)).astype(np.float32)
        right = rng.integers(0, 30, (rows, cols)).astype(np.float32)
    return left, right


@pytest.mark.parametrize("rad,lo,hi,flags,kind", [
    (1, -20, 0, 0, "noise"), (2, -70, 5, 1, "levels"), (3, 0, 127, 2, "far"), (4, -128, 127, 3, "noise"),
    (5, -127, 0, 0, "flat"), (5, -30, 40, 4, "noise"), (5, -128, 127, 4, "levels"), (6, -70, 5, 8, "far"),
    (6, -128, 127, 0, "levels"), (7, -100, 27, 11, "noise"), (7, -3, -3, 0, "flat"), (5, 0, 63, 3, "far")])
def test_ssd_exact_sum_kernels(M, rad, lo, hi, flags, kind):
    """8-bit-valued images take the exact-sum kernels (stereo_exact.hip: dot4 column sums, sliding windows, lanes =
    disparities); they, the float kernels and the oracle agree byte for byte -- ragged sizes, one to four chunks of
    64 disparities, every flag, serial:: semantics, ties (first minimum) and the largest costs."""
    harris, stereo, hough, synth = M
    exact, flt = _stereo_ctxs()
    rng = np.random.default_rng(rad * 1000 + hi)
    for rows, cols in ((9, 40), (33, 141), (70, 203)):
        left, right = _u8_pair(rng, rows, cols, kind)
        if flags & 4:
            exp = orc.disparity_ssd_serial(left, right, rad, lo, hi)
        else:
            exp = orc.disparity_ssd(left, right, rad, lo, hi, flags)
        assert np.array_equal(host(stereo.disparitySSD(dev(left), dev(right), rad, lo, hi, flags, ctx=exact)), exp)
        assert np.array_equal(host(stereo.disparitySSD(dev(left), dev(right), rad, lo, hi, flags, ctx=flt)), exp)
        assert np.array_equal(stereo.disparitySSD(left, right, rad, lo, hi, flags), exp)  # host entry, default options


def test_ssd_exact_sum_kernels_leave_other_images_to_the_float_kernels(M):
    """One pixel that is not an integer in 0..255 (a fraction, a negative, 256, NaN): the pre-pass finds it on the
    device and the float kernels do the call -- same result as with the exact-sum path switched off, and the oracle's."""
    harris, stereo, hough, synth = M
    exact, flt = _stereo_ctxs()
    rng = np.random.default_rng(3)
    for bad in (0.5, -1.0, 256.0, np.nan):
        for side in (0, 1):
            left, right = _u8_pair(rng, 64, 200, "noise")
            (left if side == 0 else right)[40, 100] = bad
            exp = orc.disparity_ssd(left, right, 5, -30, 0)
            assert np.array_equal(host(stereo.disparitySSD(dev(left), dev(right), 5, -30, 0, ctx=exact)), exp)
            assert np.array_equal(host(stereo.disparitySSD(dev(left), dev(right), 5, -30, 0, ctx=flt)), exp)
            # the next call on the same context with clean images takes the exact-sum kernels again
            l2, r2 = _u8_pair(rng, 64, 200, "noise")
            assert np.array_equal(host(stereo.disparitySSD(dev(l2), dev(r2), 5, -30, 0, ctx=exact)),
                                  orc.disparity_ssd(l2, r2, 5, -30, 0))


@pytest.mark.parametrize("rad,lo,hi,flags,kind", [
    (2, -20, 0, 0, "noise"), (3, -70, 5, 1, "levels"), (5, -127, 0, 0, "flat"), (5, -128, 127, 8, "levels"), (4, -5, 5, 0, "far")])
def test_ncorr_on_8bit_images_keeps_the_float_kernels(M, rad, lo, hi, flags, kind):
    """disparityNCorr on the same 8-bit-valued pairs (equal and nearly equal scores: three grey levels, flat patches,
    windows of zeros): the float kernels with either option value, byte for byte the oracle's result.  (An exact-sum NCC
    search was built in r06 and dropped: slower than these kernels on flat and on textured pairs)"""
    harris, stereo, hough, synth = M
    exact, flt = _stereo_ctxs()
    rng = np.random.default_rng(rad * 77 + hi)
    for rows, cols in ((9, 40), (33, 141), (64, 203)):
        left, right = _u8_pair(rng, rows, cols, kind)
        exp = orc.disparity_ncorr(left, right, rad, lo, hi, flags)
        assert np.array_equal(host(stereo.disparityNCorr(dev(left), dev(right), rad, lo, hi, flags, ctx=exact)), exp)
        assert np.array_equal(host(stereo.disparityNCorr(dev(left), dev(right), rad, lo, hi, flags, ctx=flt)), exp)


def test_ssd_1080p_c3_bit_exact(M):
    """BASELINE C3 at full size against the oracle: 1080x1920, 11x11 window, all 128 candidate
    disparities (d in [-127, 0]), every pixel, byte for byte -- through the exact-sum kernels (the default for this
    8-bit-valued pair) and through the float kernels."""
    harris, stereo, hough, synth = M
    left, right, negd = synth.stereo_pair(0x5EED0002, 1080, 1920)
    exp = orc.disparity_ssd(left, right, 5, -127, 0)
    got = host(stereo.disparitySSD(dev(left), dev(right), 5, -127, 0))
    assert got.dtype == np.int8 and np.array_equal(got, exp)
    exact, flt = _stereo_ctxs()
    assert np.array_equal(host(stereo.disparitySSD(dev(left), dev(right), 5, -127, 0, ctx=flt)), exp)
    # the reference's own geometry: right-reference pass d in [0, 127]
    exp_r = orc.disparity_ssd(right, left, 5, 0, 127)
    assert np.array_equal(host(stereo.disparitySSD(dev(right), dev(left), 5, 0, 127)), exp_r)


@pytest.mark.parametrize("rows,cols,rad,dmin", [(540, 960, 5, -63), (1080, 1920, 5, -127), (300, 700, 7, -95)])
def test_ncorr_at_size(M, rows, cols, rad, dmin):
    """disparityNCorr against the oracle at BASELINE-scale sizes (the small cases are in test_ncorr)."""
    harris, stereo, hough, synth = M
    left, right, negd = synth.stereo_pair(0x5EED0002, rows, cols)
    left = left + 1.0  # keep window energies > 0
    right = right + 1.0
    exp = orc.disparity_ncorr(left, right, rad, dmin, 0)
    got = host(stereo.disparityNCorr(dev(left), dev(right), rad, dmin, 0))
    assert np.array_equal(got, exp)


# ------------------------------------------------------------------------------ ps1 ------

@pytest.mark.parametrize("rho_bin,theta_bin", [(1, 1), (2, 3), (5, 7)])
def test_hough_lines(M, rho_bin, theta_bin):
    harris, stereo, hough, synth = M
    mask, lines, circles = synth.hough_mask(180, 260, n_lines=6, radii=(20, 30))
    exp = orc.hough_lines(mask, rho_bin, theta_bin)
    got = hough.houghLinesAccumulate(dev(mask), rho_bin, theta_bin)
    assert tuple(got.shape) == exp.shape
    assert np.array_equal(host(got), exp)
    assert np.array_equal(hough.houghLinesAccumulate(mask, rho_bin, theta_bin), exp)
    assert exp.sum() == (mask > 0).sum() * len(range(-90, 90, theta_bin))


@pytest.mark.parametrize("radius", [20, 30, 7])
def test_hough_circles(M, radius):
    harris, stereo, hough, synth = M
    mask, lines, circles = synth.hough_mask(150, 210, n_lines=2, radii=(20, 30))
    exp = orc.hough_circles(mask, radius)
    got = hough.houghCirclesAccumulate(dev(mask), radius)
    assert np.array_equal(host(got), exp)
    assert np.array_equal(hough.houghCirclesAccumulate(mask, radius), exp)


def test_hough_lines_tall_image_takes_the_global_atomics_kernel(M):
    """rho axis of 2*ceil(sqrt(H^2+W^2)) > 16384 bins does not fit one workgroup's LDS."""
    harris, stereo, hough, synth = M
    rng = np.random.default_rng(3)
    mask = ((rng.random((9000, 48)) < 0.01) * 255).astype(np.uint8)
    exp = orc.hough_lines(mask, 1, 2)
    assert exp.shape[0] * 4 > 64 * 1024
    assert np.array_equal(host(hough.houghLinesAccumulate(dev(mask), 1, 2)), exp)


@pytest.mark.parametrize("rows,cols,radius", [(300, 400, 200), (65, 33, 1), (40, 700, 64)])
def test_hough_circles_reach_beyond_tiles(M, rows, cols, radius):
    """Radii larger than the 64x32 accumulator tile, and images smaller than one tile."""
    harris, stereo, hough, synth = M
    rng = np.random.default_rng(radius)
    mask = ((rng.random((rows, cols)) < 0.01) * 255).astype(np.uint8)
    assert np.array_equal(host(hough.houghCirclesAccumulate(dev(mask), radius)), orc.hough_circles(mask, radius))


@pytest.mark.parametrize("num_peaks,thr", [(10, 50), (3, 0), (40, 20), (0, 10)])
def test_hough_peaks(M, num_peaks, thr):
    harris, stereo, hough, synth = M
    mask, lines, circles = synth.hough_mask(180, 260, n_lines=8, radii=())
    acc = orc.hough_lines(mask, 1, 1)
    exp = orc.hough_peaks(acc, num_peaks, thr)
    got = hough.findLocalMaxima(dev(acc), num_peaks, thr)
    assert np.array_equal(host(got).astype(np.uint32), exp)
    assert np.array_equal(hough.findLocalMaxima(acc, num_peaks, thr), exp)


@pytest.mark.parametrize("num_peaks", [20, 64, 65, 300])
def test_hough_peaks_many_candidates(M, num_peaks):
    """> 4096 candidates (keys not cached in LDS) through the single-launch selection
    (num_peaks <= 64) and through the launch-per-round path (> 64); many equal votes."""
    harris, stereo, hough, synth = M
    rng = np.random.default_rng(num_peaks)
    acc = rng.integers(0, 50, (300, 400)).astype(np.int32)
    exp = orc.hough_peaks(acc, num_peaks, 10)
    assert len(exp) == num_peaks
    got = hough.findLocalMaxima(dev(acc), num_peaks, 10)
    assert np.array_equal(host(got).astype(np.uint32), exp)


@pytest.mark.parametrize("shape,thr", [((20, 30), 30), ((40, 60), 30), ((120, 200), 38), ((200, 300), 40)])  # 140 / 519 / 3708 / > 4096 candidates
@pytest.mark.parametrize("num_peaks", [1, 10, 64])
def test_hough_peaks_two_stage_selection(M, shape, thr, num_peaks):
    """Candidate lists of a few dozen, a few hundred and a few thousand entries (<= 4096: every wave's own top K, then
    the K rounds over the survivors) with many equal votes: the survivors of different waves interleave in the answer,
    ties go to the smaller index, and a list shorter than num_peaks ends early."""
    harris, stereo, hough, synth = M
    rng = np.random.default_rng(shape[0] * 7 + num_peaks)
    acc = rng.integers(0, 48, shape).astype(np.int32)
    exp = orc.hough_peaks(acc, num_peaks, thr)
    got = hough.findLocalMaxima(dev(acc), num_peaks, thr)
    assert np.array_equal(host(got).astype(np.uint32), exp)
    # the same through the device-side count (no read-back)
    peaks, cnt = hough.findLocalMaxima(dev(acc), num_peaks, thr, lazy=True)
    assert int(cnt.item()) == len(exp) and np.array_equal(host(peaks)[:len(exp)].astype(np.uint32), exp)


def test_hough_peaks_ties_are_stable(M):
    harris, stereo, hough, synth = M
    acc = np.zeros((40, 50), np.int32)
    acc[5, 7] = acc[30, 2] = acc[12, 40] = 9   # equal votes -> row-major order
    acc[20, 20] = 11
    exp = orc.hough_peaks(acc, 10, 5)
    got = hough.findLocalMaxima(dev(acc), 10, 5)
    assert np.array_equal(host(got).astype(np.uint32), exp)
    assert exp.tolist()[0] == [20, 20] and exp.tolist()[1:4] == [[5, 7], [12, 40], [30, 2]]


def test_hough_1080p_peaks_find_the_drawn_lines(M):
    harris, stereo, hough, synth = M
    mask, lines, circles = synth.hough_mask(1080, 1920)
    acc = hough.houghLinesAccumulate(dev(mask), 1, 1)
    assert int(acc.sum().item()) == int((mask > 0).sum()) * 180  # every vote lands
    peaks = host(hough.findLocalMaxima(acc, 60, 300))
    diag = int(np.ceil(np.hypot(1080, 1920)))
    found = {(int(round(p[0] - diag)), int(p[1]) - 90) for p in peaks}
    for rho, theta in lines:
        assert any(abs(fr - rho) <= 2 and abs(ft - theta) <= 1 for fr, ft in found), (rho, theta)


def test_c5_4k_harris_keypoints_lk(M):
    """BASELINE config C5 at full size (3840x2160): Harris -> corner list -> keypoint angles -> 5-level
    LK sampled at the corners.  Harris response / corner list / flow are compared bit-for-bit with
    the oracle at this size; the flow at the corners must be the known translation."""
    harris, stereo, hough, synth = M
    from introtocomputervision_amd import lk
    rows, cols = 2160, 3840
    tex = synth.smooth_noise(0x5EED0004, rows, cols)
    chk = synth.checkerboard(rows, cols, square=40)
    prev = np.round(tex * (chk / 192.0)).astype(np.float32)      # corners + texture, integer valued
    nxt = np.ascontiguousarray(np.roll(prev, shift=(-2, 3), axis=(0, 1)))
    dp, dn = dev(prev), dev(nxt)
    gx, gy = harris.getGradients(dp, 3)
    R = harris.getCornerResponse(gx, gy, 5, 1.5, 0.04)
    corners, locs = harris.refineCorners(R, 5e8, 5, capacity=1 << 20)
    egx, egy = orc.sobel(prev, 3, 1.0)
    eR = orc.harris_response(egx, egy, 5, 1.5, 0.04)
    _, el = orc.harris_refine(eR, 5e8, 5)
    assert np.array_equal(host(R), eR)
    assert np.array_equal(host(locs), el) and len(el) > 1000
    kp = host(harris.getKeypoints(gx, gy, locs, 10))
    ekp = orc.sift_keypoints(egx, egy, el, 10)
    assert np.array_equal(kp[:, :3], ekp[:, :3]) and np.allclose(kp[:, 3], ekp[:, 3], atol=1e-3, rtol=0)
    u, v = lk.calcOpticalFlowPyr(dp, dn, 15, 5)
    eu, ev = orc.lk_flow_pyr(prev, nxt, 15, 5)
    assert np.array_equal(host(u), eu) and np.array_equal(host(v), ev)
    inner = el[(el[:, 0] > 100) & (el[:, 0] < rows - 100) & (el[:, 1] > 100) & (el[:, 1] < cols - 100)]
    fu, fv = eu[inner[:, 0], inner[:, 1]], ev[inner[:, 0], inner[:, 1]]
    # the checker edges bias LK at the corners themselves (the oracle gives (2.99, -2.55) here)
    assert abs(np.median(fu) - 3.0) < 0.3 and abs(np.median(fv) + 2.0) < 0.75


def test_pitched_views_all_families(M):
    """cv::Mat ROIs are non-continuous: every entry point takes a byte stride.  Inputs here are column
    slices of wider buffers (row pitch != cols * elem)."""
    harris, stereo, hough, synth = M
    rows, cols = 70, 90
    big = dev(np.zeros((rows, cols + 38), np.float32))
    img = synth.checkerboard(rows, cols, 10, seed=4)
    view = big[:, 19:19 + cols]
    view.copy_(dev(img))
    assert not view.is_contiguous()
    egx, egy = orc.sobel(img, 3, 1.0)
    gx, gy = harris.getGradients(view, 3)
    assert np.array_equal(host(gx), egx) and np.array_equal(host(gy), egy)
    bigx, bigy = torch.zeros_like(big), torch.zeros_like(big)
    vx, vy = bigx[:, 5:5 + cols], bigy[:, 7:7 + cols]
    vx.copy_(gx), vy.copy_(gy)
    assert np.array_equal(host(harris.getCornerResponse(vx, vy, 5, 1.5, 0.04)),
                          orc.harris_response(egx, egy, 5, 1.5, 0.04))
    left, right, _ = synth.stereo_pair(9, rows, cols)
    bl, br = torch.zeros_like(big), torch.zeros_like(big)
    vl, vr = bl[:, 3:3 + cols], br[:, 30:30 + cols]
    vl.copy_(dev(left)), vr.copy_(dev(right))
    assert np.array_equal(host(stereo.disparitySSD(vl, vr, 3, -20, 0)), orc.disparity_ssd(left, right, 3, -20, 0))
    mask, _, _ = synth.hough_mask(rows, cols, n_lines=3, radii=(10,))
    bm = dev(np.zeros((rows, cols + 21), np.uint8))
    vm = bm[:, 11:11 + cols]
    vm.copy_(dev(mask))
    assert np.array_equal(host(hough.houghLinesAccumulate(vm, 1, 1)), orc.hough_lines(mask, 1, 1))
    assert np.array_equal(host(hough.houghCirclesAccumulate(vm, 10)), orc.hough_circles(mask, 10))


@pytest.mark.parametrize("rows,cols", [(480, 640), (64, 64), (33, 129), (97, 61), (1, 70), (70, 1), (2, 2), (131, 200), (2160, 3840), (300, 1283)])
@pytest.mark.parametrize("win,cpu", [(5, False), (3, False), (7, True), (5, True)])
def test_harris_corners_chain_is_the_three_calls(M, rows, cols, win, cpu):
    """micv_harris_corners_dev (r05): image -> [gradients] -> R -> ordered list, the Sobel formed inside the response
    kernel's LDS tile (ps4_cpp/src/Solution.cpp:77-124 as one call).  Every output -- gradients, R, the sparse map, the
    list -- equals the three separate calls bit for bit (NaN-poisoned outputs first), on sizes around / below the tile,
    borders, an unaligned width and 4K; C1's size also against the oracle."""
    harris, stereo, hough, synth = M
    img = synth.smooth_noise(11 + rows, rows, cols) * np.float32(37.0) if rows * cols < 3000000 else synth.checkerboard(rows, cols, square=40, seed=3)
    d = dev(img)
    thr = 1e6
    gx, gy = harris.getGradients(d, 3)
    R = harris.getCornerResponse(gx, gy, win, 1.5, 0.04, cpu_arithmetic=cpu)
    corners, locs = harris.refineCorners(R, thr, 5)
    out = harris.cornersFromImage(d, 3, win, 1.5, 0.04, thr, 5, cpu_arithmetic=cpu, want_gradients=True, want_response=True, want_corners=True)
    assert host(out["gx"]).tobytes() == host(gx).tobytes() and host(out["gy"]).tobytes() == host(gy).tobytes()
    assert host(out["response"]).tobytes() == host(R).tobytes()
    assert host(out["corners"]).tobytes() == host(corners).tobytes()
    assert np.array_equal(host(out["locs"]), host(locs))
    # nothing but the list: R lives in context scratch, no gradient / map stores
    lean = harris.cornersFromImage(d, 3, win, 1.5, 0.04, thr, 5, cpu_arithmetic=cpu, want_gradients=False)
    assert np.array_equal(host(lean["locs"]), host(locs)) and set(lean) == {"locs"}
    if rows * cols <= 480 * 640:
        egx, egy = orc.sobel(img, 3, 1.0)
        eR = orc.harris_response_ex(egx, egy, win, 1.5, 0.04, orc.HARRIS_CPU) if cpu else orc.harris_response(egx, egy, win, 1.5, 0.04)
        assert np.array_equal(host(out["response"]), eR, equal_nan=True)
        assert np.array_equal(host(out["locs"]), orc.harris_refine(eR, thr, 5)[1])


def test_harris_corners_chain_other_sizes_host_form_and_views(M):
    """Sobel 5 / window 9 take the three launches inside the same call; the host form (one upload) and a pitched device
    view give the same list."""
    harris, stereo, hough, synth = M
    img = synth.checkerboard(240, 320, square=40, seed=5)
    for ks, win in ((5, 5), (3, 9), (7, 3)):
        gx, gy = harris.getGradients(dev(img), ks)
        R = harris.getCornerResponse(gx, gy, win, 1.5, 0.04)
        _, locs = harris.refineCorners(R, 5e8, 5)
        out = harris.cornersFromImage(dev(img), ks, win, 1.5, 0.04, 5e8, 5, want_response=True)
        assert host(out["response"]).tobytes() == host(R).tobytes() and np.array_equal(host(out["locs"]), host(locs))
        assert host(out["gx"]).tobytes() == host(gx).tobytes()
    gx, gy = harris.getGradients(dev(img), 3)
    R = harris.getCornerResponse(gx, gy, 5, 1.5, 0.04)
    _, locs = harris.refineCorners(R, 5e8, 5)
    h = harris.cornersFromImage(img, 3, 5, 1.5, 0.04, 5e8, 5, want_response=True, want_corners=True)
    assert np.array_equal(h["locs"], host(locs)) and h["response"].tobytes() == host(R).tobytes() and len(h["locs"]) > 10
    big = torch.zeros((250, 336), device="cuda")
    view = big[5:245, 8:328]
    view.copy_(dev(img))
    v = harris.cornersFromImage(view, 3, 5, 1.5, 0.04, 5e8, 5, want_gradients=False)
    assert np.array_equal(host(v["locs"]), host(locs))
    with pytest.raises(Exception):
        harris.cornersFromImage(dev(img), 4, 5, 1.5, 0.04, 5e8, 5)


@pytest.mark.parametrize("win", [3, 5, 7])
@pytest.mark.parametrize("rows", [67, 69, 98, 100, 131])
def test_harris_image_tiles_read_nothing_below_the_image(M, win, rows):
    """the fused Sobel -> response tiles load their rows in jobs of three, so a tile could count as interior
    while its last job reached one or two rows past the image (rows % 16 in 3..5, rows % 32 in 2..4).  The image here
    is the LAST rows of its allocation (torch's caching allocator rounds the block, so the read cannot be made to fault:
    the test pins the result -- identical to the three separate calls -- and the predicate is what is LOADED, harris.hip)."""
    harris, stereo, hough, synth = M
    cols = 200
    img = synth.checkerboard(rows + 40, cols, square=20, seed=rows * 10 + win)[:rows]
    buf = torch.full((rows + 64, cols), float("nan"), device="cuda")  # whatever lies below must not matter either
    view = buf[:rows]
    view.copy_(dev(img))
    last = torch.full((rows + 64, cols), float("nan"), device="cuda")[64:]  # the image ends where the allocation does
    last.copy_(dev(img))
    gx, gy = harris.getGradients(dev(img), 3)
    R = harris.getCornerResponse(gx, gy, win, 1.5, 0.04)
    for d in (view, last):
        out = harris.cornersFromImage(d, 3, win, 1.5, 0.04, 5e8, 5, want_response=True)
        assert host(out["response"]).tobytes() == host(R).tobytes()
