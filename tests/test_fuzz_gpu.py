"""Differential fuzz of the device entry points against the oracle: hypothesis draws
shapes, row pitches, windows, pyramid depths, flags and seeds; every draw is compared bit for bit
(disparities / votes / corner lists: exactly; float fields: bit patterns, NaN == NaN).  Sizes are small
(the oracle is a scalar C port) and the example counts are set so the whole file runs in about half a
minute on the GPU box; `derandomize=True` keeps the driver's round-end run reproducible."""
import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
hyp = pytest.importorskip("hypothesis")
from hypothesis import HealthCheck, given, settings  # noqa: E402
from hypothesis import strategies as st  # noqa: E402

import os  # noqa: E402

# MICV_FUZZ_SCALE=n: n times the examples, drawn at random instead of the fixed sequence (a soak run by hand)
SCALE = max(1, int(os.environ.get("MICV_FUZZ_SCALE", "1")))
COMMON = dict(deadline=None, derandomize=SCALE == 1, suppress_health_check=list(HealthCheck), print_blob=True)


def dev(a, pad=0):
    """Device copy of a 2-D array; pad > 0 gives it a row pitch of cols + pad elements."""
    a = np.ascontiguousarray(a)
    if pad == 0:
        return torch.from_numpy(a).cuda()
    wide = torch.full((a.shape[0], a.shape[1] + pad), 7, dtype=torch.from_numpy(a).dtype, device="cuda")
    wide[:, :a.shape[1]] = torch.from_numpy(a).cuda()
    return wide[:, :a.shape[1]]


def host(t):
    return t.cpu().numpy()


def same(a, b):
    """True when equal (float32: equal bit patterns, NaN == NaN); otherwise raises with the first cells."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    if a.dtype != np.float32:
        bad = a != b
    else:
        na, nb = np.isnan(a), np.isnan(b)
        bad = (na != nb) | (~na & ~nb & (a.view(np.uint32) != b.view(np.uint32)))
    if bad.any():
        idx = np.argwhere(bad)
        raise AssertionError(f"{len(idx)} of {a.size} cells differ, first {idx[:5].tolist()}: "
                             f"got {a[bad][:5].tolist()}, want {b[bad][:5].tolist()}")
    return True


def image(seed, rows, cols, kind):
    rng = np.random.default_rng(seed)
    if kind == 0:
        from introtocomputervision_amd import synth
        return synth.smooth_noise(seed, rows, cols)
    if kind == 1:
        return (rng.random((rows, cols)) * 255).astype(np.float32)
    if kind == 2:  # mostly flat with a few textured blocks: det < tau regions
        a = np.full((rows, cols), 50.0, np.float32)
        a[rows // 4:rows // 2, cols // 4:cols // 2] = (rng.random((rows // 2 - rows // 4, cols // 2 - cols // 4)) * 200)
        return a
    return (rng.standard_normal((rows, cols)) * 1e3).astype(np.float32)


shape = st.tuples(st.integers(1, 90), st.integers(1, 140))
pad = st.sampled_from([0, 0, 1, 3, 4, 64])
seed = st.integers(0, 2 ** 31 - 1)
kind = st.integers(0, 3)


@settings(max_examples=300 * SCALE, **COMMON)
@given(shape, pad, seed, kind, st.sampled_from([1, 3, 5, 7, 11, 15, 21, 23, 43]), st.integers(1, 5), st.booleans())
def test_fuzz_lk(shape, pad, seed, kind, win, levels, shift):
    from introtocomputervision_amd import lk
    rows, cols = shape
    levels = max(1, min(levels, int(np.log2(max(1, min(rows, cols)))) + 1))
    prev = image(seed, rows, cols, kind)
    nxt = np.roll(prev, (1, -2), (0, 1)) if shift else image(seed + 1, rows, cols, kind)
    eu, ev = orc.lk_flow_pyr(prev, nxt, win, levels)
    gu, gv = lk.calcOpticalFlowPyr(dev(prev, pad), dev(nxt, pad), winSize=win, levels=levels)
    assert same(host(gu), eu) and same(host(gv), ev), (rows, cols, pad, win, levels)
    if levels == 1:
        # the single-level function returns the solve's own values (a -0 stays -0; the pyramid adds it to a +0 base)
        su, sv = lk.calcOpticalFlow(dev(prev, pad), dev(nxt, pad), winSize=win)
        e1u, e1v = orc.lk_flow(prev, nxt, win)
        assert same(host(su), e1u) and same(host(sv), e1v)


@settings(max_examples=120 * SCALE, **COMMON)
@given(st.tuples(st.integers(1, 70), st.integers(1, 420)), pad, seed, kind, st.sampled_from([5, 9, 13, 23, 27, 43, 63]),
       st.sampled_from([3, 3, 2, 0]), st.booleans())
def test_fuzz_lk_generic_forms(shape, pad, seed, kind, win, form, shift):
    """The generic level in its two-launch (3; window 43: unrolled, LDS-DMA on interior 128-column tiles) and
    four-launch (2) forms and with the size-dependent default (0): widths past 256 give the row pass more than one
    tile and the column pass interior tiles."""
    from introtocomputervision_amd import lk, _capi
    rows, cols = shape
    prev = image(seed, rows, cols, kind)
    nxt = np.roll(prev, (1, -2), (0, 1)) if shift else image(seed + 1, rows, cols, kind)
    ctx = _capi.Context(0)
    ctx.set_option(_capi.OPT_LK_FORCE_GENERIC, form)
    su, sv = lk.calcOpticalFlow(dev(prev, pad), dev(nxt, pad), winSize=win, ctx=ctx)
    eu, ev = orc.lk_flow(prev, nxt, win)
    assert same(host(su), eu) and same(host(sv), ev), (rows, cols, pad, win, form)


@settings(max_examples=200 * SCALE, **COMMON)
@given(shape, pad, seed, st.floats(0.0, 40.0), st.booleans())
def test_fuzz_warp_pyr_resize(shape, pad, seed, amp, wild):
    from introtocomputervision_amd import lk, pyr
    rows, cols = shape
    rng = np.random.default_rng(seed)
    src = image(seed, rows, cols, 1)
    du = (rng.standard_normal((rows, cols)) * amp).astype(np.float32)
    dv = (rng.standard_normal((rows, cols)) * amp).astype(np.float32)
    if wild:
        bad = np.array([np.nan, np.inf, -np.inf, 1e12, -1e12, 2.0 ** 26, 6.7e7, -6.7e7], np.float32)
        idx = rng.integers(0, rows * cols, 6)
        du.flat[idx[:3]] = rng.choice(bad, 3)
        dv.flat[idx[3:]] = rng.choice(bad, 3)
    assert same(host(lk.warp(dev(src, pad), dev(du, pad), dev(dv, pad))), orc.lk_warp(src, du, dv))
    assert same(host(pyr.pyrUp(dev(src, pad))), orc.pyr_up(src))
    if rows >= 2 and cols >= 2:
        assert same(host(pyr.pyrDown(dev(src, pad))), orc.pyr_down(src))
    dr, dc = int(rng.integers(1, 2 * rows + 2)), int(rng.integers(1, 2 * cols + 2))
    assert same(host(pyr.resizeLinear(dev(src, pad), dr, dc)), orc.resize_linear(src, dr, dc))
    lv = max(1, min(4, int(np.log2(min(rows, cols))) + 1))
    for g, e in zip(pyr.makeGaussianPyramid(dev(src, pad), lv), orc.gaussian_pyramid(src, lv)):
        assert same(host(g), e)


@settings(max_examples=200 * SCALE, **COMMON)
@given(shape, pad, seed, kind, st.sampled_from([1, 3, 5, 7]), st.sampled_from([3, 5, 7, 9]),
       st.floats(0.5, 3.0), st.integers(1, 9))
def test_fuzz_harris(shape, pad, seed, kind, ksize, window, sigma, min_dist):
    from introtocomputervision_amd import harris
    rows, cols = shape
    img = image(seed, rows, cols, kind % 3)
    gx, gy = harris.getGradients(dev(img, pad), ksize)
    egx, egy = orc.sobel(img, ksize, 1.0)
    assert same(host(gx), egx) and same(host(gy), egy)
    resp = harris.getCornerResponse(gx, gy, window, sigma, 0.04)
    eresp = orc.harris_response(egx, egy, window, sigma, 0.04)
    assert same(host(resp), eresp)
    # harris::cpu's own arithmetic (MICV_HARRIS_CPU) on the same gradients
    assert same(host(harris.getCornerResponse(gx, gy, window, sigma, 0.04, cpu_arithmetic=True)),
                orc.harris_response_ex(egx, egy, window, sigma, 0.04, orc.HARRIS_CPU))
    thr = float(np.percentile(eresp, 90)) if eresp.size > 4 else 0.0
    corners, locs = harris.refineCorners(resp, thr, min_dist)
    ecorners, elocs = orc.harris_refine(eresp, thr, min_dist)
    assert same(host(corners), ecorners) and np.array_equal(host(locs), elocs)


@settings(max_examples=300 * SCALE, **COMMON)
@given(st.tuples(st.integers(1, 100), st.integers(1, 150)), pad, seed, st.integers(0, 12), st.integers(-40, 20),
       st.integers(0, 40), st.sampled_from([0, 1, 2, 3, 8, 9, 11]), st.booleans(), st.booleans())
def test_fuzz_stereo(shape, pad, seed, rad, min_d, span, flags, ncc, integer):
    from introtocomputervision_amd import stereo
    rows, cols = shape
    if rad == 0 and (flags & 1):
        flags &= ~1
    max_d = min(127, min_d + span)
    rng = np.random.default_rng(seed)
    left = (rng.random((rows, cols)) * 40).astype(np.float32)
    right = np.roll(left, int(rng.integers(-8, 8)), axis=1) + (rng.random((rows, cols)) * 2).astype(np.float32)
    if integer:
        left, right = np.floor(left), np.floor(right)
    if ncc:
        left, right = left + 1.0, right + 1.0
        exp = orc.disparity_ncorr(left, right, rad, min_d, max_d, flags & ~2)
        got = stereo.disparityNCorr(dev(left, pad), dev(right, pad), rad, min_d, max_d, flags & ~2)
    else:
        exp = orc.disparity_ssd(left, right, rad, min_d, max_d, flags)
        got = stereo.disparitySSD(dev(left, pad), dev(right, pad), rad, min_d, max_d, flags)
    assert np.array_equal(host(got), exp), (rows, cols, rad, min_d, max_d, flags, ncc, int((host(got) != exp).sum()))


@settings(max_examples=150 * SCALE, **COMMON)
@given(st.tuples(st.integers(1, 70), st.integers(1, 260)), pad, seed, st.integers(1, 7), st.integers(-128, 100),
       st.integers(0, 255), st.sampled_from([0, 1, 2, 3, 4, 8, 11]), st.booleans(), st.sampled_from([256, 3, 2]))
def test_fuzz_stereo_exact_sum(shape, pad, seed, rad, min_d, span, flags, ncc, levels):
    """8-bit-valued images (disparitySSD: the exact-sum kernels; disparityNCorr: the float kernels) against the oracle:
    any size, stride, radius 1..7, up to 256 disparities, every flag; few grey levels make ties the rule."""
    from introtocomputervision_amd import stereo
    from introtocomputervision_amd._capi import Context, OPT_STEREO_EXACT
    rows, cols = shape
    max_d = min(127, min_d + span)
    if rad < 2:
        flags &= ~1
    rng = np.random.default_rng(seed)
    scale = 255 // (levels - 1) if levels <= 3 else 1
    left = (rng.integers(0, levels, (rows, cols)) * scale).astype(np.float32)
    right = np.roll(left, int(rng.integers(-8, 8)), axis=1)
    noisy = rng.random((rows, cols)) < 0.3
    right[noisy] = (rng.integers(0, levels, int(noisy.sum())) * scale).astype(np.float32)
    ctx = Context(0)
    ctx.set_option(OPT_STEREO_EXACT, 0)
    if ncc:
        flags &= ~(2 | 4)
        exp = orc.disparity_ncorr(left, right, rad, min_d, max_d, flags)
        got = stereo.disparityNCorr(dev(left, pad), dev(right, pad), rad, min_d, max_d, flags, ctx=ctx)
    elif flags & 4:
        flags = 4
        exp = orc.disparity_ssd_serial(left, right, rad, min_d, max_d)
        got = stereo.disparitySSD(dev(left, pad), dev(right, pad), rad, min_d, max_d, flags, ctx=ctx)
    else:
        exp = orc.disparity_ssd(left, right, rad, min_d, max_d, flags)
        got = stereo.disparitySSD(dev(left, pad), dev(right, pad), rad, min_d, max_d, flags, ctx=ctx)
    assert np.array_equal(host(got), exp), (rows, cols, rad, min_d, max_d, flags, ncc, levels, int((host(got) != exp).sum()))


@settings(max_examples=150 * SCALE, **COMMON)
@given(st.tuples(st.integers(2, 120), st.integers(2, 160)), pad, seed, st.floats(0.0, 0.2), st.integers(1, 3),
       st.integers(1, 5), st.integers(1, 30), st.integers(1, 20))
def test_fuzz_hough(shape, pad, seed, density, rho_bin, theta_bin, radius, num_peaks):
    from introtocomputervision_amd import hough
    rows, cols = shape
    rng = np.random.default_rng(seed)
    mask = ((rng.random((rows, cols)) < density) * 255).astype(np.uint8)
    acc = hough.houghLinesAccumulate(dev(mask, pad), rho_bin, theta_bin)
    eacc = orc.hough_lines(mask, rho_bin, theta_bin)
    assert np.array_equal(host(acc), eacc)
    circ = hough.houghCirclesAccumulate(dev(mask, pad), radius)
    ecirc = orc.hough_circles(mask, radius)
    assert np.array_equal(host(circ), ecirc)
    thr = int(max(1, np.percentile(ecirc, 95)))
    assert np.array_equal(host(hough.findLocalMaxima(circ, num_peaks, thr)).astype(np.uint32),
                          orc.hough_peaks(ecirc, num_peaks, thr))
    thr = int(max(1, eacc.max() // 2))
    assert np.array_equal(host(hough.findLocalMaxima(acc, num_peaks, thr)).astype(np.uint32),
                          orc.hough_peaks(eacc, num_peaks, thr))


@settings(max_examples=60 * SCALE, **COMMON)
@given(st.tuples(st.integers(3, 90), st.integers(3, 120)), pad, seed,
       st.one_of(st.integers(1, 40), st.integers(1, 40), st.integers(1, 40), st.sampled_from([1799, 1801, 2300, 6143, 6145])),
       st.sampled_from([0, 0, 1, 2, 3]), st.floats(-8.0, 8.0))
def test_fuzz_sift_descriptors(shape, pad, seed, nkp, poison, log_scale):
    """Descriptor windows over random gradient fields of any magnitude (2^-8 .. 2^8 of 8-bit gradients), keypoints in
    and out of the image with sizes from a pixel to more than the image, and poisoned fields: a NaN, an infinity, a
    flat block -- the corner the NaN-share fix of r03 came from."""
    import torch
    from introtocomputervision_amd import harris
    rows, cols = shape
    rng = np.random.default_rng(seed)
    scale = np.float32(2.0 ** log_scale)
    gx = (rng.standard_normal((rows, cols)) * 60).astype(np.float32) * scale
    gy = (rng.standard_normal((rows, cols)) * 60).astype(np.float32) * scale
    if poison == 1:
        gx[rng.integers(0, rows), rng.integers(0, cols)] = np.nan
    elif poison == 2:
        gy[rng.integers(0, rows), rng.integers(0, cols)] = np.float32(np.inf) * (1 if seed & 1 else -1)
    elif poison == 3:
        gx[: rows // 2, : cols // 2] = 0
        gy[: rows // 2, : cols // 2] = 0
    # (long lists -- two waves / one wave per keypoint instead of four -- keep to small windows: the oracle is scalar)
    sizes = [0.5, 1.5, 8 / 3, 4, 10, 40] if nkp <= 40 else [0.5, 1.5, 8 / 3, 4]
    kps = np.stack([rng.uniform(-10, cols + 10, nkp), rng.uniform(-10, rows + 10, nkp),
                    rng.choice(sizes, nkp), rng.uniform(-400, 800, nkp)], 1).astype(np.float32)
    exp = orc.sift_descriptors(gx, gy, kps)
    got = harris.computeDescriptors(dev(gx, pad), dev(gy, pad), torch.from_numpy(kps).cuda())
    assert same(host(got), exp), (rows, cols, pad, nkp, poison)


def poison(a, seed, how):
    """A copy of `a` with a few non-finite / extreme cells."""
    rng = np.random.default_rng(seed ^ 0xBAD)
    a = a.copy()
    bad = {1: [np.nan], 2: [np.inf, -np.inf], 3: [np.nan, np.inf, -np.inf, 3e38, -3e38, 1e-38, -0.0]}[how]
    idx = rng.integers(0, a.size, 1 + seed % 4)
    a.flat[idx] = rng.choice(np.array(bad, np.float32), len(idx))
    return a


@settings(max_examples=120 * SCALE, **COMMON)
@given(st.tuples(st.integers(2, 80), st.integers(2, 150)), pad, seed, st.sampled_from([1, 2, 3]), st.sampled_from([3, 5, 7]),
       st.sampled_from([3, 5, 7, 9]), st.integers(0, 6), st.sampled_from([7, 15, 21, 9]))
def test_fuzz_poisoned_images(shape, pad, seed, how, ksize, window, rad, win):
    """NaN, +-inf, near-overflow, subnormal and -0 cells in the INPUT images of the Harris chain, the window stereo and
    single-level LK: whatever the reference's arithmetic makes of them (NaN responses, indefinite conversions,
    comparisons that are false) must come out of the kernels the same way (r03: a NaN gradient exposed a
    difference in the descriptor kernel's fixed-point conversion)."""
    from introtocomputervision_amd import harris, stereo, lk
    rows, cols = shape
    img = poison(image(seed, rows, cols, 1), seed, how)
    gx, gy = harris.getGradients(dev(img, pad), ksize)
    egx, egy = orc.sobel(img, ksize, 1.0)
    assert same(host(gx), egx) and same(host(gy), egy)
    resp = harris.getCornerResponse(gx, gy, window, 1.2, 0.04)
    eresp = orc.harris_response(egx, egy, window, 1.2, 0.04)
    assert same(host(resp), eresp)
    ec, el = orc.harris_refine(eresp, 1e6, 3)
    c, l = harris.refineCorners(resp, 1e6, 3)
    assert same(host(c), ec) and np.array_equal(host(l), el)
    right = poison(np.roll(image(seed, rows, cols, 1), 3, axis=1), seed + 1, how)
    for ncc in (False, True):
        fo = orc.disparity_ncorr if ncc else orc.disparity_ssd
        fg = stereo.disparityNCorr if ncc else stereo.disparitySSD
        assert np.array_equal(host(fg(dev(img, pad), dev(right, pad), rad, -12, 3, 0)), fo(img, right, rad, -12, 3, 0)), (ncc, rad)
    eu, ev = orc.lk_flow(img, right, win)
    su, sv = lk.calcOpticalFlow(dev(img, pad), dev(right, pad), winSize=win)
    assert same(host(su), eu) and same(host(sv), ev), win


@settings(max_examples=120 * SCALE, **COMMON)
@given(st.tuples(st.integers(1, 150), st.integers(1, 200)), pad, seed, st.sampled_from([1, 3, 5, 9, 31]), st.floats(0.3, 6.0),
       st.integers(0, 120), st.integers(0, 250), st.integers(0, 2))
def test_fuzz_generate_edge(shape, pad, seed, gs, sigma, lo, hi, kind):
    """sol::generateEdge on random byte images: the tiled blur, the fused gradient / NMS / threshold kernel whose ballots
    are the bit planes, the wave-per-tile hysteresis and its round loop, at sizes around the 64 x 62 tile."""
    import ctypes as C
    from introtocomputervision_amd import hough
    rows, cols = shape
    rng = np.random.default_rng(seed)
    if kind == 0:
        img = rng.integers(0, 256, (rows, cols)).astype(np.uint8)
    elif kind == 1:  # smooth: long connected candidate chains
        from introtocomputervision_amd import synth
        img = synth.smooth_noise(seed, rows, cols).astype(np.uint8)
    else:  # flat with a few steps
        img = np.full((rows, cols), 90, np.uint8)
        img[rows // 3:, cols // 4:] = 140
        img[: rows // 2, : cols // 2] += rng.integers(0, 3, (rows // 2, cols // 2)).astype(np.uint8)
    fn = orc._sig("orc_generate_edge", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_double, C.c_double,
                                                 C.c_double, C.c_void_p, C.c_size_t])
    exp = np.empty((rows, cols), np.uint8)
    assert fn(img.ctypes.data, rows, cols, cols, gs, float(sigma), float(lo), float(hi), exp.ctypes.data, cols) == 0
    got = hough.generateEdge(dev(img, pad), gs, float(sigma), lo, hi)
    assert same(host(got), exp), (rows, cols, pad, gs, lo, hi)


@settings(max_examples=40 * SCALE, **COMMON)
@given(st.tuples(st.integers(40, 200), st.integers(16, 260)), seed, kind, st.sampled_from([7, 15, 15, 21]), st.integers(1, 4),
       st.integers(1, 6), st.integers(1, 3), st.integers(0, 2))
def test_fuzz_rowshard_virtual_and_level_options(shape, seed, kind, win, levels, world, batch, direct):
    """The C ABI's row-shard driver as virtual ranks (NaN-poisoned private memory) and the level-kernel options that
    change how a level is staged (levels read straight from level 0) against the plain call, on random shapes."""
    from introtocomputervision_amd import lk, shard, _capi
    rows, cols = shape
    levels = max(1, min(levels, int(np.log2(max(1, min(rows, cols)))) - 1))
    world = max(1, min(world, rows >> (levels - 1)))
    prev = np.stack([image(seed + i, rows, cols, kind) for i in range(batch)])
    nxt = np.stack([np.roll(p, (1, -2), (0, 1)) for p in prev])
    ctx = _capi.Context(0)
    dp, dn = torch.from_numpy(prev).cuda(), torch.from_numpy(nxt).cuda()
    ru, rv = lk.calcOpticalFlowPyrBatch(dp, dn, win, levels, ctx=ctx)
    eu, ev = orc.lk_flow_pyr(prev[0], nxt[0], win, levels)
    assert same(host(ru[0]), eu) and same(host(rv[0]), ev)
    u, v = shard.run_virtual_native(ctx, world, dp, dn, win, levels, poison=True)
    assert same(host(u), host(ru)) and same(host(v), host(rv)), (rows, cols, win, levels, world, batch)
    if direct:
        ctx.set_option(_capi.OPT_LK_DIRECT_LEVELS, direct)
        du, dv = lk.calcOpticalFlowPyrBatch(dp, dn, win, levels, ctx=ctx)
        assert same(host(du), host(ru)) and same(host(dv), host(rv)), (rows, cols, win, levels, direct)
        ctx.set_option(_capi.OPT_LK_DIRECT_LEVELS, 0)
    # the pyramid build carried by the level launches (every batch) / never: same bits as the default
    for carried in (1, -1):
        ctx.set_option(_capi.OPT_LK_BUILD_OVERLAP, carried)
        cu, cv = lk.calcOpticalFlowPyrBatch(dp, dn, win, levels, ctx=ctx)
        assert same(host(cu), host(ru)) and same(host(cv), host(rv)), (rows, cols, win, levels, batch, carried)


@settings(max_examples=25 * SCALE, **COMMON)
@given(st.tuples(st.integers(64, 420), st.integers(64, 560)), seed, kind, st.integers(2, 3), st.integers(1, 3), st.integers(1, 9),
       st.integers(1, 3))
def test_fuzz_split_and_strip_launches(shape, seed, kind, levels, batch, blocks, split):
    """r05: the two other divisions of a level launch -- MICV_OPT_LK_SPLIT (pre-pass + streaming sums, three forms) and
    MICV_OPT_LK_STRIP (the interior as streamed strips) -- on random even shapes, textures with NaN / inf / flat regions,
    against the tile launch and the oracle.  Shapes without an interior tile fall back to the tile launch inside the call."""
    from introtocomputervision_amd import lk, _capi
    rows, cols = shape[0] & ~3, shape[1] & ~3   # (doubling coarse flow on the finest level; 16-byte rows)
    prev = np.stack([image(seed + i, rows, cols, kind) for i in range(batch)])
    nxt = np.stack([np.roll(p, (2, -1), (0, 1)) for p in prev])
    ctx = _capi.Context(0)
    dp, dn = torch.from_numpy(prev).cuda(), torch.from_numpy(nxt).cuda()
    ru, rv = lk.calcOpticalFlowPyrBatch(dp, dn, 15, levels, ctx=ctx)
    eu, ev = orc.lk_flow_pyr(prev[0], nxt[0], 15, levels)
    assert same(host(ru[0]), eu) and same(host(rv[0]), ev)
    ctx.set_option(_capi.OPT_LK_STRIP, blocks)
    su, sv = lk.calcOpticalFlowPyrBatch(dp, dn, 15, levels, ctx=ctx)
    assert same(host(su), host(ru)) and same(host(sv), host(rv)), ("strip", rows, cols, levels, batch, blocks)
    ctx.set_option(_capi.OPT_LK_STRIP, 0)
    ctx.set_option(_capi.OPT_LK_SPLIT, split)
    pu, pv = lk.calcOpticalFlowPyrBatch(dp, dn, 15, levels, ctx=ctx)
    assert same(host(pu), host(ru)) and same(host(pv), host(rv)), ("split", rows, cols, levels, batch, split)


@settings(max_examples=60 * SCALE, **COMMON)
@given(shape, pad, seed, st.integers(0, 3), st.sampled_from([3, 5, 7, 9]), st.sampled_from([3, 3, 5]), st.integers(0, 8), st.booleans(),
       st.floats(0.0, 1.0))
def test_fuzz_harris_corners_chain(shape, pad, seed, kind, window, ksize, min_dist, cpu, thr_q):
    """micv_harris_corners_dev (r05) against the three separate calls on random shapes / pitches / textures (incl. NaN and
    inf), both arithmetics, thresholds from "everything" to "nothing": gradients, R, the sparse map and the list."""
    from introtocomputervision_amd import harris
    rows, cols = shape
    img = image(seed, rows, cols, kind).copy()
    if seed % 3 == 0 and img.size > 4:  # a few NaN / inf / huge pixels: the chain and the three calls must agree on them too
        rng = np.random.default_rng(seed)
        for val in (np.nan, np.inf, -np.inf, 3e38):
            img[rng.integers(0, rows), rng.integers(0, cols)] = val
    d = dev(img, pad)
    gx, gy = harris.getGradients(d, ksize)
    R = harris.getCornerResponse(gx, gy, window, 1.5, 0.04, cpu_arithmetic=cpu)
    finite = host(R)[np.isfinite(host(R))]
    thr = float(np.quantile(finite, thr_q)) if finite.size else 0.0
    corners, locs = harris.refineCorners(R, thr, min_dist)
    out = harris.cornersFromImage(d, ksize, window, 1.5, 0.04, thr, min_dist, cpu_arithmetic=cpu, want_response=True, want_corners=True)
    assert same(host(out["gx"]), host(gx)) and same(host(out["gy"]), host(gy)), (rows, cols, pad, ksize)
    assert same(host(out["response"]), host(R)), (rows, cols, pad, window, cpu)
    assert same(host(out["corners"]), host(corners)) and np.array_equal(host(out["locs"]), host(locs)), (rows, cols, thr, min_dist)


@settings(max_examples=60 * SCALE, **COMMON)
@given(shape, pad, seed, st.sampled_from([1, 3, 5, 9, 31]), st.sampled_from([1, 3, 7]), st.floats(0.4, 8.0), st.integers(0, 60), st.integers(0, 2))
def test_fuzz_mhi_frame_difference(shape, pad, seed, kw, kh, sigma, thr, kind):
    """mhi::frameDifference on bit planes (r05) on random byte frames: dense, sparse and blocky masks, any blur size pair,
    widths around the 64-column words, images smaller than the 7x7 element -- byte-exact against the oracle."""
    from introtocomputervision_amd import mhi
    rows, cols = shape
    rng = np.random.default_rng(seed)
    f1 = rng.integers(0, 256, (rows, cols)).astype(np.uint8)
    if kind == 0:
        f2 = rng.integers(0, 256, (rows, cols)).astype(np.uint8)
    elif kind == 1:  # sparse motion
        f2 = f1.copy()
        m = rng.random((rows, cols)) < 0.08
        f2[m] = np.clip(f2[m].astype(np.int32) + 90, 0, 255).astype(np.uint8)
    else:  # a moving block
        f2 = f1.copy()
        f2[rows // 4: rows // 4 + max(1, rows // 3), cols // 5: cols // 5 + max(1, cols // 2)] = 255
    exp = orc.mhi_frame_difference(f1, f2, thr, (kw, kh), sigma)
    got = mhi.frameDifference(dev(f1, pad), dev(f2, pad), thr, (kw, kh), sigma)
    assert same(host(got), exp), (rows, cols, pad, kw, kh, thr, kind)


@settings(max_examples=40 * SCALE, **COMMON)
@given(st.tuples(st.integers(33, 560), st.integers(33, 700)), seed, kind, st.sampled_from([7, 11, 15, 15, 21]), st.integers(1, 4), st.integers(1, 9),
       st.lists(st.tuples(st.sampled_from(["OPT_LK_STREAM_GROUPS", "OPT_LK_NARROW_TILES", "OPT_LK_CHAIN", "OPT_LK_SHORT_TILES", "OPT_LK_STREAM",
                                            "OPT_LK_TALL_TILES", "OPT_LK_DIRECT_LEVELS", "OPT_LK_BUILD_OVERLAP", "OPT_LK_SPLIT", "OPT_LK_STRIP",
                                            "OPT_LK_FORCE_GENERIC"]), st.integers(0, 5)), min_size=1, max_size=4))
def test_fuzz_lk_option_combinations(shape, seed, kind, win, levels, batch, opts):
    """"None of these changes a result" (mi_cv.h) for COMBINATIONS of the execution options on random shapes and batches:
    the deterministic tests take the options one at a time."""
    from introtocomputervision_amd import lk, _capi
    rows, cols = shape
    levels = max(1, min(levels, int(np.log2(max(1, min(rows, cols)))) - 1))
    legal = {"OPT_LK_STREAM_GROUPS": [0, 1, 2, 3, 4, 2], "OPT_LK_NARROW_TILES": [0, 1, 1, 0, 1, 0], "OPT_LK_CHAIN": [0, 1, 2, 4, -1, 32],
             "OPT_LK_SHORT_TILES": [0, -1, 64, 2000, 100000, 0], "OPT_LK_STREAM": [0, 1, 1, 0, 1, 0], "OPT_LK_TALL_TILES": [0, 1, 2, 3, -1, 1],
             "OPT_LK_DIRECT_LEVELS": [0, 1, 2, 3, 1, 0], "OPT_LK_BUILD_OVERLAP": [0, 1, -1, 1, -1, 0], "OPT_LK_SPLIT": [0, 1, 2, 3, 1, 2],
             "OPT_LK_STRIP": [0, 1, 3, 16, 8195, 2], "OPT_LK_FORCE_GENERIC": [0, 1, 2, 3, 0, 0]}
    prev = np.stack([image(seed + i, rows, cols, kind) for i in range(batch)])
    nxt = np.stack([np.roll(p, (1, -2), (0, 1)) for p in prev])
    dp, dn = torch.from_numpy(prev).cuda(), torch.from_numpy(nxt).cuda()
    ru, rv = lk.calcOpticalFlowPyrBatch(dp, dn, win, levels, ctx=_capi.Context(0))
    ctx = _capi.Context(0)
    for name, i in opts:
        ctx.set_option(getattr(_capi, name), legal[name][i])
    ou, ov = lk.calcOpticalFlowPyrBatch(dp, dn, win, levels, ctx=ctx)
    assert same(host(ou), host(ru)) and same(host(ov), host(rv)), (rows, cols, win, levels, batch, [(n, legal[n][i]) for n, i in opts])


@settings(max_examples=60 * SCALE, **COMMON)
@given(st.integers(1, 700), st.integers(2, 1500), st.one_of(st.integers(1, 160), st.sampled_from([4, 32, 64, 128, 132])), seed, st.integers(0, 2),
       st.floats(0.3, 1.0), pad, st.booleans())  # (k = 2 needs two train rows: the C ABI says so)
def test_fuzz_bf_knn2_and_ratio(nq, nt, dim, seed, kind, ratio, pad, shifted):
    """BFMatcher knn2 + ratio test (match.hip; its chunk staging is prefetched since r05) on random set sizes and
    dimensions around the 64 x 128 x 32 tile, with duplicated train rows (ties: the lower index wins) and clustered
    descriptors (near-ties): indices and distances bit-exact against the oracle."""
    import ctypes as C
    from introtocomputervision_amd import match
    vp, i32, i64, sz, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_double
    knn = orc._sig("orc_bf_knn2", None, [vp, i32, sz, vp, i32, sz, i32, vp, vp])
    ratf = orc._sig("orc_bf_ratio_filter", i64, [vp, vp, i32, f64, vp, vp, i64])
    rng = np.random.default_rng(seed)
    if kind == 0:
        t = rng.integers(0, 256, (nt, dim)).astype(np.float32)
    elif kind == 1:  # clustered: many near-ties
        t = (rng.integers(0, 4, (nt, dim)) * 64).astype(np.float32) + rng.integers(0, 2, (nt, dim)).astype(np.float32)
    else:
        t = rng.standard_normal((nt, dim)).astype(np.float32) * 100
    if nt > 3:
        t[nt // 2] = t[0]  # exact duplicate rows
    q = (t[rng.integers(0, nt, nq)] + (rng.integers(-3, 4, (nq, dim)).astype(np.float32) if kind < 2 else 0)).astype(np.float32)
    q = np.ascontiguousarray(q); t = np.ascontiguousarray(t)
    eidx = np.empty((nq, 2), np.int32); edist = np.empty((nq, 2), np.float32)
    knn(q.ctypes.data, nq, dim, t.ctypes.data, nt, dim, dim, eidx.ctypes.data, edist.ctypes.data)
    # row pitches and base addresses that are / are not multiples of 16 bytes: the staging loads by float4 only when it can
    dq, dt = dev(q, pad), dev(t, pad)
    if shifted:
        wide = torch.zeros((nt, dim + 5), device="cuda")
        wide[:, 1:1 + dim] = torch.from_numpy(t).cuda()
        dt = wide[:, 1:1 + dim]
    idx, dist = match.knnMatch2(dq, dt)
    assert np.array_equal(host(idx), eidx), (nq, nt, dim, kind)
    assert same(host(dist), edist), (nq, nt, dim, kind)
    em = np.empty((nq, 2), np.int32); ed = np.empty(nq, np.float32)
    n = ratf(eidx.ctypes.data, edist.ctypes.data, nq, float(ratio), em.ctypes.data, ed.ctypes.data, nq)
    m, d = match.ratioTest(idx, dist, float(ratio))
    assert np.array_equal(host(m), em[:n]) and same(host(d), ed[:n]), (nq, nt, dim, ratio)


@settings(max_examples=60 * SCALE, **COMMON)
@given(shape, seed, st.sampled_from([1, 3, 4]), st.booleans(), st.integers(1, 5))
def test_fuzz_to_gray_and_laplacian_pyramid(shape, seed, cn, as_float, levels):
    """Pyramids.cpp:9-15 (colour branch) and Solution.cpp:187-200 (Laplacian levels) on random shapes,
    channel counts and element types; the host and the device entry point of toGray both."""
    from introtocomputervision_amd import pyr
    rows, cols = shape
    rng = np.random.default_rng(seed)
    full = (rows, cols) if cn == 1 else (rows, cols, cn)
    img = rng.integers(0, 256, full, dtype=np.uint8)
    if as_float:
        img = (img.astype(np.float32) * np.float32(1.0 / 3) - np.float32(20)).astype(np.float32)
    want = orc.to_gray(img)
    assert same(pyr.toGray(img), want)
    grey = pyr.toGray(torch.from_numpy(img).cuda())
    assert same(host(grey), want)
    if cn == 3 and not as_float:
        assert same(host(pyr.rgb8ToGray(torch.from_numpy(img).cuda())), orc.rgb8_to_gray(img))
    levels = max(1, min(levels, int(np.log2(min(rows, cols))) + 1))
    g = orc.gaussian_pyramid(want, levels)
    for l, lap in enumerate(pyr.makeLaplacianPyramid(grey, levels)):
        e = g[l]
        if l < levels - 1:
            up = orc.pyr_up(g[l + 1])
            if up.shape[0] < e.shape[0] or up.shape[1] < e.shape[1]:  # odd sizes: Solution.cpp:194-196
                up = orc.resize_linear(up, *e.shape)
            e = e - up
        assert same(host(lap), e)


@settings(max_examples=40 * SCALE, **COMMON)
@given(st.tuples(st.integers(16, 130), st.integers(8, 200)), seed, st.integers(1, 8), st.integers(0, 7), st.integers(-30, 5),
       st.integers(0, 30), st.sampled_from([3, 5]), st.sampled_from([3, 5, 7]), st.integers(0, 6), st.floats(0.0, 0.15))
def test_fuzz_virtual_shards_of_stereo_harris_hough(shape, seed, world, rad, min_d, span, ksize, window, min_dist, density):
    """Row shards (band + static halo) of stereo, the Harris chain and the Hough vote on one GPU, as 1..8
    virtual ranks, against the unsharded device call; ragged cuts and bands thinner than the halo included."""
    from introtocomputervision_amd import harris, hough, shard_ops as so, stereo, synth
    from introtocomputervision_amd._capi import Context
    rows, cols = shape
    world = min(world, rows)
    ctx = Context(0)
    fns = so.gpu_fns(ctx)
    rng = np.random.default_rng(seed)
    left, right, _ = synth.stereo_pair(seed % 1000, rows, cols)
    dl, dr = dev(left), dev(right)
    max_d = min_d + span
    for fn, whole in ((fns.ssd(rad, min_d, max_d), stereo.disparitySSD(dl, dr, rad, min_d, max_d, ctx=ctx)),
                      (fns.ncorr(rad, min_d, max_d), stereo.disparityNCorr(dl, dr, rad, min_d, max_d, ctx=ctx))):
        parts = []
        for g in range(world):
            band, held = so.band_with_halo(rows, world, g, rad)
            parts.append(so.stereo_sharded(dl[held[0]:held[1]], dr[held[0]:held[1]], band, held, fn))
        assert torch.equal(torch.cat(parts), whole)

    img = image(seed, rows, cols, 1)
    di = dev(img)
    gx, gy = harris.getGradients(di, ksize, ctx=ctx)
    resp = harris.getCornerResponse(gx, gy, window, 1.5, 0.04, ctx=ctx)
    thr = float(np.quantile(host(resp), 0.9))
    corners, locs = harris.refineCorners(resp, thr, min_dist, ctx=ctx)
    halo = so.harris_halo(ksize, window, min_dist)
    rs, cs, ls = [], [], []
    for g in range(world):
        band, held = so.band_with_halo(rows, world, g, halo)
        r, c, l = so.harris_sharded(di[held[0]:held[1]], band, held, ksize, window, 1.5, 0.04, thr, min_dist,
                                    fns.grad, fns.response, fns.refine, so.LocalComm())
        rs.append(r), cs.append(c), ls.append(l)
    assert same(host(torch.cat(rs)), host(resp))
    assert torch.equal(torch.cat(cs), corners)
    assert torch.equal(torch.cat(ls), locs)

    mask = dev((rng.random((rows, cols)) < density).astype(np.uint8) * 255)
    whole = hough.houghLinesAccumulate(mask, 1, 1, ctx=ctx)
    radius = int(rng.integers(1, 12))
    whole_c = hough.houghCirclesAccumulate(mask, radius, ctx=ctx)
    total, total_c = torch.zeros_like(whole), torch.zeros_like(whole_c)
    for g in range(world):
        (a, b), _ = so.band_with_halo(rows, world, g, 0)
        total += so.hough_lines_sharded(mask[a:b], (a, b), rows, 1, 1, fns.hough_lines_band, so.LocalComm())
        total_c += so.hough_circles_sharded(mask[a:b], (a, b), rows, radius, fns.hough_circles_band, so.LocalComm())
    assert torch.equal(total, whole) and torch.equal(total_c, whole_c)


def pitched(a, pad):
    """A numpy view with a row pitch of cols + pad elements (what a cv::Mat ROI hands to the `_host` entry points)."""
    a = np.ascontiguousarray(a)
    if pad == 0:
        return a
    wide = np.full((a.shape[0], a.shape[1] + pad), 7, a.dtype)
    wide[:, :a.shape[1]] = a
    return wide[:, :a.shape[1]]


@settings(max_examples=80 * SCALE, **COMMON)
@given(st.tuples(st.integers(8, 70), st.integers(8, 110)), pad, seed, st.integers(0, 9))
def test_fuzz_host_entry_points_match_the_device_ones(shape, pad, seed, op):
    """Every `_host` entry point (host pointers in and out, staging and copies inside) against its `_dev` twin on the same
    inputs, with pitched rows as a cv::Mat region of interest has them: the same bits."""
    from introtocomputervision_amd import harris, hough, lk, mhi, pyr, stereo
    rows, cols = shape
    rng = np.random.default_rng(seed)
    a = image(seed, rows, cols, 1)
    b = np.roll(a, (1, -2), (0, 1)) + np.float32(0.5)
    A, B_ = pitched(a, pad), pitched(b, pad)
    if op == 0:
        win, levels = int(rng.choice([5, 7, 15, 21])), int(rng.integers(1, 4))
        for g, e in zip(lk.calcOpticalFlowPyr(A, B_, win, levels), lk.calcOpticalFlowPyr(dev(a), dev(b), win, levels)):
            assert same(g, host(e))
    elif op == 1:
        win = int(rng.choice([3, 7, 15, 23]))
        for g, e in zip(lk.calcOpticalFlow(A, B_, win), lk.calcOpticalFlow(dev(a), dev(b), win)):
            assert same(g, host(e))
    elif op == 2:
        du = (rng.standard_normal((rows, cols)) * 3).astype(np.float32)
        dv = (rng.standard_normal((rows, cols)) * 3).astype(np.float32)
        assert same(lk.warp(A, pitched(du, pad), pitched(dv, pad)), host(lk.warp(dev(a), dev(du), dev(dv))))
    elif op == 3:
        assert same(pyr.pyrDown(A), host(pyr.pyrDown(dev(a))))
        assert same(pyr.pyrUp(A), host(pyr.pyrUp(dev(a))))
        for g, e in zip(pyr.makeGaussianPyramid(A, 3), pyr.makeGaussianPyramid(dev(a), 3)):
            assert same(g, host(e))
    elif op == 4:
        k, w = int(rng.choice([3, 5])), int(rng.choice([3, 5, 7]))
        gx, gy = harris.getGradients(A, k)
        dgx, dgy = harris.getGradients(dev(a), k)
        assert same(gx, host(dgx)) and same(gy, host(dgy))
        R = harris.getCornerResponse(pitched(gx, pad), pitched(gy, pad), w, 1.5, 0.04)
        dR = harris.getCornerResponse(dgx, dgy, w, 1.5, 0.04)
        assert same(R, host(dR))
        thr = float(np.quantile(R, 0.9))
        (c, l), (dc, dl) = harris.refineCorners(pitched(R, pad), thr, 3), harris.refineCorners(dR, thr, 3)
        assert same(c, host(dc)) and np.array_equal(l, host(dl))
        kp, dkp = harris.getKeypoints(gx, gy, l, 4), harris.getKeypoints(dgx, dgy, dl, 4)
        assert same(np.asarray(kp), host(dkp) if torch.is_tensor(dkp) else np.asarray(dkp))
        if len(l):
            assert same(harris.computeDescriptors(gx, gy, np.asarray(kp)), host(harris.computeDescriptors(dgx, dgy, dkp)))
    elif op == 5:
        w = int(rng.choice([3, 5, 7, 9]))
        h, d = harris.cornersFromImage(A, 3, w, 1.5, 0.04, 1e6, 2), harris.cornersFromImage(dev(a), 3, w, 1.5, 0.04, 1e6, 2)
        assert np.array_equal(h["locs"], host(d["locs"])) and same(h["gx"], host(d["gx"])) and same(h["gy"], host(d["gy"]))
    elif op == 6:
        rad, mn = int(rng.integers(0, 6)), -int(rng.integers(0, 20))
        for fn in (stereo.disparitySSD, stereo.disparityNCorr):
            assert np.array_equal(fn(A, B_, rad, mn, mn + 12), host(fn(dev(a), dev(b), rad, mn, mn + 12)))
    elif op == 7:
        m = (rng.random((rows, cols)) < 0.05).astype(np.uint8) * 255
        M = pitched(m, pad)
        acc, dacc = hough.houghLinesAccumulate(M, 1, 1), hough.houghLinesAccumulate(dev(m), 1, 1)
        assert np.array_equal(acc, host(dacc))
        r = int(rng.integers(1, 9))
        assert np.array_equal(hough.houghCirclesAccumulate(M, r), host(hough.houghCirclesAccumulate(dev(m), r)))
        assert np.array_equal(hough.findLocalMaxima(acc, 8, 3), host(hough.findLocalMaxima(dacc, 8, 3)).astype(np.uint32))
    elif op == 8:
        img = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
        gs, lo = int(rng.choice([1, 3, 5])), float(rng.integers(0, 60))
        assert np.array_equal(hough.generateEdge(pitched(img, pad), gs, 1.2, lo, lo + 50), host(hough.generateEdge(dev(img), gs, 1.2, lo, lo + 50)))
    else:
        f1 = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
        f2 = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
        d = mhi.frameDifference(f1, f2, 20, (3, 3), 1.0)  # (the mhi wrappers take contiguous frames)
        dd = mhi.frameDifference(dev(f1), dev(f2), 20, (3, 3), 1.0)
        assert np.array_equal(d, host(dd))
        hist = rng.integers(0, 40, (rows, cols), dtype=np.uint8)
        assert np.array_equal(mhi.calcMotionHistory(hist.copy(), d, 30), host(mhi.calcMotionHistory(dev(hist), dd, 30)))


@settings(max_examples=40 * SCALE, **COMMON)
@given(st.tuples(st.integers(8, 130), st.integers(8, 300)), seed, kind, st.sampled_from([5, 9, 13, 23, 27, 43, 63]), st.integers(1, 4),
       st.integers(1, 5), st.sampled_from([0, 0, 2, 3]))
def test_fuzz_lk_generic_windows_batched(shape, seed, kind, win, levels, batch, form):
    """The generic chain (windows without a fused kernel), one launch per step for the whole batch: every pair of a random
    batch against the oracle, in the size-dependent default and in the forced four- and two-launch forms."""
    from introtocomputervision_amd import lk, _capi
    rows, cols = shape
    levels = max(1, min(levels, int(np.log2(min(rows, cols)))))
    prev = np.stack([image(seed + i, rows, cols, kind) for i in range(batch)])
    nxt = np.stack([np.roll(p, (1, -2), (0, 1)) for p in prev])
    ctx = _capi.Context(0)
    ctx.set_option(_capi.OPT_LK_FORCE_GENERIC, form)
    u, v = lk.calcOpticalFlowPyrBatch(torch.from_numpy(prev).cuda(), torch.from_numpy(nxt).cuda(), win, levels, ctx=ctx)
    for b in range(batch):
        eu, ev = orc.lk_flow_pyr(prev[b], nxt[b], win, levels)
        assert same(host(u[b]), eu) and same(host(v[b]), ev), (rows, cols, win, levels, batch, b, form)
