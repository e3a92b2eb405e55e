"""Parity of the HIP Lucas-Kanade / pyramid path (through the C ABI) against the CPU oracle.

Bar: BIT-EXACT (np.array_equal, so -0 == +0) -- tighter than the 1e-4 the north star asks
for, and the only way to be safe at the det < tau discontinuity (OpticalFlow.cpp:82,95).
"""
import os

import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def rand_img(rows, cols, seed):
    from introtocomputervision_amd import synth
    return synth.smooth_noise(seed, rows, cols)


@pytest.fixture(scope="module")
def mods():
    from introtocomputervision_amd import lk, pyr
    return lk, pyr


SIZES = [(48, 64), (37, 53), (96, 130), (135, 240), (17, 9), (64, 200)]


@pytest.mark.parametrize("rows,cols", SIZES)
def test_pyr_down_up(mods, rows, cols):
    lk, pyr = mods
    a = rand_img(rows, cols, 11)
    assert np.array_equal(host(pyr.pyrDown(dev(a))), orc.pyr_down(a))
    assert np.array_equal(host(pyr.pyrUp(dev(a))), orc.pyr_up(a))
    # host flavour
    assert np.array_equal(pyr.pyrDown(a), orc.pyr_down(a))
    assert np.array_equal(pyr.pyrUp(a), orc.pyr_up(a))


def test_gaussian_pyramid(mods):
    lk, pyr = mods
    a = rand_img(270, 300, 5)
    got = pyr.makeGaussianPyramid(dev(a), 5)
    exp = orc.gaussian_pyramid(a, 5)
    assert [tuple(g.shape) for g in got] == [e.shape for e in exp]
    for g, e in zip(got, exp):
        assert np.array_equal(host(g), e)
    got_h = pyr.makeGaussianPyramid(a, 4)
    for g, e in zip(got_h, exp):
        assert np.array_equal(g, e)


@pytest.mark.parametrize("shape,dshape", [((134, 240), (135, 240)), ((20, 30), (41, 61)), ((33, 17), (32, 17))])
def test_resize_linear(mods, shape, dshape):
    lk, pyr = mods
    a = rand_img(shape[0], shape[1], 3) * 0.37
    assert np.array_equal(host(pyr.resizeLinear(dev(a), *dshape)), orc.resize_linear(a, *dshape))


def test_rgb_to_gray(mods):
    lk, pyr = mods
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (45, 67, 3), dtype=np.uint8)
    assert np.array_equal(host(pyr.rgb8ToGray(dev(rgb))), orc.rgb8_to_gray(rgb))


@pytest.mark.parametrize("rows,cols", SIZES)
def test_warp(mods, rows, cols):
    lk, pyr = mods
    a = rand_img(rows, cols, 7)
    rng = np.random.default_rng(rows * cols)
    du = (rng.standard_normal((rows, cols)) * 3).astype(np.float32)
    dv = (rng.standard_normal((rows, cols)) * 3).astype(np.float32)
    du[0, 0] = 1e4  # far outside -> constant border
    dv[-1, -1] = -1e4
    exp = orc.lk_warp(a, du, dv)
    assert np.array_equal(host(lk.warp(dev(a), dev(du), dev(dv))), exp)
    assert np.array_equal(lk.warp(a, du, dv), exp)
    zero = np.zeros_like(a)
    assert np.array_equal(host(lk.warp(dev(a), dev(zero), dev(zero))), a)  # identity


@pytest.mark.parametrize("win", [15, 7, 5, 21, 43])
@pytest.mark.parametrize("rows,cols", [(48, 64), (37, 53), (96, 130), (70, 200)])
def test_lk_single_level(mods, rows, cols, win):
    lk, pyr = mods
    from introtocomputervision_amd import synth
    prev, nxt = synth.lk_pair(1234 + win, rows, cols, dx=1, dy=-1)
    eu, ev = orc.lk_flow(prev, nxt, win)
    gu, gv = lk.calcOpticalFlow(dev(prev), dev(nxt), win)
    assert np.array_equal(host(gu), eu)
    assert np.array_equal(host(gv), ev)


@pytest.mark.parametrize("win", [43, 9, 27])
@pytest.mark.parametrize("rows,cols,pad", [(150, 520, 0), (131, 388, 4), (75, 516, 3), (33, 256, 0), (40, 131, 0)])
def test_lk_generic_two_launch_tiles(mods, rows, cols, pad, win):
    """The two-launch generic level (window 43 packed and unrolled, the others with run-time taps): images wide
    enough for interior 128-column tiles (staged by LDS-DMA) next to edge tiles, widths that are / are not
    multiples of 4 and 128, row pitches that differ from the width, rows that are not multiples of 8 / 32."""
    import torch
    lk, pyr = mods
    from introtocomputervision_amd import synth
    prev, nxt = synth.lk_pair(77 + win + cols, rows, cols, dx=2, dy=-1)
    eu, ev = orc.lk_flow(prev, nxt, win)
    dp = torch.zeros((rows, cols + pad), dtype=torch.float32, device="cuda")
    dn = torch.zeros((rows, cols + pad), dtype=torch.float32, device="cuda")
    dp[:, :cols] = torch.from_numpy(prev)
    dn[:, :cols] = torch.from_numpy(nxt)
    from introtocomputervision_amd import _capi
    # 3 = always two launches (window 43: the unrolled kernels, which the default takes from 1 M pixels on),
    # 2 = always four, 0 = by size
    for form in (3, 2, 0):
        ctx = _capi.Context(0)
        ctx.set_option(_capi.OPT_LK_FORCE_GENERIC, form)
        gu, gv = lk.calcOpticalFlow(dp[:, :cols], dn[:, :cols], win, ctx=ctx)
        assert host(gu).tobytes() == eu.tobytes(), form
        assert host(gv).tobytes() == ev.tobytes(), form


def test_lk_window43_1080p_default_path(mods):
    """config/ps5.yaml's window 43 at the C2 frame size: from 1 M pixels on the default is the unrolled two-launch
    form (interior 128-column tiles by LDS-DMA, 1080 = 33 x 32 + 24 rows); every pixel against the oracle."""
    lk, pyr = mods
    from introtocomputervision_amd import synth
    prev, nxt = synth.lk_pair(0x5EED0005, 1080, 1920, 3, -2)
    eu, ev = orc.lk_flow(prev, nxt, 43)
    gu, gv = lk.calcOpticalFlow(dev(prev), dev(nxt), 43)
    assert host(gu).tobytes() == eu.tobytes()
    assert host(gv).tobytes() == ev.tobytes()


def test_lk_single_level_generic_vs_fused(mods):
    lk, pyr = mods
    from introtocomputervision_amd import synth, _capi
    prev, nxt = synth.lk_pair(99, 135, 240, dx=1, dy=0)
    eu, ev = orc.lk_flow(prev, nxt, 15)
    gctx = _capi.Context(0)
    gctx.set_option(_capi.OPT_LK_FORCE_GENERIC, 1)
    gu, gv = lk.calcOpticalFlow(dev(prev), dev(nxt), 15, ctx=gctx)
    fu, fv = lk.calcOpticalFlow(dev(prev), dev(nxt), 15)
    assert np.array_equal(host(gu), eu) and np.array_equal(host(gv), ev)
    assert np.array_equal(host(fu), eu) and np.array_equal(host(fv), ev)


def test_lk_host_flavour(mods):
    lk, pyr = mods
    from introtocomputervision_amd import synth
    prev, nxt = synth.lk_pair(5, 60, 80, dx=1, dy=1)
    eu, ev = orc.lk_flow(prev, nxt, 15)
    gu, gv = lk.calcOpticalFlow(prev, nxt, 15)
    assert np.array_equal(gu, eu) and np.array_equal(gv, ev)


@pytest.mark.parametrize("win", [15, 9, 21, 11, 7])
@pytest.mark.parametrize("rows,cols,levels", [(270, 480, 5), (135, 240, 4), (128, 192, 3), (101, 203, 4), (64, 64, 1)])
def test_lk_pyr(mods, rows, cols, levels, win):
    lk, pyr = mods
    from introtocomputervision_amd import synth
    prev, nxt = synth.lk_pair(4242, rows, cols, dx=3, dy=-2)
    eu, ev = orc.lk_flow_pyr(prev, nxt, win, levels)
    gu, gv = lk.calcOpticalFlowPyr(dev(prev), dev(nxt), win, levels)
    assert np.array_equal(host(gu), eu)
    assert np.array_equal(host(gv), ev)


def test_lk_pyr_generic_path_matches(mods):
    lk, pyr = mods
    from introtocomputervision_amd import synth, _capi
    prev, nxt = synth.lk_pair(77, 270, 480, dx=3, dy=-2)
    eu, ev = orc.lk_flow_pyr(prev, nxt, 15, 5)
    gctx = _capi.Context(0)
    gctx.set_option(_capi.OPT_LK_FORCE_GENERIC, 1)
    gu, gv = lk.calcOpticalFlowPyr(dev(prev), dev(nxt), 15, 5, ctx=gctx)
    assert np.array_equal(host(gu), eu) and np.array_equal(host(gv), ev)


def test_lk_pyr_batch_and_host(mods):
    lk, pyr = mods
    from introtocomputervision_amd import synth
    pairs = [synth.lk_pair(100 + i, 135, 240, dx=2 + i % 2, dy=-1) for i in range(3)]
    prev = np.stack([p for p, _ in pairs]); nxt = np.stack([n for _, n in pairs])
    gu, gv = lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 15, 4)
    for i in range(3):
        eu, ev = orc.lk_flow_pyr(prev[i], nxt[i], 15, 4)
        assert np.array_equal(host(gu[i]), eu) and np.array_equal(host(gv[i]), ev)
    hu, hv = lk.calcOpticalFlowPyr(prev[0], nxt[0], 15, 4)
    eu, ev = orc.lk_flow_pyr(prev[0], nxt[0], 15, 4)
    assert np.array_equal(hu, eu) and np.array_equal(hv, ev)


def test_lk_pyr_1080p_known_translation(mods):
    """Full BASELINE size (C2): size-independent property -- a (+3,-2) px circular shift must
    come back as flow u ~ +3, v ~ -2 away from the borders.  The CPU oracle itself recovers
    (2.88, -1.97) on this texture, so the KAT bound is 0.25 px (SURVEY 8d guessed 0.1)."""
    lk, pyr = mods
    from introtocomputervision_amd import synth
    prev, nxt = synth.lk_pair(0x5EED0005, 1080, 1920, dx=3, dy=-2)
    gu, gv = lk.calcOpticalFlowPyr(dev(prev), dev(nxt), 15, 5)
    u = host(gu)[64:-64, 64:-64]; v = host(gv)[64:-64, 64:-64]
    assert abs(np.median(u) - 3.0) < 0.25 and abs(np.median(v) + 2.0) < 0.25
    # linearity-free consistency: batch of 2 identical pairs gives identical fields
    p2 = torch.stack([dev(prev), dev(prev)]); n2 = torch.stack([dev(nxt), dev(nxt)])
    bu, bv = lk.calcOpticalFlowPyrBatch(p2, n2, 15, 5)
    assert torch.equal(bu[0], gu) and torch.equal(bu[1], gu) and torch.equal(bv[1], gv)


def test_lk_pyr_1080p_bench_pairs_bit_exact(mods):
    """BASELINE C2 at full size on the very pairs bench.py times (seeds 0x5EED0005 + i), batched as
    the bench runs them: bit-exact against the oracle (bench.py repeats this comparison on its
    cpu_baseline sample and reports it as config.parity_1080p)."""
    lk, pyr = mods
    from introtocomputervision_amd import synth
    pairs = [synth.lk_pair(0x5EED0005 + i, 1080, 1920, 3, -2) for i in range(3)]
    prev = np.stack([p for p, _ in pairs]); nxt = np.stack([n for _, n in pairs])
    gu, gv = lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 15, 5)
    for i in range(3):
        eu, ev = orc.lk_flow_pyr(prev[i], nxt[i], 15, 5)
        assert np.array_equal(host(gu[i]), eu) and np.array_equal(host(gv[i]), ev)


def test_bad_arguments(mods):
    lk, pyr = mods
    from introtocomputervision_amd._capi import MicvError
    a = dev(np.zeros((32, 32), np.float32))
    with pytest.raises(MicvError):
        lk.calcOpticalFlow(a, a, 14)  # even window (assert at OpticalFlow.cpp:48)
    with pytest.raises(MicvError):
        lk.calcOpticalFlowPyr(a, a, 15, 7)  # 32 >> 6 == 0
    with pytest.raises(ValueError):
        lk.calcOpticalFlow(a, dev(np.zeros((16, 32), np.float32)), 15)


@pytest.mark.parametrize("rows,cols,levels,win", [
    (1, 1, 1, 15), (2, 3, 1, 15), (5, 200, 1, 15), (200, 5, 1, 15), (9, 9, 3, 15), (33, 65, 5, 15),
    (16, 16, 5, 7), (31, 17, 2, 1), (40, 130, 3, 21), (64, 64, 7, 15)])
def test_lk_pyr_tiny_and_ragged(mods, rows, cols, levels, win):
    """Edge cases: images smaller than the window / the tile, 1-pixel coarsest levels, window 1."""
    lk, pyr = mods
    rng = np.random.default_rng(rows * 131 + cols)
    prev = (rng.random((rows, cols)) * 255).astype(np.float32)
    nxt = np.roll(prev, 1, axis=1) if cols > 1 else prev.copy()
    eu, ev = orc.lk_flow_pyr(prev, nxt, win, levels)
    gu, gv = lk.calcOpticalFlowPyr(dev(prev), dev(nxt), win, levels)
    assert np.array_equal(host(gu), eu) and np.array_equal(host(gv), ev)
    e1u, e1v = orc.lk_flow(prev, nxt, win)
    g1u, g1v = lk.calcOpticalFlow(dev(prev), dev(nxt), win)
    assert np.array_equal(host(g1u), e1u) and np.array_equal(host(g1v), e1v)


@pytest.mark.parametrize("seed", range(6))
def test_lk_pyr_random_even_shapes_border_tiles(mods, seed):
    """Border tiles run the marching body (zero-padded `next` window, replicated coarse block, pyrUp's
    reflected taps fixed at the first / last image column and row).  Random shapes whose levels all
    double exactly (the fused pyrUp), with widths that are and are not multiples of 4 (LDS-DMA or the
    register path for the window), heights from under one tile to several, a motion with a large and a
    sub-pixel part so the warp's taps cross the image edge, windows 7 / 11 / 15 / 21."""
    lk, pyr = mods
    from introtocomputervision_amd import synth
    rng = np.random.default_rng(0xB0DE + seed)
    for _ in range(5):
        levels = int(rng.integers(2, 5))
        unit = 1 << (levels - 1)
        rows = unit * int(rng.integers(max(1, 4 // unit + 1), max(3, 260 // unit)))
        cols = unit * int(rng.integers(max(1, 4 // unit + 1), max(3, 330 // unit)))
        win = int(rng.choice([7, 11, 15, 21]))
        dx, dy = int(rng.integers(-9, 10)), int(rng.integers(-9, 10))
        prev = synth.smooth_noise(int(rng.integers(1 << 30)), rows, cols)
        nxt = np.ascontiguousarray(np.roll(prev, (dy, dx), (0, 1)))
        nxt = (0.75 * nxt + 0.25 * np.roll(nxt, 1, 1)).astype(np.float32)  # a quarter pixel more in x
        eu, ev = orc.lk_flow_pyr(prev, nxt, win, levels)
        gu, gv = lk.calcOpticalFlowPyr(dev(prev), dev(nxt), win, levels)
        assert np.array_equal(host(gu), eu) and np.array_equal(host(gv), ev), (rows, cols, levels, win, dx, dy)


def test_lk_pitched_views(mods):
    """cv::Mat ROIs: row pitch larger than the width, for inputs and outputs of the host flavour."""
    lk, pyr = mods
    from introtocomputervision_amd import synth
    big_p, big_n = synth.lk_pair(8, 100, 300, 2, -1)
    prev, nxt = big_p[10:90, 20:220], big_n[10:90, 20:220]  # pitch 300 floats, width 200
    assert not prev.flags.c_contiguous
    eu, ev = orc.lk_flow_pyr(np.ascontiguousarray(prev), np.ascontiguousarray(nxt), 15, 3)
    gu, gv = lk.calcOpticalFlowPyr(prev, nxt, 15, 3)
    assert np.array_equal(gu, eu) and np.array_equal(gv, ev)
    dprev, dnxt = dev(big_p)[10:90, 20:220], dev(big_n)[10:90, 20:220]  # device views, same pitch
    du, dv = lk.calcOpticalFlowPyr(dprev, dnxt, 15, 3)
    assert np.array_equal(host(du), eu) and np.array_equal(host(dv), ev)
    su, sv = lk.calcOpticalFlow(dprev, dnxt, 15)
    e1 = orc.lk_flow(np.ascontiguousarray(prev), np.ascontiguousarray(nxt), 15)
    assert np.array_equal(host(su), e1[0]) and np.array_equal(host(sv), e1[1])


def test_lk_pyr_repeated_runs_are_identical(mods):
    """Race detector: odd-sized levels (base flow expanded + resized) and a multi-pair batch, run
    many times -- every run must reproduce the oracle (an output aliasing its own halo input showed
    up here as a rare mismatch)."""
    lk, pyr = mods
    from introtocomputervision_amd import synth
    pairs = [synth.lk_pair(900 + i, 270, 480, 3, -2) for i in range(4)]
    prev = np.stack([p for p, _ in pairs]); nxt = np.stack([n for _, n in pairs])
    exp = [orc.lk_flow_pyr(prev[i], nxt[i], 15, 5) for i in range(4)]
    dp, dn = dev(prev), dev(nxt)
    for _ in range(25):
        gu, gv = lk.calcOpticalFlowPyrBatch(dp, dn, 15, 5)
        for i in range(4):
            assert np.array_equal(host(gu[i]), exp[i][0]) and np.array_equal(host(gv[i]), exp[i][1])


def test_large_flow_leaves_the_staged_window(mods):
    """Flows beyond the 8-px margin of the LDS `next` window take the global-memory fallback."""
    lk, pyr = mods
    from introtocomputervision_amd import synth
    prev, nxt = synth.lk_pair(55, 256, 384, 37, -21)
    eu, ev = orc.lk_flow_pyr(prev, nxt, 15, 4)
    gu, gv = lk.calcOpticalFlowPyr(dev(prev), dev(nxt), 15, 4)
    assert np.array_equal(host(gu), eu) and np.array_equal(host(gv), ev)
    assert np.abs(eu).max() > 12  # the fallback really ran


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,levels", [(540, 960, 3), (1080, 1920, 5), (300, 1000, 2), (700, 330, 3)])
def test_every_tile_is_written_once_border_first_order(rows, cols, levels):
    """The fused kernel deals border tiles first within each XCD (a remap of blockIdx -> tile).
    Outputs pre-filled with NaN through the raw C ABI: a tile the remap missed would stay NaN, and
    the result must still equal the oracle bit for bit."""
    import torch

    from introtocomputervision_amd import synth
    from introtocomputervision_amd._capi import Context, check, lib
    prev, nxt = synth.lk_pair(99, rows, cols, 3, -2)
    dp, dn = torch.from_numpy(prev).cuda(), torch.from_numpy(nxt).cuda()
    u = torch.full((rows, cols), float("nan"), device="cuda")
    v = torch.full((rows, cols), float("nan"), device="cuda")
    ctx = Context(0)
    s = torch.cuda.current_stream().cuda_stream
    check(lib.micv_lk_flow_pyr_dev(ctx.handle, dp.data_ptr(), dn.data_ptr(), rows, cols, cols * 4, 15, levels,
                                   u.data_ptr(), v.data_ptr(), cols * 4, s))
    torch.cuda.synchronize()
    assert not torch.isnan(u).any() and not torch.isnan(v).any()
    # bit for bit at every size, BASELINE C2 (1080x1920, 5 levels) included: ~1 s of oracle
    eu, ev = orc.lk_flow_pyr(prev, nxt, 15, levels)
    assert np.array_equal(u.cpu().numpy(), eu) and np.array_equal(v.cpu().numpy(), ev)
    # the batched driver covers the same tiles with grid.y = pairs
    from introtocomputervision_amd import lk
    bu, bv = lk.calcOpticalFlowPyrBatch(torch.stack([dp, dp]), torch.stack([dn, dn]), 15, levels, ctx=ctx)
    assert torch.equal(bu[0], u) and torch.equal(bu[1], u) and torch.equal(bv[1], v)


@pytest.mark.parametrize("rows,cols,levels", [(128, 96, 3), (135, 241, 4), (67, 120, 1), (270, 481, 5)])
def test_laplacian_pyramid(rows, cols, levels):
    """ps5 runProblem2's Laplacian pyramid (Solution.cpp:187-200), composed on the oracle side from
    its pyrDown / pyrUp / resize restatements; odd sizes take the resize branch (:194-196)."""
    from introtocomputervision_amd import pyr
    a = rand_img(rows, cols, 31)
    G = orc.gaussian_pyramid(a, levels)
    exp = []
    for i in range(levels - 1):
        nxt = orc.pyr_up(G[i + 1])
        if nxt.shape[0] < G[i].shape[0] or nxt.shape[1] < G[i].shape[1]:
            nxt = orc.resize_linear(nxt, *G[i].shape)
        exp.append(G[i] - nxt)
    exp.append(G[levels - 1])
    got = pyr.makeLaplacianPyramid(dev(a), levels)
    assert len(got) == levels
    for g, e in zip(got, exp):
        assert np.array_equal(host(g), e)


@pytest.mark.parametrize("cn,dtype", [(3, np.uint8), (4, np.uint8), (1, np.uint8), (3, np.float32), (4, np.float32), (1, np.float32)])
def test_to_gray_all_frame_formats(mods, cn, dtype):
    """The colour branch of pyr::makeGaussianPyramid (Pyramids.cpp:9-15): cvtColor(COLOR_RGB2GRAY) for
    3/4 channels + convertTo(CV_32F), device and host entry points against the oracle."""
    lk, pyr = mods
    rng = np.random.default_rng(cn * 7 + (1 if dtype == np.uint8 else 2))
    shape = (75, 133) if cn == 1 else (75, 133, cn)
    img = rng.integers(0, 256, shape).astype(dtype) if dtype == np.uint8 else (rng.random(shape) * 255).astype(np.float32)
    exp = orc.to_gray(img)
    assert np.array_equal(host(pyr.toGray(torch.from_numpy(img).cuda())), exp)
    assert np.array_equal(pyr.toGray(img), exp)
    if cn >= 3 and dtype == np.uint8:  # independent statement of the fixed-point formula
        w = img[..., 0].astype(np.int64) * 4899 + img[..., 1].astype(np.int64) * 9617 + img[..., 2].astype(np.int64) * 1868
        assert np.array_equal(exp, ((w + 8192) >> 14).astype(np.float32))


def test_lk_pyr_on_colour_frames_one_upload(mods):
    """lk::calcOpticalFlowPyr as ps5's denseLKWrapper calls it (Solution.cpp:63): CV_8UC3 frames in,
    conversion on the device (micv_lk_flow_pyr_frames_host), bit-exact vs to_gray -> lk_flow_pyr."""
    lk, pyr = mods
    from introtocomputervision_amd import synth
    prev, nxt = synth.lk_pair(61, 200, 320, 3, -2)
    rgb = lambda g: np.clip(np.stack([g * 0.9 + 10, g * 0.7 + 40, 255 - g * 0.8], -1), 0, 255).astype(np.uint8)
    p3, n3 = rgb(prev), rgb(nxt)
    eu, ev = orc.lk_flow_pyr(orc.to_gray(p3), orc.to_gray(n3), 15, 4)
    gu, gv = lk.calcOpticalFlowPyrFrames(p3, n3, 15, 4)
    assert np.array_equal(gu, eu) and np.array_equal(gv, ev)
    g8u, g8v = lk.calcOpticalFlowPyrFrames(prev.astype(np.uint8), nxt.astype(np.uint8), 15, 4)
    e8 = orc.lk_flow_pyr(prev.astype(np.uint8).astype(np.float32), nxt.astype(np.uint8).astype(np.float32), 15, 4)
    assert np.array_equal(g8u, e8[0]) and np.array_equal(g8v, e8[1])


@pytest.mark.parametrize("rows,cols,levels,batch", [(270, 480, 2, 1), (540, 960, 3, 2), (1080, 1920, 5, 1), (330, 700, 3, 3),
                                                    (200, 210, 2, 1), (97, 400, 2, 2)])
@pytest.mark.parametrize("max_chain", [2, 3, 4, 8, 32])
def test_tile_chains_are_bit_exact(mods, rows, cols, levels, batch, max_chain):
    """Tile chains of the fused level kernel (one workgroup walks down a column of tiles and keeps the
    last 14 gradient rows in LDS for the tile below): any chain length, any image size, the same bits
    as single tiles and as the oracle.  MICV_OPT_LK_CHAIN forces chains at sizes the automatic rule
    would run as single tiles."""
    lk, pyr = mods
    from introtocomputervision_amd import synth, _capi
    pairs = [synth.lk_pair(4000 + i + rows, rows, cols, 3, -2) for i in range(batch)]
    prev = np.stack([p for p, _ in pairs]); nxt = np.stack([n for _, n in pairs])
    ctx = _capi.Context(0)
    ctx.set_option(_capi.OPT_LK_CHAIN, max_chain)
    u = torch.full((batch, rows, cols), float("nan"), device="cuda")
    v = torch.full((batch, rows, cols), float("nan"), device="cuda")
    for rep in range(2):  # second call: the cached schedule
        lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 15, levels, ctx=ctx, out=(u, v))
    for i in range(batch):
        eu, ev = orc.lk_flow_pyr(prev[i], nxt[i], 15, levels)
        assert np.array_equal(host(u[i]), eu) and np.array_equal(host(v[i]), ev), (i, max_chain)
    ctx.set_option(_capi.OPT_LK_CHAIN, 1)  # chains off: identical
    su, sv = lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 15, levels, ctx=ctx)
    assert torch.equal(su, u) and torch.equal(sv, v)


@pytest.mark.parametrize("rows,cols,levels,batch", [(540, 960, 3, 2), (330, 700, 3, 3), (97, 400, 2, 2)])
@pytest.mark.parametrize("max_chain", [2, 4, 32])
def test_tile_chains_window_11(mods, rows, cols, levels, batch, max_chain):
    """The chain kernel's other instantiation (window 11, halo 8): forced chains against chains off and the oracle."""
    lk, pyr = mods
    from introtocomputervision_amd import synth, _capi
    pairs = [synth.lk_pair(4400 + i + rows, rows, cols, 2, -3) for i in range(batch)]
    prev = np.stack([p for p, _ in pairs]); nxt = np.stack([n for _, n in pairs])
    ctx = _capi.Context(0)
    ctx.set_option(_capi.OPT_LK_CHAIN, max_chain)
    u, v = lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 11, levels, ctx=ctx)
    ctx.set_option(_capi.OPT_LK_CHAIN, 1)
    su, sv = lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 11, levels, ctx=ctx)
    assert torch.equal(su, u) and torch.equal(sv, v)
    eu, ev = orc.lk_flow_pyr(prev[0], nxt[0], 11, levels)
    assert np.array_equal(host(u[0]), eu) and np.array_equal(host(v[0]), ev)


@pytest.mark.parametrize("win", [15, 11, 7])
@pytest.mark.parametrize("rows,cols,batch", [(270, 480, 8), (270, 480, 16), (270, 480, 7)])
def test_automatic_chain_rule_is_bit_exact(mods, rows, cols, batch, win):
    """MICV_OPT_LK_CHAIN = 0 (default) runs pairs of tiles where a launch is a little over one or two rounds of
    workgroups: 8 (16) pairs of 270x480 are 576 (1152) 64x32 tiles on 512 slots -- level 2 of the bench's step.
    Same bits as chains off (1) and as the oracle; 7 pairs (504 tiles) stay on the plain grid.  Window 11 has a chain
    kernel too (window 7's region does not split into whole LDS-DMA waves: always the plain grid)."""
    lk, pyr = mods
    from introtocomputervision_amd import synth, _capi
    pairs = [synth.lk_pair(5100 + i, rows, cols, 2, -1) for i in range(batch)]
    prev = np.stack([p for p, _ in pairs]); nxt = np.stack([n for _, n in pairs])
    out = {}
    for opt in (0, 1):
        ctx = _capi.Context(0)
        ctx.set_option(_capi.OPT_LK_CHAIN, opt)
        ctx.set_lk_groups(1)
        for rep in range(2):
            out[opt] = lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), win, 2, ctx=ctx)
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    for i in (0, batch - 1):
        eu, ev = orc.lk_flow_pyr(prev[i], nxt[i], win, 2)
        assert np.array_equal(host(out[0][0][i]), eu) and np.array_equal(host(out[0][1][i]), ev), i


@pytest.mark.parametrize("rows,cols,levels,batch,win", [(270, 480, 2, 1, 15), (540, 960, 3, 2, 15), (1080, 1920, 5, 1, 15),
                                                        (330, 700, 3, 3, 15), (200, 210, 2, 1, 15), (97, 400, 2, 2, 15),
                                                        (540, 960, 3, 9, 15), (300, 520, 3, 2, 7), (300, 520, 2, 3, 11)])
def test_streamed_launch_is_bit_exact(mods, rows, cols, levels, batch, win):
    """The streamed level launch (persistent workgroups take tiles by ticket; an interior tile's `next`
    window and coarse block are staged during the tile before it): MICV_OPT_LK_STREAM = 1 (off by
    default: measured slower than the plain grid).  Same bits as the plain launch and as the oracle, call
    after call (the launch resets its own ticket counters)."""
    lk, pyr = mods
    from introtocomputervision_amd import synth, _capi
    pairs = [synth.lk_pair(9000 + i + rows, rows, cols, 3, -2) for i in range(batch)]
    prev = np.stack([p for p, _ in pairs]); nxt = np.stack([n for _, n in pairs])
    ctx = _capi.Context(0)
    ctx.set_option(_capi.OPT_LK_STREAM, 1)
    u = torch.full((batch, rows, cols), float("nan"), device="cuda")
    v = torch.full((batch, rows, cols), float("nan"), device="cuda")
    for rep in range(20):  # more calls than ticket slots: every slot is reused
        u.fill_(float("nan")); v.fill_(float("nan"))
        lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), win, levels, ctx=ctx, out=(u, v))
        if rep in (0, 1, 19):
            for i in range(batch if rep == 0 else 1):
                eu, ev = orc.lk_flow_pyr(prev[i], nxt[i], win, levels)
                assert np.array_equal(host(u[i]), eu) and np.array_equal(host(v[i]), ev), (rep, i)
    ctx.set_option(_capi.OPT_LK_STREAM, 0)
    su, sv = lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), win, levels, ctx=ctx)
    assert torch.equal(su, u) and torch.equal(sv, v)
    with pytest.raises(Exception):
        ctx.set_option(_capi.OPT_LK_STREAM, 2)


@pytest.mark.parametrize("rows,cols,levels,batch,win", [(1080, 1920, 5, 2, 15), (540, 960, 4, 9, 15), (270, 481, 3, 2, 15),
                                                        (333, 517, 4, 1, 15), (97, 400, 2, 2, 15), (300, 520, 3, 2, 7),
                                                        (300, 520, 3, 3, 11), (540, 960, 3, 2, 21), (64, 64, 3, 1, 15)])
@pytest.mark.parametrize("direct", [1, 2])
def test_levels_read_straight_from_level_0_are_bit_exact(mods, rows, cols, levels, batch, win, direct):
    """MICV_OPT_LK_DIRECT_LEVELS = n: the fused level kernels of levels >= n take prev / next from level 0 itself
    (L_k(y, x) = L_0(2^k y + 2^k - 1, 2^k x + 2^k - 1), Pyramids.cu:31, staged by dword LDS-DMA gathers) instead of
    from a pyramid another launch built; n = 1 drops that launch.  Odd sizes (unaligned rows, odd-sized levels
    with a resized base flow), border-only levels, tile chains (540x960 x 9 pairs at level 2) and NaN pixels:
    the same bits as the built pyramid and as the oracle.  Off by default (measured); the gather
    kernels exist for window 15, other windows keep building the pyramid whatever the option says."""
    lk, pyr = mods
    from introtocomputervision_amd import synth, _capi
    pairs = [synth.lk_pair(7000 + i + rows, rows, cols, 3, -2) for i in range(batch)]
    prev = np.stack([p for p, _ in pairs]); nxt = np.stack([n for _, n in pairs])
    prev[0, rows // 3, cols // 3] = np.nan
    nxt[-1, rows // 2, cols // 2] = np.inf
    ctx = _capi.Context(0)
    bu, bv = lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), win, levels, ctx=ctx)
    ctx.set_option(_capi.OPT_LK_DIRECT_LEVELS, direct)
    u = torch.full((batch, rows, cols), float("nan"), device="cuda")
    v = torch.full((batch, rows, cols), float("nan"), device="cuda")
    for rep in range(2):
        lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), win, levels, ctx=ctx, out=(u, v))
    assert host(u).tobytes() == host(bu).tobytes() and host(v).tobytes() == host(bv).tobytes()
    eu, ev = orc.lk_flow_pyr(prev[batch - 1], nxt[batch - 1], win, levels)
    # (against the oracle a NaN matches a NaN: the sign of a produced NaN is not part of the contract)
    assert np.array_equal(host(u[batch - 1]), eu, equal_nan=True) and np.array_equal(host(v[batch - 1]), ev, equal_nan=True)
    # a pitched view (rows of a wider buffer, base not 16-byte aligned): the gather has no alignment rules
    wide_p = torch.zeros((batch, rows, cols + 7), device="cuda"); wide_n = torch.zeros_like(wide_p)
    wide_p[:, :, 3:3 + cols] = dev(prev); wide_n[:, :, 3:3 + cols] = dev(nxt)
    if batch == 1:
        gu, gv = lk.calcOpticalFlowPyr(wide_p[0, :, 3:3 + cols], wide_n[0, :, 3:3 + cols], win, levels, ctx=ctx)
        assert host(gu).tobytes() == host(bu[0]).tobytes() and host(gv).tobytes() == host(bv[0]).tobytes()
    with pytest.raises(Exception):
        ctx.set_option(_capi.OPT_LK_DIRECT_LEVELS, 16)


@pytest.mark.parametrize("rows,cols,levels,batch", [(1080, 1920, 4, 1), (540, 960, 3, 4), (1085, 1925, 2, 1), (600, 700, 2, 3)])
def test_window_21_tile_shapes_are_bit_exact(mods, rows, cols, levels, batch):
    """Window 21, the reference's default winSize (OpticalFlow.h:10,18): launches of at least one round run 64x64 tiles
    with 1024 threads (r04: 149 KB of LDS, region 1.89x its outputs instead of 2.4x -- level 0 of 4 x 1080p 216 -> 148 us);
    MICV_OPT_LK_TALL_TILES = 1 keeps r03's 64x32 / 1024-thread tiles, -1 the 64x16 / 512-thread ones.  Same bits, and the
    oracle's, on sizes that are not multiples of the tile and with a NaN pixel."""
    lk, pyr = mods
    from introtocomputervision_amd import synth, _capi
    pairs = [synth.lk_pair(2100 + i + rows, rows, cols, 3, -2) for i in range(batch)]
    prev = np.stack([p for p, _ in pairs]); nxt = np.stack([n for _, n in pairs])
    nxt[0, rows // 2, cols // 2] = np.nan
    outs = []
    for tall in (0, 1, -1):
        ctx = _capi.Context(0)
        ctx.set_option(_capi.OPT_LK_TALL_TILES, tall)
        outs.append(lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 21, levels, ctx=ctx))
    for u, v in outs[1:]:
        assert host(u).tobytes() == host(outs[0][0]).tobytes() and host(v).tobytes() == host(outs[0][1]).tobytes()
    eu, ev = orc.lk_flow_pyr(prev[batch - 1], nxt[batch - 1], 21, levels)
    assert np.array_equal(host(outs[0][0][batch - 1]), eu, equal_nan=True) and np.array_equal(host(outs[0][1][batch - 1]), ev, equal_nan=True)


@pytest.mark.parametrize("rows,cols,levels,batch", [(1080, 1920, 5, 2), (540, 960, 3, 8), (1080, 1920, 2, 1), (700, 1000, 3, 3), (1090, 1930, 3, 2)])
def test_32x64_tiles_are_bit_exact(mods, rows, cols, levels, batch):
    """MICV_OPT_LK_TALL_TILES = 2: launches of at least two rounds of the window-15 level kernel on 32x64 tiles
    (512 threads, two workgroups per CU; the row pass runs 78 gradient rows for 64 output rows instead of 46 for 32,
    row buffers two rows to a 64-float unit, column values by (I, I + 8) pairs).  Same bits as 64x32 tiles and as the
    oracle; sizes that are not multiples of the tile, launches under the threshold (which keep 64x32 tiles)."""
    lk, pyr = mods
    from introtocomputervision_amd import synth, _capi
    pairs = [synth.lk_pair(4200 + i + rows, rows, cols, 3, -2) for i in range(batch)]
    prev = np.stack([p for p, _ in pairs]); nxt = np.stack([n for _, n in pairs])
    prev[0, rows // 3, cols // 3] = np.nan
    ctx = _capi.Context(0)
    bu, bv = lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 15, levels, ctx=ctx)
    ctx.set_option(_capi.OPT_LK_TALL_TILES, 2)
    u = torch.full((batch, rows, cols), float("nan"), device="cuda")
    v = torch.full((batch, rows, cols), float("nan"), device="cuda")
    for rep in range(2):
        lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 15, levels, ctx=ctx, out=(u, v))
    assert host(u).tobytes() == host(bu).tobytes() and host(v).tobytes() == host(bv).tobytes()
    eu, ev = orc.lk_flow_pyr(prev[batch - 1], nxt[batch - 1], 15, levels)
    assert np.array_equal(host(u[batch - 1]), eu, equal_nan=True) and np.array_equal(host(v[batch - 1]), ev, equal_nan=True)
    with pytest.raises(Exception):
        ctx.set_option(_capi.OPT_LK_TALL_TILES, 4)  # 3 = the 1024-thread 64x32 experiment, the last valid value


@pytest.mark.parametrize("rows,cols,levels,batch,pct", [(1080, 1920, 5, 2, 1), (1080, 1920, 5, 1, 0), (540, 960, 3, 1, 0), (272, 484, 4, 3, 1),
                                                         (300, 332, 3, 2, 1), (96, 132, 5, 1, 0), (1080, 1920, 6, 8, 1), (2160, 3840, 5, 1, 0), (1000, 4100, 4, 1, 0),
                                                         (2160, 3840, 3, 5, 1)])  # the last: more build units than carried workgroups (stride loop)
def test_carried_pyramid_build_is_bit_exact(mods, rows, cols, levels, batch, pct):
    """MICV_OPT_LK_BUILD_OVERLAP (0 = single pairs only, n = every batch; window 15, >= 3 levels): no build launch -- the top level reads level 0
    itself and the launches of levels top .. 2 carry the pyramid build as extra workgroups (LkBuildJob).  Same bits as
    the single build launch in front (-1) and as the oracle; repeated
    calls on one context rewrite the same arena."""
    lk, pyr = mods
    from introtocomputervision_amd import synth, _capi
    pairs = [synth.lk_pair(6200 + i + rows, rows, cols, 2, -3) for i in range(batch)]
    prev = np.stack([p for p, _ in pairs]); nxt = np.stack([n for _, n in pairs])
    ctx = _capi.Context(0)
    assert ctx.get_option(_capi.OPT_LK_BUILD_OVERLAP) == 0
    ctx.set_option(_capi.OPT_LK_BUILD_OVERLAP, pct)
    u = torch.full((batch, rows, cols), float("nan"), device="cuda"); v = torch.full_like(u, float("nan"))
    for rep in range(3):
        lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 15, levels, ctx=ctx, out=(u, v))
    ctx.set_option(_capi.OPT_LK_BUILD_OVERLAP, -1)
    bu, bv = lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 15, levels, ctx=ctx)
    assert host(u).tobytes() == host(bu).tobytes() and host(v).tobytes() == host(bv).tobytes()
    if rows * cols * levels <= 1080 * 1920 * 5:
        eu, ev = orc.lk_flow_pyr(prev[batch - 1], nxt[batch - 1], 15, levels)
        assert np.array_equal(host(u[batch - 1]), eu, equal_nan=True) and np.array_equal(host(v[batch - 1]), ev, equal_nan=True)
    with pytest.raises(Exception):
        ctx.set_option(_capi.OPT_LK_BUILD_OVERLAP, 2)


def test_carried_build_leaves_unaligned_inputs_to_the_build_launch(mods):
    """Rows that are not 16-byte multiples (cols % 4 != 0) or a strided view: the carried build does not apply and the
    call takes the build launch; results as the oracle's either way."""
    lk, pyr = mods
    from introtocomputervision_amd import synth
    for rows, cols in ((131, 203), (270, 482)):
        p, n = synth.lk_pair(77 + cols, rows, cols, 1, 2)
        u, v = lk.calcOpticalFlowPyr(dev(p), dev(n), winSize=15, levels=3)
        eu, ev = orc.lk_flow_pyr(p, n, 15, 3)
        assert np.array_equal(host(u), eu, equal_nan=True) and np.array_equal(host(v), ev, equal_nan=True)


@pytest.mark.parametrize("rows,cols,levels,batch", [(1080, 1920, 5, 2), (700, 1000, 3, 3)])
def test_1024_thread_tiles_are_bit_exact(mods, rows, cols, levels, batch):
    """MICV_OPT_LK_TALL_TILES = 3: the 64x32 tile with 1024 threads (an r04 experiment kept as an option,
    profiles/r04/lk_ab.txt).  Same bits as the default launch."""
    lk, pyr = mods
    from introtocomputervision_amd import synth, _capi
    pairs = [synth.lk_pair(5200 + i + rows, rows, cols, -2, 3) for i in range(batch)]
    prev = np.stack([p for p, _ in pairs]); nxt = np.stack([n for _, n in pairs])
    ctx = _capi.Context(0)
    bu, bv = lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 15, levels, ctx=ctx)
    ctx.set_option(_capi.OPT_LK_TALL_TILES, 3)
    u, v = lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 15, levels, ctx=ctx)
    assert host(u).tobytes() == host(bu).tobytes() and host(v).tobytes() == host(bv).tobytes()


@pytest.mark.parametrize("rows,cols,levels,batch", [(270, 480, 3, 1), (135, 240, 2, 8), (67, 120, 1, 3), (1080, 1920, 5, 1), (100, 333, 3, 2)])
def test_short_tiles_are_bit_exact(mods, rows, cols, levels, batch):
    """Launches of at most one 64x16 tile per CU run the half-height form of the win-15 level kernel
    (MICV_OPT_LK_SHORT_TILES = 0, the default); -1 keeps 64x32 tiles.  Same bits either way."""
    lk, pyr = mods
    from introtocomputervision_amd import synth, _capi
    pairs = [synth.lk_pair(7000 + i + cols, rows, cols, 3, -2) for i in range(batch)]
    prev = np.stack([p for p, _ in pairs]); nxt = np.stack([n for _, n in pairs])
    exp = [orc.lk_flow_pyr(prev[i], nxt[i], 15, levels) for i in range(batch)]
    for opt in (0, -1):
        ctx = _capi.Context(0)
        ctx.set_option(_capi.OPT_LK_SHORT_TILES, opt)
        u, v = lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 15, levels, ctx=ctx)
        for i in range(batch):
            assert np.array_equal(host(u[i]), exp[i][0]) and np.array_equal(host(v[i]), exp[i][1]), (opt, i)
    su, sv = lk.calcOpticalFlow(dev(prev[0]), dev(nxt[0]), 15)  # single level, NONE mode, short tiles when small
    e1 = orc.lk_flow(prev[0], nxt[0], 15)
    assert np.array_equal(host(su), e1[0]) and np.array_equal(host(sv), e1[1])


def test_batch_wrapper_validates_its_buffers(mods):
    """a wrong `out` / `next` must be refused before anything is written out of bounds, and
    two torch streams must not share one default context (one scratch arena)."""
    lk, pyr = mods
    prev = torch.zeros((2, 64, 96), device="cuda")
    nxt = torch.zeros_like(prev)
    good = (torch.empty_like(prev), torch.empty_like(prev))
    lk.calcOpticalFlowPyrBatch(prev, nxt, 15, 2, out=good)
    bad_outs = [(torch.empty((2, 32, 96), device="cuda"), good[1]),            # too small
                (good[0].double(), good[1]),                                      # dtype
                (good[0].cpu(), good[1]),                                         # device
                (good[0][:, :, ::2], good[1]),                                    # not contiguous
                (good[0], good[0]),                                               # aliased outputs
                (prev, good[1])]                                                  # output aliases an input
    for out in bad_outs:
        with pytest.raises(ValueError):
            lk.calcOpticalFlowPyrBatch(prev, nxt, 15, 2, out=out)
    with pytest.raises(ValueError):
        lk.calcOpticalFlowPyrBatch(prev, nxt.cpu(), 15, 2)
    with pytest.raises(ValueError):
        lk.calcOpticalFlowPyrBatch(prev, nxt[:, :32], 15, 2)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    assert lk.default_context(0, s1.cuda_stream) is not lk.default_context(0, s2.cuda_stream)
    assert lk.default_context(0, s1.cuda_stream) is lk.default_context(0, s1.cuda_stream)
    # the cache evicts only contexts nobody else holds; a held one stays usable, a closed one raises
    held = lk.default_context(0, s1.cuda_stream)
    streams = [torch.cuda.Stream() for _ in range(12)]
    unheld_ids = [id(lk.default_context(0, t.cuda_stream)) for t in streams]
    assert lk.default_context(0, s1.cuda_stream) is held and held.handle
    assert len(lk._default_ctx) <= lk._DEFAULT_CTX_MAX + 1 and len(set(unheld_ids)) >= 2
    from introtocomputervision_amd._capi import Context
    tmp = Context(0)
    tmp.close()
    with pytest.raises(RuntimeError):
        tmp.handle
    # two streams, default contexts, concurrently: both results right
    from introtocomputervision_amd import synth
    a = [synth.lk_pair(1 + i, 270, 480, 3, -2) for i in range(2)]
    exp = [orc.lk_flow_pyr(p, n, 15, 4) for p, n in a]
    res = []
    for (p, n), s in zip(a, (s1, s2)):
        with torch.cuda.stream(s):
            dp, dn = dev(p[None]), dev(n[None])
            for _ in range(5):
                r = lk.calcOpticalFlowPyrBatch(dp, dn, 15, 4)
            res.append(r)
    torch.cuda.synchronize()
    for (gu, gv), (eu, ev) in zip(res, exp):
        assert np.array_equal(host(gu[0]), eu) and np.array_equal(host(gv[0]), ev)


@pytest.mark.parametrize("fr,fc,drows,dcols", [(67, 120, 135, 240), (67, 120, 135, 241), (33, 60, 67, 121), (5, 7, 11, 15),
                                              (1, 1, 3, 3), (100, 37, 201, 75), (8, 300, 17, 600)])
def test_level_on_odd_sizes_uses_the_resized_base_flow(mods, fr, fc, drows, dcols):
    """OpticalFlow.cpp:139-151 for levels that are not twice the coarser one: 2 * pyrUp then cv::resize.
    micv_lk_level_dev against the oracle composition (pyr_up -> resize_linear -> warp -> lk_flow), with
    the tiled expand + resize kernel behind it; several pairs through the batch entry point too."""
    lk, pyr = mods
    from introtocomputervision_amd import synth, shard, _capi
    rng = np.random.default_rng(fr * 31 + dcols)
    prev, nxt = synth.lk_pair(50 + fr, drows, dcols, 2, -1)
    fu = (rng.standard_normal((fr, fc)) * 1.5).astype(np.float32)
    fv = (rng.standard_normal((fr, fc)) * 1.5).astype(np.float32)
    du = orc.resize_linear(2.0 * orc.pyr_up(fu), drows, dcols)
    dv = orc.resize_linear(2.0 * orc.pyr_up(fv), drows, dcols)
    ex, ey = orc.lk_flow(prev, orc.lk_warp(nxt, du, dv), 15)
    eu, ev = du + ex, dv + ey
    ctx = _capi.Context(0)
    fn = shard.gpu_level_fn(ctx, 15)
    gu, gv = fn(dev(prev), dev(nxt), dev(fu), dev(fv), 0, drows)
    assert np.array_equal(host(gu), eu) and np.array_equal(host(gv), ev)


@pytest.mark.parametrize("rows,cols,levels,batch", [(540, 960, 3, 8), (1080, 1920, 5, 3), (600, 1000, 2, 7), (517, 1111, 3, 7)])
@pytest.mark.parametrize("streamed", [0, 1])
def test_tall_tiles_are_bit_exact(mods, rows, cols, levels, batch, streamed):
    """MICV_OPT_LK_TALL_TILES: the win-15 level kernel on 64x64 tiles with 1024 threads (one workgroup per
    CU), alone and under the streamed launch that stages the next tile ahead; launches of at least 1024
    such tiles.  Same bits as the 64x32 tiles and as the oracle (incl. partial last tile rows: 540 = 8.4
    tiles, 1080 = 16.9; odd sizes run the FULL-flow mode on the coarser levels)."""
    lk, pyr = mods
    from introtocomputervision_amd import synth, _capi
    assert (-(-cols // 64)) * (-(-rows // 64)) * batch >= 1024
    pairs = [synth.lk_pair(5000 + i + rows, rows, cols, 3, -2) for i in range(batch)]
    prev = np.stack([p for p, _ in pairs]); nxt = np.stack([n for _, n in pairs])
    ref_u, ref_v = lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 15, levels, ctx=_capi.Context(0))
    ctx = _capi.Context(0)
    ctx.set_option(_capi.OPT_LK_TALL_TILES, 1)
    ctx.set_option(_capi.OPT_LK_STREAM, streamed)
    u = torch.full((batch, rows, cols), float("nan"), device="cuda")
    v = torch.full((batch, rows, cols), float("nan"), device="cuda")
    for rep in range(3):
        u.fill_(float("nan")); v.fill_(float("nan"))
        lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 15, levels, ctx=ctx, out=(u, v))
        assert torch.equal(u, ref_u) and torch.equal(v, ref_v), rep
    eu, ev = orc.lk_flow_pyr(prev[0], nxt[0], 15, levels)
    assert np.array_equal(host(u[0]), eu) and np.array_equal(host(v[0]), ev)


@pytest.mark.parametrize("rows,cols,levels,batch", [(540, 960, 3, 4), (1080, 1920, 5, 1), (517, 1111, 3, 2), (300, 400, 2, 1)])
def test_window_21_tile_forms_are_bit_exact(mods, rows, cols, levels, batch):
    """Window 21 (the reference's default winSize, OpticalFlow.h:9,18): big launches run 64x32 tiles with 1024
    threads, small ones 64x16 tiles with 512 (MICV_OPT_LK_TALL_TILES = -1 forces the latter).  Same bits,
    equal to the oracle."""
    lk, pyr = mods
    from introtocomputervision_amd import synth, _capi
    pairs = [synth.lk_pair(100 + i, rows, cols, 3, -2) for i in range(batch)]
    prev = np.stack([p for p, _ in pairs]); nxt = np.stack([n for _, n in pairs])
    outs = []
    for opt in (0, -1, 1):
        ctx = _capi.Context(0)
        ctx.set_option(_capi.OPT_LK_TALL_TILES, opt)
        outs.append(lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 21, levels, ctx=ctx))
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])
    eu, ev = orc.lk_flow_pyr(prev[0], nxt[0], 21, levels)
    assert np.array_equal(host(outs[0][0][0]), eu) and np.array_equal(host(outs[0][1][0]), ev)


def test_streamed_launch_with_stream_groups(mods):
    """the streamed launch's ticket counters were handed out round-robin, so two stream groups of
    one call (launches in flight at the same time on different streams) could share a slot and take tickets
    off each other.  Slots are per stream now: four groups x streamed launches, repeated, equal the plain run."""
    lk, pyr = mods
    from introtocomputervision_amd import synth, _capi
    rows, cols, levels, batch = 540, 960, 4, 8
    pairs = [synth.lk_pair(300 + i, rows, cols, 3, -2) for i in range(batch)]
    prev = dev(np.stack([p for p, _ in pairs])); nxt = dev(np.stack([n for _, n in pairs]))
    ref = lk.calcOpticalFlowPyrBatch(prev, nxt, 15, levels, ctx=_capi.Context(0))
    ctx = _capi.Context(0)
    ctx.set_option(_capi.OPT_LK_STREAM, 1)
    ctx.set_option(_capi.OPT_LK_STREAM_GROUPS, 4)
    for rep in range(12):
        u = torch.full((batch, rows, cols), float("nan"), device="cuda")
        v = torch.full((batch, rows, cols), float("nan"), device="cuda")
        lk.calcOpticalFlowPyrBatch(prev, nxt, 15, levels, ctx=ctx, out=(u, v))
        assert torch.equal(u, ref[0]) and torch.equal(v, ref[1]), rep


@pytest.mark.parametrize("rows,cols,levels,batch", [(128, 128, 2, 1), (270, 480, 3, 2), (540, 960, 3, 3), (1080, 1920, 5, 2), (64, 64, 2, 1),
                                                     (200, 328, 2, 5), (1088, 1924, 2, 1), (66, 1000, 2, 2), (1000, 68, 2, 2), (2160, 3840, 3, 1)])
@pytest.mark.parametrize("mode", [1, 2, 3])
def test_split_level_launch_is_bit_exact(mods, rows, cols, levels, batch, mode):
    """MICV_OPT_LK_SPLIT (r05, lk_split.hip): a level launch as a pre-pass -- pyrUp + warp + Sobel once per pixel, Ix / Iy / It
    into padded planes whose ring holds the BORDER_REFLECT_101 copies, base flow into u, v -- plus the streaming
    window-sum kernel.  1 = every launch that can, 0 = never (the default: the split measured slower):
    same bits, and the oracle's.  Repeated calls rewrite the same planes."""
    lk, pyr = mods
    from introtocomputervision_amd import synth, _capi
    pairs = [synth.lk_pair(8100 + i + rows, rows, cols, 3, -2) for i in range(batch)]
    prev = np.stack([p for p, _ in pairs]); nxt = np.stack([n for _, n in pairs])
    ctx = _capi.Context(0)
    ctx.set_option(_capi.OPT_LK_SPLIT, mode)  # 2: base flow through u, v; 3: variant A' (warped image only)
    u = torch.full((batch, rows, cols), float("nan"), device="cuda"); v = torch.full_like(u, float("nan"))
    for rep in range(2):
        lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 15, levels, ctx=ctx, out=(u, v))
    ctx.set_option(_capi.OPT_LK_SPLIT, 0)
    bu, bv = lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 15, levels, ctx=ctx)
    du = host(u).view(np.uint32) != host(bu).view(np.uint32); dv = host(v).view(np.uint32) != host(bv).view(np.uint32)
    assert not du.any() and not dv.any(), (int(du.sum()), int(dv.sum()), np.argwhere(du | dv)[:8].tolist())
    if rows * cols * levels <= 1080 * 1920 * 5:
        eu, ev = orc.lk_flow_pyr(prev[batch - 1], nxt[batch - 1], 15, levels)
        assert np.array_equal(host(u[batch - 1]), eu, equal_nan=True) and np.array_equal(host(v[batch - 1]), ev, equal_nan=True)
    with pytest.raises(Exception):
        ctx.set_option(_capi.OPT_LK_SPLIT, 4)


def test_level_kernel_name_comes_from_the_dispatch(mods):
    """micv_lk_level_kernel_name: the launch dispatch answers (name_out: same code path, nothing launched), so the names
    bench.py / the PMC tools filter profiler rows by follow every option -- the Python mirror it replaces had drifted
    (TALL_TILES = 3, the gather forms)."""
    from introtocomputervision_amd import _capi
    ctx = _capi.Context(0)
    assert ctx.lk_level_kernel_name(15, 1080, 1920, 8) == "lk_level_kernel<7, 1, 512, 32, false, 64>"
    assert ctx.lk_level_kernel_name(15, 270, 480, 1) == "lk_level_kernel<7, 1, 512, 16, false, 64>"
    assert ctx.lk_level_kernel_name(21, 1080, 1920, 8) == "lk_level_kernel<10, 1, 1024, 64, false, 64>"
    assert ctx.lk_level_kernel_name(7, 1080, 1920, 8) == "lk_level_kernel<3, 1, 512, 32, false, 64>"
    ctx.set_option(_capi.OPT_LK_TALL_TILES, 3)
    assert ctx.lk_level_kernel_name(15, 1080, 1920, 8) == "lk_level_kernel<7, 1, 1024, 32, false, 64>"
    ctx.set_option(_capi.OPT_LK_TALL_TILES, 2)
    assert ctx.lk_level_kernel_name(15, 1080, 1920, 8) == "lk_level_kernel<7, 1, 512, 64, false, 32>"
    ctx.set_option(_capi.OPT_LK_TALL_TILES, 0)
    ctx.set_option(_capi.OPT_LK_CHAIN, 2)
    assert ctx.lk_level_kernel_name(15, 1080, 1920, 8) == "lk_level_chain_kernel<7, 512, false>"
    ctx.set_option(_capi.OPT_LK_CHAIN, 0)
    ctx.set_option(_capi.OPT_LK_STREAM, 1)
    assert ctx.lk_level_kernel_name(15, 1080, 1920, 8) == "lk_level_stream_kernel<7, 512, 32>"
    ctx.set_option(_capi.OPT_LK_STREAM, 0)
    ctx.set_option(_capi.OPT_LK_SPLIT, 1)
    assert ctx.lk_level_kernel_name(15, 1080, 1920, 8) == "lk_grad_kernel<3, 512, 32> + lk_sums_stream_kernel<7>"
    ctx.set_option(_capi.OPT_LK_SPLIT, 0)
    ctx.set_option(_capi.OPT_LK_NARROW_TILES, 1)
    assert ctx.lk_level_kernel_name(15, 1080, 1920, 8) == "lk_level_kernel<7, 1, 256, 32, false, 64>"


@pytest.mark.parametrize("rows,cols,levels,batch,blocks", [(270, 480, 2, 1, 4), (540, 960, 3, 3, 16), (1080, 1920, 5, 2, 16), (1080, 1920, 2, 1, 7),
                                                            (200, 328, 2, 5, 2), (1088, 1924, 2, 1, 64), (160, 1000, 2, 2, 1), (1000, 164, 2, 2, 3),
                                                            (2160, 3840, 3, 1, 16), (128, 128, 2, 1, 16)])
def test_strip_launch_is_bit_exact(mods, rows, cols, levels, batch, blocks):
    """MICV_OPT_LK_STRIP (r05, lk_strip.hpp): the interior tiles of a level as strips streamed down in blocks of 16 rows
    (carried row-pass rows of the five fields, carried warped rows), the border tiles as tiles, one launch.  Same bits as
    the tile launch and the oracle for every segment length, incl. segments that are not whole, one-strip and one-segment
    interiors, and images too small to have an interior (the tile launch then runs)."""
    lk, pyr = mods
    from introtocomputervision_amd import synth, _capi
    pairs = [synth.lk_pair(9100 + i + rows, rows, cols, 3, -2) for i in range(batch)]
    prev = np.stack([p for p, _ in pairs]); nxt = np.stack([n for _, n in pairs])
    ctx = _capi.Context(0)
    ctx.set_option(_capi.OPT_LK_STRIP, blocks)
    u = torch.full((batch, rows, cols), float("nan"), device="cuda"); v = torch.full_like(u, float("nan"))
    for rep in range(2):
        lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 15, levels, ctx=ctx, out=(u, v))
    ctx.set_option(_capi.OPT_LK_STRIP, 0)
    bu, bv = lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 15, levels, ctx=ctx)
    du = host(u).view(np.uint32) != host(bu).view(np.uint32); dv = host(v).view(np.uint32) != host(bv).view(np.uint32)
    assert not du.any() and not dv.any(), (int(du.sum()), int(dv.sum()), np.argwhere(du | dv)[:8].tolist())
    if rows * cols * levels <= 1080 * 1920 * 5:
        eu, ev = orc.lk_flow_pyr(prev[batch - 1], nxt[batch - 1], 15, levels)
        assert np.array_equal(host(u[batch - 1]), eu, equal_nan=True) and np.array_equal(host(v[batch - 1]), ev, equal_nan=True)


@pytest.mark.parametrize("rows,cols,levels,batch,win", [(270, 480, 3, 3, 43), (135, 241, 3, 2, 9), (540, 960, 4, 2, 43), (97, 400, 2, 5, 23),
                                                        (64, 64, 3, 1, 5), (300, 332, 4, 4, 13), (200, 1030, 2, 2, 63), (1080, 1920, 5, 2, 43)])
def test_generic_windows_batched(mods, rows, cols, levels, batch, win):
    """Windows without a fused kernel (config/ps5.yaml runs 43) through the batch entry point: since r05 every step of the
    generic chain is ONE launch for all pairs (blockIdx.z = pair; r04: one launch per pair on four forked streams, and the host
    was the limit) -- each pair against the oracle, odd level sizes (the expand + resize launch) and the unrolled window-43
    kernels at 1080p included (pitched buffers: test_batch_entry_point_with_pitched_buffers)."""
    lk, pyr = mods
    from introtocomputervision_amd import synth
    pairs = [synth.lk_pair(0x5EED0100 + 7 * i, rows, cols, 2 + i % 2, -1 - i % 3) for i in range(batch)]
    prev = np.stack([p for p, _ in pairs])
    nxt = np.stack([n for _, n in pairs])
    bu, bv = lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), win, levels)
    for b in (range(batch) if rows * cols <= 540 * 960 else (0, batch - 1)):
        eu, ev = orc.lk_flow_pyr(prev[b], nxt[b], win, levels)
        assert np.array_equal(host(bu[b]), eu) and np.array_equal(host(bv[b]), ev), b
    su, sv = lk.calcOpticalFlowPyr(dev(prev[batch - 1]), dev(nxt[batch - 1]), win, levels)
    assert torch.equal(su, bu[batch - 1]) and torch.equal(sv, bv[batch - 1])


@pytest.mark.parametrize("win", [9, 43, 15])
def test_batch_entry_point_with_pitched_buffers(mods, win):
    """micv_lk_flow_pyr_batch_dev on buffers with a row pitch and a pair stride that are not the dense ones (and not
    multiples of 16 bytes: the generic chain's batched launches fall back to their scalar forms where a kernel wants
    aligned rows), inputs and outputs alike: the same bits as the dense call."""
    lk, pyr = mods
    from introtocomputervision_amd import synth
    from introtocomputervision_amd._capi import lib, check, Context
    rows, cols, levels, batch = 150, 262, 3, 3
    pairs = [synth.lk_pair(0x5EED0200 + i, rows, cols, 2, -1) for i in range(batch)]
    prev = torch.from_numpy(np.stack([p for p, _ in pairs])).cuda()
    nxt = torch.from_numpy(np.stack([n for _, n in pairs])).cuda()
    ctx = Context(0)
    du, dv = lk.calcOpticalFlowPyrBatch(prev, nxt, win, levels, ctx=ctx)
    for pitch, extra in ((cols + 3, 5), (cols + 4, 8)):
        pair = rows * pitch + extra
        def wide(src=None):
            t = torch.full((batch * pair,), float("nan"), device="cuda")
            if src is not None:
                for b in range(batch):
                    t[b * pair: b * pair + rows * pitch].view(rows, pitch)[:, :cols] = src[b]
            return t
        wp, wn, wu, wv = wide(prev), wide(nxt), wide(), wide()
        check(lib.micv_lk_flow_pyr_batch_dev(ctx.handle, wp.data_ptr(), wn.data_ptr(), batch, pair * 4, rows, cols, pitch * 4,
                                             win, levels, wu.data_ptr(), wv.data_ptr(), pair * 4, pitch * 4,
                                             torch.cuda.current_stream().cuda_stream))
        for b in range(batch):
            gu = wu[b * pair: b * pair + rows * pitch].view(rows, pitch)
            gv = wv[b * pair: b * pair + rows * pitch].view(rows, pitch)
            assert torch.equal(gu[:, :cols], du[b]) and torch.equal(gv[:, :cols], dv[b]), (win, pitch, b)
            assert bool(torch.isnan(gu[:, cols:]).all()) and bool(torch.isnan(gv[:, cols:]).all())  # nothing written past a row


def test_generic_chain_chunks_a_very_large_batch(mods):
    """More pairs than one launch of the generic chain carries in blockIdx.z (4096): the batch runs in chunks -- pairs on
    both sides of the chunk boundary against the oracle, and every pair of the second chunk against its twin in the first."""
    lk, pyr = mods
    rows, cols, nb = 12, 20, 4100
    rng = np.random.default_rng(77)
    base = (rng.random((4096, rows, cols)) * 255).astype(np.float32)
    prev = np.concatenate([base, base[:nb - 4096]])
    nxt = np.roll(prev, (1, -1), (1, 2)) + np.float32(0.5)
    u, v = lk.calcOpticalFlowPyrBatch(dev(prev), dev(nxt), 5, 2)
    for b in (0, 1, 4095, 4096, 4099):
        eu, ev = orc.lk_flow_pyr(prev[b], nxt[b], 5, 2)
        assert np.array_equal(host(u[b]), eu) and np.array_equal(host(v[b]), ev), b
    assert torch.equal(u[4096:], u[:nb - 4096]) and torch.equal(v[4096:], v[:nb - 4096])


@pytest.mark.parametrize("n,shape,dtype", [(2, (48, 80), np.float32), (4, (96, 160, 3), np.uint8), (8, (135, 240), np.uint8),
                                           (6, (70, 100, 4), np.float32)])
def test_frame_sequence_matches_the_per_pair_host_calls(n, shape, dtype):
    """lk.calcOpticalFlowPyrSequence (micv_lk_flow_seq_host): flows of pairs (t, t + 1) byte-identical to the per-pair host
    entry, which the other tests pin to the oracle."""
    from introtocomputervision_amd import lk
    rng = np.random.default_rng(n * 7 + len(shape))
    base = (rng.random(shape) * 255).astype(np.float32)
    frames = []
    for t in range(n):
        f = np.roll(base, (t, 2 * t), axis=(0, 1)) + (rng.random(shape) * 3).astype(np.float32)
        frames.append(np.clip(f, 0, 255).astype(dtype))
    su, sv = lk.calcOpticalFlowPyrSequence(frames, 15, 4)
    assert su.shape == (n - 1,) + tuple(shape[:2])
    for p in range(n - 1):
        pu, pv = lk.calcOpticalFlowPyrFrames(frames[p], frames[p + 1], 15, 4)
        assert np.array_equal(su[p], pu) and np.array_equal(sv[p], pv), p
    with pytest.raises(ValueError):
        lk.calcOpticalFlowPyrSequence(frames[:1], 15, 4)
