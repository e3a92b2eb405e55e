"""The reference's common/ helpers on the GPU and the context options.

GpuTimer (common/include/common/GpuTimer.h:5-22) -> micv_timer_*; common::warmup
(common/src/CudaWarmup.cu:5-19) -> micv_warmup; micv_ctx_set_option selects between kernels that
must produce identical bits (the library reads no environment variables)."""
import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def test_gpu_timer_agrees_with_an_event_pair():
    """Elapsed time of a known-length piece of GPU work: micv_timer within 20 % of a
    torch.cuda.Event pair recorded around the same launches on the same stream."""
    from introtocomputervision_amd import lk, synth
    from introtocomputervision_amd._capi import Context, Timer
    ctx = Context(0)
    prev, nxt = synth.lk_pair(3, 540, 960, 3, -2)
    dp = torch.stack([dev(prev)] * 4)
    dn = torch.stack([dev(nxt)] * 4)
    out = (torch.empty_like(dp), torch.empty_like(dp))
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lk.calcOpticalFlowPyrBatch(dp, dn, 15, 5, ctx=ctx, out=out)
    torch.cuda.synchronize()
    t = Timer()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t.start(stream)
    for _ in range(40):
        lk.calcOpticalFlowPyrBatch(dp, dn, 15, 5, ctx=ctx, out=out)
    t.stop(stream)  # GpuTimer::stop synchronises on the stop event
    e1.record()
    e1.synchronize()
    ms_timer, ms_events = t.elapsed_ms(), e0.elapsed_time(e1)
    assert ms_timer > 0.5, ms_timer  # 40 passes over 4 x 540x960 pairs: milliseconds, not zero
    assert abs(ms_timer - ms_events) <= 0.2 * ms_events, (ms_timer, ms_events)
    # a second use of the same timer measures the new interval, not the old one
    t.start(stream)
    t.stop(stream)
    assert t.elapsed_ms() < 0.5 * ms_timer


def test_timer_on_a_side_stream_and_null_arguments():
    from introtocomputervision_amd import _capi
    s = torch.cuda.Stream()
    t = _capi.Timer()
    with torch.cuda.stream(s):
        t.start(s.cuda_stream)
        x = torch.ones(1 << 22, device="cuda")
        y = (x * 2).sum()
        t.stop(s.cuda_stream)
    assert float(y) == float(2 << 22) and t.elapsed_ms() > 0
    ms = _capi.f32()
    assert _capi.lib.micv_timer_elapsed_ms(None, _capi.C.byref(ms)) == _capi.EINVAL
    assert _capi.lib.micv_timer_start(None, None) == _capi.EINVAL
    assert _capi.lib.micv_timer_create(None) == _capi.EINVAL
    _capi.lib.micv_timer_destroy(None)  # no-op


def test_warmup_runs_on_any_stream_and_leaves_the_context_usable():
    from introtocomputervision_amd import lk, synth
    from introtocomputervision_amd._capi import Context, EINVAL, lib
    ctx = Context(0)
    ctx.warmup(None)  # default stream
    s = torch.cuda.Stream()
    ctx.warmup(s.cuda_stream)
    s.synchronize()
    torch.cuda.synchronize()
    assert lib.micv_warmup(None, None) == EINVAL
    prev, nxt = synth.lk_pair(4, 64, 96, 1, 1)
    eu, ev = orc.lk_flow(prev, nxt, 15)
    gu, gv = lk.calcOpticalFlow(dev(prev), dev(nxt), 15, ctx=ctx)
    assert np.array_equal(host(gu), eu) and np.array_equal(host(gv), ev)


def test_options_select_kernels_with_identical_results():
    from introtocomputervision_amd import _capi, harris, lk, stereo, synth
    ctx = _capi.Context(0)
    with pytest.raises(_capi.MicvError):
        ctx.set_option(99, 1)
    with pytest.raises(_capi.MicvError):
        ctx.set_option(_capi.OPT_LK_STREAM_GROUPS, 9)
    with pytest.raises(_capi.MicvError):
        ctx.set_option(_capi.OPT_STEREO_ROWS, 7)
    assert ctx.get_option(_capi.OPT_LK_STREAM_GROUPS) == 0

    # LK: stream groups 1..4 and the narrow tile form
    pairs = [synth.lk_pair(200 + i, 270, 480, 3, -2) for i in range(5)]
    prev = dev(np.stack([p for p, _ in pairs])); nxt = dev(np.stack([n for _, n in pairs]))
    exp = [orc.lk_flow_pyr(p, n, 15, 5) for p, n in pairs]
    for groups, narrow in [(0, 0), (1, 0), (3, 0), (4, 1), (2, 1)]:
        ctx.set_option(_capi.OPT_LK_STREAM_GROUPS, groups)
        ctx.set_option(_capi.OPT_LK_NARROW_TILES, narrow)
        gu, gv = lk.calcOpticalFlowPyrBatch(prev, nxt, 15, 5, ctx=ctx)
        for i in range(5):
            assert np.array_equal(host(gu[i]), exp[i][0]) and np.array_equal(host(gv[i]), exp[i][1]), (groups, narrow, i)
    ctx.set_option(_capi.OPT_LK_STREAM_GROUPS, 0)
    ctx.set_option(_capi.OPT_LK_NARROW_TILES, 0)

    # Sobel / Harris response / NMS: tiled vs generic
    img = dev(synth.checkerboard(200, 333, seed=11))
    gx0, gy0 = harris.getGradients(img, 3, ctx=ctx)
    r0 = harris.getCornerResponse(gx0, gy0, 5, 1.5, 0.04, ctx=ctx)
    c0, l0 = harris.refineCorners(r0, 5e8, 5, ctx=ctx)
    for opt in (_capi.OPT_SOBEL_GENERIC, _capi.OPT_HARRIS_GENERIC, _capi.OPT_NMS_SCAN):
        ctx.set_option(opt, 1)
    gx1, gy1 = harris.getGradients(img, 3, ctx=ctx)
    r1 = harris.getCornerResponse(gx1, gy1, 5, 1.5, 0.04, ctx=ctx)
    c1, l1 = harris.refineCorners(r1, 5e8, 5, ctx=ctx)
    for opt in (_capi.OPT_SOBEL_GENERIC, _capi.OPT_HARRIS_GENERIC, _capi.OPT_NMS_SCAN):
        ctx.set_option(opt, 0)
    assert torch.equal(gx0, gx1) and torch.equal(gy0, gy1) and torch.equal(r0, r1) and torch.equal(c0, c1)
    assert torch.equal(l0, l1) and len(l0) > 0

    # stereo: 8 or 10 rows per strip
    left, right, _ = synth.stereo_pair(5, 120, 300)
    exp_d = orc.disparity_ssd(left, right, 5, -40, 0)
    for rows in (0, 8, 10):
        ctx.set_option(_capi.OPT_STEREO_ROWS, rows)
        assert np.array_equal(host(stereo.disparitySSD(dev(left), dev(right), 5, -40, 0, ctx=ctx)), exp_d), rows


def test_phase_stamps_need_a_diagnostic_build():
    """The default build compiles the in-kernel stamps / stop-after-phase path out: asking for them
    is refused instead of silently producing garbage flow."""
    from introtocomputervision_amd import _capi
    ctx = _capi.Context(0)
    buf = (_capi.C.c_uint64 * 16)()
    rc = _capi.lib.micv_profile_lk_phases(ctx.handle, 1, buf)
    if rc == _capi.OK:  # a -DMICV_DIAG build: allowed, switch it off again
        assert _capi.lib.micv_profile_lk_phases(ctx.handle, 0, buf) == _capi.OK
    else:
        assert rc == _capi.EUNSUPPORTED and "MICV_DIAG" in _capi.last_error()
    assert _capi.lib.micv_profile_lk_phases(ctx.handle, 0, buf) == _capi.OK
