"""Error behaviour of the C ABI (SURVEY.md §8b "Errors"): bad arguments come back as status codes
with a message (the reference asserts / exits), nothing is written, and the context stays usable."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_status_codes_and_messages():
    from introtocomputervision_amd import harris, hough, match, stereo
    from introtocomputervision_amd._capi import EINVAL, MicvError, last_error
    f = dev(np.ones((40, 50), np.float32))
    m = dev(np.zeros((40, 50), np.uint8))
    cases = [
        lambda: harris.getCornerResponse(f, f, 4, 1.5, 0.04),          # even window
        lambda: harris.getCornerResponse(f, f, 5, 0.0, 0.04),          # sigma must be > 0
        lambda: harris.getGradients(f, 4),                              # even Sobel size
        lambda: harris.refineCorners(f, 0.5, -1),                       # negative distance
        lambda: stereo.disparitySSD(f, f, 5, -200, 0),                  # does not fit CV_8SC1
        lambda: stereo.disparitySSD(f, f, 40, -5, 0),                   # radius out of range
        lambda: stereo.disparityNCorr(f, f, 3, 5, 1),                   # min > max
        lambda: hough.houghLinesAccumulate(m, 0, 1),                    # zero bin size
        lambda: hough.findLocalMaxima(dev(np.zeros((8, 8), np.int32)), 5000, 1),  # too many peaks
        lambda: match.knnMatch2(dev(np.ones((4, 8), np.float32)), dev(np.ones((1, 8), np.float32))),  # k=2 needs 2 rows
    ]
    for i, fn in enumerate(cases):
        with pytest.raises(MicvError) as e:
            fn()
        assert e.value.code == EINVAL, (i, e.value)
        assert len(last_error()) > 10
    # the context is still usable afterwards
    R = harris.getCornerResponse(f, f, 5, 1.5, 0.04)
    assert torch.isfinite(R).all()


def test_null_pointers_are_refused():
    from introtocomputervision_amd._capi import EINVAL, Context, lib
    ctx = Context(0)
    assert lib.micv_lk_flow_pyr_dev(ctx.handle, None, None, 8, 8, 32, 15, 1, None, None, 32, None) == EINVAL
    assert lib.micv_sobel_dev(ctx.handle, None, 8, 8, 32, 3, 1.0, None, None, 32, None) == EINVAL
    assert lib.micv_hough_circles_dev(ctx.handle, None, 8, 8, 8, 3, None, None) == EINVAL
    assert lib.micv_lk_flow_pyr_dev(None, None, None, 8, 8, 32, 15, 1, None, None, 32, None) == EINVAL
    n = C.c_int()
    assert lib.micv_hough_lines_dims(0, 5, 1, 1, C.byref(n), C.byref(n)) == EINVAL
