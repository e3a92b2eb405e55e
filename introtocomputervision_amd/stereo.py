"""Window stereo of the reference's ps2 (`cuda::` / `serial::` disparitySSD, disparityNCorr;
ProblemSets/ps2_cpp/include/DisparitySSD.h:18-43, DisparityNCorr.h:19-44)."""
import numpy as np

from . import _buf as B
from ._capi import STEREO_COLS_2R, STEREO_MIN_SSD_5E6, STEREO_ROLLING, STEREO_SERIAL, check, lib
from .lk import _ctx_for

# window and threshold of DisparitySSD.cu, every row's column sums formed afresh (the fast kernel)
AS_WRITTEN_CUDA = STEREO_COLS_2R | STEREO_MIN_SSD_5E6
# + the kernel's rolling subtract / add column sums down 40-row strips (DisparitySSD.cu:97-138): differs
# from AS_WRITTEN_CUDA only on images that are not integer-valued
AS_WRITTEN_CUDA_ROLLING = AS_WRITTEN_CUDA | STEREO_ROLLING


def _run(dev_fn, host_fn, left, right, windowRad, minDisparity, maxDisparity, flags, ctx):
    B.check2d(left, np.float32, name="left")
    B.check2d(right, np.float32, name="right")
    if tuple(left.shape) != tuple(right.shape) or B.stride_bytes(left) != B.stride_bytes(right):
        raise ValueError("left and right differ in size / stride")
    rows, cols = left.shape
    disp = B.empty_like_shape(left, (rows, cols), np.int8)
    c = _ctx_for(left, ctx)
    if B.is_dev(left):
        check(dev_fn(c.handle, B.ptr(left), B.ptr(right), rows, cols, B.stride_bytes(left),
                     int(windowRad), int(minDisparity), int(maxDisparity), int(flags),
                     B.ptr(disp), cols, B.stream_of(left)))
    else:
        check(host_fn(c.handle, B.ptr(left), B.ptr(right), rows, cols, B.stride_bytes(left),
                      int(windowRad), int(minDisparity), int(maxDisparity), int(flags),
                      B.ptr(disp), cols))
    return disp


def disparitySSD(left, right, windowRad, minDisparity, maxDisparity, flags=0, ctx=None):
    """disparitySSD(left, right, windowRad, minDisparity, maxDisparity) -> int8 disparity.
    flags=0: (2r+1)^2 window, CUDA-path addressing; AS_WRITTEN_CUDA: DisparitySSD.cu as written;
    STEREO_SERIAL: DisparitySSD.cpp as written."""
    return _run(lib.micv_disparity_ssd_dev, lib.micv_disparity_ssd_host, left, right, windowRad,
                minDisparity, maxDisparity, flags, ctx)


def disparityNCorr(left, right, windowRad, minDisparity, maxDisparity, flags=0, ctx=None):
    """disparityNCorr, CUDA-path semantics (DisparityNCorr.cu:60-174)."""
    return _run(lib.micv_disparity_ncorr_dev, lib.micv_disparity_ncorr_host, left, right,
                windowRad, minDisparity, maxDisparity, flags, ctx)
