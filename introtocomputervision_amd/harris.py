"""`harris::` and `sift::` namespaces of the reference (ProblemSets/ps4_cpp/include/Harris.h,
Descriptors.h).  `harris.cpu.*` and `harris.gpu.*` of the reference share one implementation
here (their results are identical by specification)."""
import ctypes as C

import numpy as np

from . import _buf as B
from ._capi import check, i64, lib
from .lk import _ctx_for


def getGradients(img, kernelSize=3, scale=1.0, ctx=None):
    """harris::getGradients (Harris.cpp:14-41) -> (diffX, diffY).  scale = 1/9 gives the LK
    computeGradients (OpticalFlow.cpp:12-39)."""
    B.check2d(img, np.float32, name="in")
    rows, cols = img.shape
    gx = B.empty_like_shape(img, (rows, cols))
    gy = B.empty_like_shape(img, (rows, cols))
    c = _ctx_for(img, ctx)
    if B.is_dev(img):
        check(lib.micv_sobel_dev(c.handle, B.ptr(img), rows, cols, B.stride_bytes(img),
                                 int(kernelSize), float(scale), B.ptr(gx), B.ptr(gy),
                                 B.stride_bytes(gx), B.stream_of(img)))
    else:
        check(lib.micv_sobel_host(c.handle, B.ptr(img), rows, cols, B.stride_bytes(img),
                                  int(kernelSize), float(scale), B.ptr(gx), B.ptr(gy),
                                  B.stride_bytes(gx)))
    return gx, gy


HARRIS_CPU = 1  # mi_cv.h MICV_HARRIS_CPU


def getCornerResponse(gradX, gradY, windowSize, gaussianSigma, harrisScore, ctx=None, cpu_arithmetic=False):
    """harris::{cpu,gpu}::getCornerResponse (Harris.cpp:43-97 / Harris.cu:96-159) -> R.
    cpu_arithmetic: harris::cpu's arithmetic as written (Harris.cpp:78-92; MICV_HARRIS_CPU) instead of
    harris::gpu's (the default, as in the reference's configuration)."""
    B.check2d(gradX, np.float32, name="gradX")
    B.check2d(gradY, np.float32, name="gradY")
    if tuple(gradX.shape) != tuple(gradY.shape) or B.stride_bytes(gradX) != B.stride_bytes(gradY):
        raise ValueError("gradX and gradY differ in size / stride")
    rows, cols = gradX.shape
    resp = B.empty_like_shape(gradX, (rows, cols))
    c = _ctx_for(gradX, ctx)
    flags = HARRIS_CPU if cpu_arithmetic else 0
    if B.is_dev(gradX):
        check(lib.micv_harris_response_ex_dev(c.handle, B.ptr(gradX), B.ptr(gradY), rows, cols,
                                              B.stride_bytes(gradX), int(windowSize),
                                              float(gaussianSigma), float(harrisScore), flags, B.ptr(resp),
                                              B.stride_bytes(resp), B.stream_of(gradX)))
    else:
        check(lib.micv_harris_response_ex_host(c.handle, B.ptr(gradX), B.ptr(gradY), rows, cols,
                                               B.stride_bytes(gradX), int(windowSize),
                                               float(gaussianSigma), float(harrisScore), flags, B.ptr(resp),
                                               B.stride_bytes(resp)))
    return resp


def refineCorners(cornerResponse, threshold, minDistance, capacity=None, ctx=None, lazy=False):
    """harris::{cpu,gpu}::refineCorners (Harris.cpp:99-147 / Harris.cu:243-329) ->
    (corners, cornerLocs) with cornerLocs an [n, 2] int32 array of (y, x) in row-major order.
    Device input with lazy=True: returns (corners, locs[capacity, 2], count) without reading the
    count back (no host synchronisation)."""
    B.check2d(cornerResponse, np.float32, name="cornerResponse")
    rows, cols = cornerResponse.shape
    cap = int(capacity) if capacity is not None else rows * cols
    corners = B.empty_like_shape(cornerResponse, (rows, cols))
    c = _ctx_for(cornerResponse, ctx)
    if B.is_dev(cornerResponse):
        import torch
        locs = torch.empty((cap, 2), dtype=torch.int32, device=cornerResponse.device)
        # the count: a device word when the caller keeps it on the device, else a pinned host word the kernel writes
        cnt = torch.zeros((1,), dtype=torch.int64, device=cornerResponse.device) if lazy else B.pinned_count(cornerResponse)
        check(lib.micv_harris_refine_dev(c.handle, B.ptr(cornerResponse), rows, cols,
                                         B.stride_bytes(cornerResponse), float(threshold),
                                         int(minDistance), B.ptr(corners), B.stride_bytes(corners),
                                         locs.data_ptr(), cap, cnt.data_ptr(),
                                         B.stream_of(cornerResponse)))
        if lazy:
            return corners, locs, cnt
        n = min(B.read_count(cnt, cornerResponse), cap)
        return corners, locs[:n]
    locs = np.empty((cap, 2), np.int32)
    cnt = i64(0)
    check(lib.micv_harris_refine_host(c.handle, B.ptr(cornerResponse), rows, cols,
                                      B.stride_bytes(cornerResponse), float(threshold),
                                      int(minDistance), B.ptr(corners), B.stride_bytes(corners),
                                      locs.ctypes.data, cap, C.byref(cnt)))
    return corners, locs[:min(cnt.value, cap)]


def cornersFromImage(img, sobelSize=3, windowSize=5, gaussianSigma=1.5, harrisScore=0.04, threshold=5e8, minDistance=5,
                     capacity=None, ctx=None, cpu_arithmetic=False, want_gradients=True, want_response=False,
                     want_corners=False, lazy=False):
    """The ps4 caller's chain (harrisHelper, ps4_cpp/src/Solution.cpp:77-124: harris::getGradients -> getCornerResponse
    -> refineCorners) as ONE call (micv_harris_corners_dev / _host): with a 3x3 Sobel and windows 3 / 5 / 7 the gradients
    are formed inside the response kernel's tile.  Same bits as the three separate calls.
    Returns a dict: "locs" ([n, 2] int32 (y, x), row-major order), and as requested "gx", "gy", "response", "corners";
    device input with lazy=True: "locs" has `capacity` rows and "count" is a device word (no host synchronisation)."""
    B.check2d(img, np.float32, name="img")
    rows, cols = img.shape
    cap = int(capacity) if capacity is not None else rows * cols
    c = _ctx_for(img, ctx)
    flags = HARRIS_CPU if cpu_arithmetic else 0
    gx = B.empty_like_shape(img, (rows, cols)) if want_gradients else None
    gy = B.empty_like_shape(img, (rows, cols)) if want_gradients else None
    resp = B.empty_like_shape(img, (rows, cols)) if want_response else None
    corners = B.empty_like_shape(img, (rows, cols)) if want_corners else None
    rb = cols * 4

    def p(a):
        return B.ptr(a) if a is not None else None

    def sb(a):
        return B.stride_bytes(a) if a is not None else rb
    out = {}
    if B.is_dev(img):
        import torch
        locs = torch.empty((cap, 2), dtype=torch.int32, device=img.device)
        cnt = torch.zeros((1,), dtype=torch.int64, device=img.device) if lazy else B.pinned_count(img)
        check(lib.micv_harris_corners_dev(c.handle, B.ptr(img), rows, cols, B.stride_bytes(img), int(sobelSize), int(windowSize),
                                          float(gaussianSigma), float(harrisScore), flags, float(threshold), int(minDistance),
                                          p(gx), p(gy), sb(gx), p(resp), sb(resp), p(corners), sb(corners), locs.data_ptr(), cap,
                                          cnt.data_ptr(), B.stream_of(img)))
        if lazy:
            out["locs"], out["count"] = locs, cnt
        else:
            out["locs"] = locs[:min(B.read_count(cnt, img), cap)]
    else:
        locs = np.empty((cap, 2), np.int32)
        cnt = i64(0)
        check(lib.micv_harris_corners_host(c.handle, B.ptr(img), rows, cols, B.stride_bytes(img), int(sobelSize), int(windowSize),
                                           float(gaussianSigma), float(harrisScore), flags, float(threshold), int(minDistance),
                                           p(gx), p(gy), sb(gx), p(resp), sb(resp), p(corners), sb(corners), locs.ctypes.data, cap,
                                           C.byref(cnt)))
        out["locs"] = locs[:min(cnt.value, cap)]
    if want_gradients:
        out["gx"], out["gy"] = gx, gy
    if want_response:
        out["response"] = resp
    if want_corners:
        out["corners"] = corners
    return out


# The reference's two namespaces, for code that spells them out.
class cpu:  # noqa: N801
    getCornerResponse = staticmethod(getCornerResponse)
    refineCorners = staticmethod(refineCorners)


gpu = cpu


def _check_grad_pair(gradX, gradY, *device_lists):
    """gradX / gradY go to the kernels with ONE shape and pitch: both must agree (and live on one device,
    together with any list the kernel indexes them by)."""
    B.check2d(gradX, np.float32, name="gradX")
    B.check2d(gradY, np.float32, name="gradY")
    if B.is_dev(gradX) != B.is_dev(gradY):
        raise ValueError("gradX and gradY: one is a device tensor, the other is not")
    if tuple(gradX.shape) != tuple(gradY.shape) or B.stride_bytes(gradX) != B.stride_bytes(gradY):
        raise ValueError("gradX and gradY differ in size / stride")
    if B.is_dev(gradX):
        if gradY.device != gradX.device:
            raise ValueError("gradX and gradY live on different devices")
        for name, t in device_lists:
            if not (B.is_dev(t) and t.is_cuda and t.device == gradX.device):
                raise ValueError(f"{name}: must be a CUDA tensor on gradX's device (numpy inputs take the host path)")


def getAnglesFromGradients(gradX, gradY, ctx=None):
    """sift::getAnglesFromGradients (Descriptors.cpp:7-25) -> angles (radians)."""
    _check_grad_pair(gradX, gradY)
    rows, cols = gradX.shape
    ang = B.empty_like_shape(gradX, (rows, cols))
    c = _ctx_for(gradX, ctx)
    if B.is_dev(gradX):
        check(lib.micv_sift_angles_dev(c.handle, B.ptr(gradX), B.ptr(gradY), rows, cols,
                                       B.stride_bytes(gradX), B.ptr(ang), B.stride_bytes(ang),
                                       B.stream_of(gradX)))
    else:
        check(lib.micv_sift_angles_host(c.handle, B.ptr(gradX), B.ptr(gradY), rows, cols,
                                        B.stride_bytes(gradX), B.ptr(ang), B.stride_bytes(ang)))
    return ang


def getKeypoints(gradX, gradY, cornerLocs, size, ctx=None):
    """sift::getKeypoints (Descriptors.cpp:27-47) -> [n, 4] float32 (x, y, size, angle_deg):
    the fields of the cv::KeyPoint the reference constructs."""
    _check_grad_pair(gradX, gradY, ("cornerLocs", cornerLocs)) if B.is_dev(gradX) else _check_grad_pair(gradX, gradY)
    rows, cols = gradX.shape
    c = _ctx_for(gradX, ctx)
    n = int(cornerLocs.shape[0])
    if B.is_dev(gradX):
        import torch
        locs = cornerLocs.to(torch.int32).contiguous()
        kp = torch.empty((n, 4), dtype=torch.float32, device=gradX.device)
        check(lib.micv_sift_keypoints_dev(c.handle, B.ptr(gradX), B.ptr(gradY), rows, cols,
                                          B.stride_bytes(gradX), locs.data_ptr(), n, float(size),
                                          kp.data_ptr(), B.stream_of(gradX)))
        return kp
    locs = np.ascontiguousarray(cornerLocs, dtype=np.int32)
    kp = np.empty((n, 4), np.float32)
    check(lib.micv_sift_keypoints_host(c.handle, B.ptr(gradX), B.ptr(gradY), rows, cols,
                                       B.stride_bytes(gradX), locs.ctypes.data, n, float(size),
                                       kp.ctypes.data))
    return kp


def computeDescriptors(gradX, gradY, keypoints, ctx=None):
    """The descriptor step of Solution::siftHelper (ps4_cpp/src/Solution.cpp:166-169,
    cv::xfeatures2d::SIFT::compute): [n, 4] keypoints (x, y, size, angle_deg) as getKeypoints returns
    them -> [n, 128] float32 descriptors (4 x 4 x 8 bins, 8-bit values).  Arithmetic: DESIGN.md §2."""
    _check_grad_pair(gradX, gradY, ("keypoints", keypoints)) if B.is_dev(gradX) else _check_grad_pair(gradX, gradY)
    rows, cols = gradX.shape
    c = _ctx_for(gradX, ctx)
    if B.is_dev(gradX):
        import torch
        kp = keypoints.to(torch.float32).contiguous().reshape(-1, 4)
        n = int(kp.shape[0])
        desc = torch.empty((n, 128), dtype=torch.float32, device=gradX.device)
        check(lib.micv_sift_descriptors_dev(c.handle, B.ptr(gradX), B.ptr(gradY), rows, cols,
                                            B.stride_bytes(gradX), kp.data_ptr(), n, desc.data_ptr(), 512,
                                            B.stream_of(gradX)))
        return desc
    kp = np.ascontiguousarray(keypoints, dtype=np.float32).reshape(-1, 4)
    n = kp.shape[0]
    desc = np.empty((n, 128), np.float32)
    check(lib.micv_sift_descriptors_host(c.handle, B.ptr(gradX), B.ptr(gradY), rows, cols,
                                         B.stride_bytes(gradX), kp.ctypes.data, n, desc.ctypes.data, 512))
    return desc
