"""`cuda::` Hough functions of the reference's ps1 (ProblemSets/ps1_cpp/src/Hough.h:22-84)."""
import ctypes as C

import numpy as np

from . import _buf as B
from ._capi import check, i32, i64, lib
from .lk import _ctx_for


def linesAccumulatorShape(rows, cols, rhoBinSize=1, thetaBinSize=1):
    rb, tb = i32(), i32()
    check(lib.micv_hough_lines_dims(int(rows), int(cols), int(rhoBinSize), int(thetaBinSize),
                                    C.byref(rb), C.byref(tb)))
    return rb.value, tb.value


def houghLinesAccumulate(edgeMask, rhoBinSize=1, thetaBinSize=1, ctx=None):
    """cuda::houghLinesAccumulate (Hough.cu:251-309) -> int32 accumulator [rhoBins, thetaBins]."""
    B.check2d(edgeMask, np.uint8, name="edgeMask")
    rows, cols = edgeMask.shape
    rb, tb = linesAccumulatorShape(rows, cols, rhoBinSize, thetaBinSize)
    acc = B.empty_like_shape(edgeMask, (rb, tb), np.int32)
    c = _ctx_for(edgeMask, ctx)
    if B.is_dev(edgeMask):
        check(lib.micv_hough_lines_dev(c.handle, B.ptr(edgeMask), rows, cols,
                                       B.stride_bytes(edgeMask), int(rhoBinSize),
                                       int(thetaBinSize), B.ptr(acc), B.stream_of(edgeMask)))
    else:
        check(lib.micv_hough_lines_host(c.handle, B.ptr(edgeMask), rows, cols,
                                        B.stride_bytes(edgeMask), int(rhoBinSize),
                                        int(thetaBinSize), B.ptr(acc)))
    return acc


def houghCirclesAccumulate(edgeMask, radius, ctx=None):
    """cuda::houghCirclesAccumulate (Hough.cu:311-364) -> int32 accumulator [rows, cols]."""
    B.check2d(edgeMask, np.uint8, name="edgeMask")
    rows, cols = edgeMask.shape
    acc = B.empty_like_shape(edgeMask, (rows, cols), np.int32)
    c = _ctx_for(edgeMask, ctx)
    if B.is_dev(edgeMask):
        check(lib.micv_hough_circles_dev(c.handle, B.ptr(edgeMask), rows, cols,
                                         B.stride_bytes(edgeMask), int(radius), B.ptr(acc),
                                         B.stream_of(edgeMask)))
    else:
        check(lib.micv_hough_circles_host(c.handle, B.ptr(edgeMask), rows, cols,
                                          B.stride_bytes(edgeMask), int(radius), B.ptr(acc)))
    return acc


def findLocalMaxima(accumulator, numPeaks, threshold, ctx=None, lazy=False):
    """cuda::findLocalMaxima (Hough.cu:366-426) -> [n, 2] uint32 (row, col) pairs ordered by
    votes descending (stable).  Device input with lazy=True: returns (peaks[numPeaks, 2], count) as
    device tensors without reading the count back (no host synchronisation; rows >= count are
    unspecified) -- for pipelines that keep consuming on the device."""
    B.check2d(accumulator, np.int32, name="accumulator")
    if B.is_dev(accumulator) and not accumulator.is_contiguous():
        raise ValueError("accumulator must be contiguous")
    if not B.is_dev(accumulator) and not accumulator.flags.c_contiguous:
        raise ValueError("accumulator must be contiguous")
    rows, cols = accumulator.shape
    c = _ctx_for(accumulator, ctx)
    k = int(numPeaks)
    if B.is_dev(accumulator):
        import torch
        peaks = torch.empty((max(k, 1), 2), dtype=torch.int32, device=accumulator.device)
        cnt = torch.zeros((1,), dtype=torch.int64, device=accumulator.device) if lazy else B.pinned_count(accumulator)
        check(lib.micv_hough_peaks_dev(c.handle, B.ptr(accumulator), rows, cols, k, int(threshold),
                                       peaks.data_ptr(), cnt.data_ptr(), B.stream_of(accumulator)))
        if lazy:
            return peaks, cnt
        return peaks[:B.read_count(cnt, accumulator)]
    peaks = np.empty((max(k, 1), 2), np.uint32)
    cnt = i64(0)
    check(lib.micv_hough_peaks_host(c.handle, B.ptr(accumulator), rows, cols, k, int(threshold),
                                    peaks.ctypes.data, C.byref(cnt)))
    return peaks[:cnt.value]


def generateEdge(image, gaussianSize, gaussianSigma, lowerThreshold, upperThreshold, ctx=None):
    """sol::generateEdge (ps1_cpp/src/Solution.cpp:21-47) on a 2-D uint8 CUDA tensor -> 255/0 edge
    mask (Gaussian blur + Canny, aperture 3)."""
    if isinstance(image, np.ndarray):  # host-pointer entry point
        img = np.ascontiguousarray(image, np.uint8)
        if img.ndim != 2:
            raise ValueError("image: need a 2-D uint8 array")
        out = np.empty_like(img)
        from .match import _host_ctx
        check(lib.micv_generate_edge_host((ctx or _host_ctx()).handle, img.ctypes.data, img.shape[0], img.shape[1],
                                          img.strides[0], int(gaussianSize), float(gaussianSigma),
                                          float(lowerThreshold), float(upperThreshold), out.ctypes.data,
                                          out.strides[0]))
        return out
    import torch
    if not (B.is_dev(image) and image.is_cuda and image.dim() == 2 and image.dtype == torch.uint8
            and image.stride(1) == 1):
        raise ValueError("image: need a 2-D uint8 CUDA tensor with unit column stride")
    rows, cols = image.shape
    edges = torch.empty((rows, cols), dtype=torch.uint8, device=image.device)
    check(lib.micv_generate_edge_dev(_ctx_for(image, ctx).handle, image.data_ptr(), rows, cols,
                                     image.stride(0), int(gaussianSize), float(gaussianSigma),
                                     float(lowerThreshold), float(upperThreshold), edges.data_ptr(), cols,
                                     B.stream_of(image)))
    return edges
