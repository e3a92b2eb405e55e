"""`pyr::` namespace of the reference (ProblemSets/ps5_cpp/include/Pyramids.h:7-12)."""
import ctypes as C

import numpy as np

from . import _buf as B
from ._capi import check, lib, vp
from .lk import _ctx_for


def pyrDown(src, ctx=None):
    """pyr::pyrDown (Pyramids.cu:34-73): dst(y,x) = src(2y+1,2x+1), (rows/2) x (cols/2)."""
    B.check2d(src, np.float32, name="src")
    rows, cols = src.shape
    if rows // 2 == 0 or cols // 2 == 0:
        return B.empty_like_shape(src, (rows // 2, cols // 2))
    dst = B.empty_like_shape(src, (rows // 2, cols // 2))
    c = _ctx_for(src, ctx)
    if B.is_dev(src):
        check(lib.micv_pyr_down_dev(c.handle, B.ptr(src), rows, cols, B.stride_bytes(src),
                                    B.ptr(dst), B.stride_bytes(dst), B.stream_of(src)))
    else:
        check(lib.micv_pyr_down_host(c.handle, B.ptr(src), rows, cols, B.stride_bytes(src),
                                     B.ptr(dst), B.stride_bytes(dst)))
    return dst


def pyrUp(src, ctx=None):
    """pyr::pyrUp (Pyramids.cu:94-131): 2x replicate + separable [1,4,6,4,1]/16."""
    B.check2d(src, np.float32, name="src")
    rows, cols = src.shape
    dst = B.empty_like_shape(src, (rows * 2, cols * 2))
    c = _ctx_for(src, ctx)
    if B.is_dev(src):
        check(lib.micv_pyr_up_dev(c.handle, B.ptr(src), rows, cols, B.stride_bytes(src),
                                  B.ptr(dst), B.stride_bytes(dst), B.stream_of(src)))
    else:
        check(lib.micv_pyr_up_host(c.handle, B.ptr(src), rows, cols, B.stride_bytes(src),
                                   B.ptr(dst), B.stride_bytes(dst)))
    return dst


def makeGaussianPyramid(src, levels, ctx=None):
    """pyr::makeGaussianPyramid (Pyramids.cpp:5-26) on a grey f32 image -> list of levels."""
    B.check2d(src, np.float32, name="src")
    rows, cols = src.shape
    outs = [B.empty_like_shape(src, (rows >> l, cols >> l)) for l in range(levels)]
    arr = (vp * levels)(*[B.ptr(o) for o in outs])
    c = _ctx_for(src, ctx)
    if B.is_dev(src):
        check(lib.micv_gaussian_pyramid_dev(c.handle, B.ptr(src), rows, cols, B.stride_bytes(src),
                                            int(levels), arr, B.stream_of(src)))
    else:
        check(lib.micv_gaussian_pyramid_host(c.handle, B.ptr(src), rows, cols, B.stride_bytes(src),
                                             int(levels), arr))
    return outs


def makeLaplacianPyramid(src, levels, ctx=None):
    """The Laplacian pyramid built by sol::runProblem2 (ps5_cpp/src/Solution.cpp:187-200) from a grey
    f32 CUDA tensor -> list of levels (device tensors only)."""
    B.check2d(src, np.float32, name="src")
    if not B.is_dev(src):
        raise ValueError("makeLaplacianPyramid: device tensors only")
    rows, cols = src.shape
    outs = [B.empty_like_shape(src, (rows >> l, cols >> l)) for l in range(levels)]
    arr = (vp * levels)(*[B.ptr(o) for o in outs])
    check(lib.micv_laplacian_pyramid_dev(_ctx_for(src, ctx).handle, B.ptr(src), rows, cols,
                                         B.stride_bytes(src), int(levels), arr, B.stream_of(src)))
    return outs


def resizeLinear(src, drows, dcols, ctx=None):
    """cv::resize(src, dst, Size(dcols, drows)) INTER_LINEAR as used at OpticalFlow.cpp:149-150
    (device tensors only)."""
    B.check2d(src, np.float32, name="src")
    if not B.is_dev(src):
        raise ValueError("resizeLinear: device tensors only")
    rows, cols = src.shape
    dst = B.empty_like_shape(src, (drows, dcols))
    c = _ctx_for(src, ctx)
    check(lib.micv_resize_linear_dev(c.handle, B.ptr(src), rows, cols, B.stride_bytes(src),
                                     B.ptr(dst), drows, dcols, B.stride_bytes(dst),
                                     B.stream_of(src)))
    return dst


def rgb8ToGray(rgb, ctx=None):
    """cvtColor(COLOR_RGB2GRAY) + convertTo(CV_32F) of Pyramids.cpp:10-15 on an [rows, cols, 3]
    uint8 CUDA tensor."""
    import torch
    if not (B.is_dev(rgb) and rgb.is_cuda and rgb.dim() == 3 and rgb.shape[2] == 3
            and rgb.dtype == torch.uint8 and rgb.is_contiguous()):
        raise ValueError("rgb8ToGray: need a contiguous [rows, cols, 3] uint8 CUDA tensor")
    rows, cols, _ = rgb.shape
    dst = torch.empty((rows, cols), dtype=torch.float32, device=rgb.device)
    c = _ctx_for(dst, ctx)
    check(lib.micv_rgb8_to_gray_f32_dev(c.handle, rgb.data_ptr(), rows, cols, cols * 3,
                                        dst.data_ptr(), cols * 4,
                                        torch.cuda.current_stream(rgb.device).cuda_stream))
    return dst
