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


def toGray(img, ctx=None):
    """The colour branch of pyr::makeGaussianPyramid (Pyramids.cpp:9-15): cvtColor(COLOR_RGB2GRAY)
    for 3/4-channel input, then convertTo(CV_32F).  img: [rows, cols] or [rows, cols, 3|4], uint8 or
    float32, numpy (host entry point) or a contiguous CUDA tensor (device entry point)."""
    if B.is_dev(img):
        import torch
        if not (img.is_contiguous() and img.dim() in (2, 3) and img.dtype in (torch.uint8, torch.float32)):
            raise ValueError("toGray: need a contiguous uint8 / float32 CUDA tensor")
        cn = 1 if img.dim() == 2 else int(img.shape[2])
        rows, cols = int(img.shape[0]), int(img.shape[1])
        es = 1 if img.dtype == torch.uint8 else 4
        dst = torch.empty((rows, cols), dtype=torch.float32, device=img.device)
        check(lib.micv_to_gray_f32_dev(_ctx_for(dst, ctx).handle, img.data_ptr(), rows, cols, cols * cn * es, cn,
                                       0 if es == 1 else 5, dst.data_ptr(), cols * 4,
                                       torch.cuda.current_stream(img.device).cuda_stream))
        return dst
    img = np.ascontiguousarray(img)
    if img.dtype not in (np.uint8, np.float32) or img.ndim not in (2, 3):
        raise ValueError("toGray: need a uint8 / float32 array of 2 or 3 dimensions")
    cn = 1 if img.ndim == 2 else img.shape[2]
    rows, cols = img.shape[:2]
    dst = np.empty((rows, cols), np.float32)
    check(lib.micv_to_gray_f32_host(_ctx_for(dst, ctx).handle, img.ctypes.data, rows, cols,
                                    cols * cn * img.dtype.itemsize, cn, 0 if img.dtype == np.uint8 else 5,
                                    dst.ctypes.data, cols * 4))
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
