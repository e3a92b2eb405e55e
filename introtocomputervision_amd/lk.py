"""`lk::` namespace of the reference (ProblemSets/ps5_cpp/include/OpticalFlow.h:5-19).

Same names, argument meaning and defaults (winSize = 21); `levels` exposes the pyramid depth
the reference hard-codes to 4 (OpticalFlow.cpp:127).  numpy inputs -> host entry points,
torch CUDA tensors -> device entry points on the current stream.
"""
import numpy as np

from . import _buf as B
from ._capi import Context, check, lib

from collections import OrderedDict

_default_ctx = OrderedDict()
_DEFAULT_CTX_MAX = 8


def default_context(device=0, stream=0):
    """One default context per (device, stream): a context owns ONE scratch arena and its auxiliary
    streams, so launches from two streams must not share it (they would race on the arena with no
    error).  The cache holds the 8 most recently used pairs; beyond that the least recently used context
    that NOBODY ELSE HOLDS is destroyed (its arena, streams and cached device blocks are freed), so a caller
    that runs under many short-lived torch streams does not grow device memory without bound.  A context a
    caller still references is never closed under it (the cache grows instead), and a closed Context raises
    on use.  Long-running callers pass their own Context."""
    import sys
    key = (int(device), int(stream or 0))
    ctx = _default_ctx.get(key)
    if ctx is None:
        ctx = Context(device)
        _default_ctx[key] = ctx
        if len(_default_ctx) > _DEFAULT_CTX_MAX:
            for k in list(_default_ctx)[:-1]:  # oldest first, never the one just made
                old = _default_ctx[k]
                if sys.getrefcount(old) <= 3:  # the cache, `old`, getrefcount's argument: no outside holder
                    del _default_ctx[k]
                    old.close()  # micv_ctx_destroy synchronises the device before freeing what launches may still use
                    if len(_default_ctx) <= _DEFAULT_CTX_MAX:
                        break
    else:
        _default_ctx.move_to_end(key)
    return ctx


def _ctx_for(a, ctx):
    if ctx is not None:
        return ctx
    if B.is_dev(a):
        return default_context(a.device.index or 0, B.stream_of(a))
    return default_context(0)


def calcOpticalFlow(prevImg, nextImg, winSize=21, ctx=None):
    """lk::calcOpticalFlow (OpticalFlow.cpp:41-104) -> (u, v)."""
    B.check2d(prevImg, np.float32, name="prevImg")
    B.check2d(nextImg, np.float32, name="nextImg")
    if tuple(prevImg.shape) != tuple(nextImg.shape):
        raise ValueError("prevImg and nextImg differ in size")
    if B.stride_bytes(prevImg) != B.stride_bytes(nextImg):
        raise ValueError("prevImg and nextImg need the same row stride")
    rows, cols = prevImg.shape
    u = B.empty_like_shape(prevImg, (rows, cols))
    v = B.empty_like_shape(prevImg, (rows, cols))
    c = _ctx_for(prevImg, ctx)
    if B.is_dev(prevImg):
        check(lib.micv_lk_flow_dev(c.handle, B.ptr(prevImg), B.ptr(nextImg), rows, cols,
                                   B.stride_bytes(prevImg), int(winSize), B.ptr(u), B.ptr(v),
                                   B.stride_bytes(u), B.stream_of(prevImg)))
    else:
        check(lib.micv_lk_flow_host(c.handle, B.ptr(prevImg), B.ptr(nextImg), rows, cols,
                                    B.stride_bytes(prevImg), int(winSize), B.ptr(u), B.ptr(v),
                                    B.stride_bytes(u)))
    return u, v


def warp(src, du, dv, ctx=None):
    """lk::warp (OpticalFlow.cpp:106-120) -> dst."""
    for a, n in ((src, "src"), (du, "du"), (dv, "dv")):
        B.check2d(a, np.float32, name=n)
    if not (tuple(src.shape) == tuple(du.shape) == tuple(dv.shape)):
        raise ValueError("src, du, dv differ in size")
    if B.stride_bytes(du) != B.stride_bytes(dv):
        raise ValueError("du and dv need the same row stride")
    rows, cols = src.shape
    dst = B.empty_like_shape(src, (rows, cols))
    c = _ctx_for(src, ctx)
    if B.is_dev(src):
        check(lib.micv_lk_warp_dev(c.handle, B.ptr(src), B.stride_bytes(src), B.ptr(du), B.ptr(dv),
                                   B.stride_bytes(du), rows, cols, B.ptr(dst), B.stride_bytes(dst),
                                   B.stream_of(src)))
    else:
        check(lib.micv_lk_warp_host(c.handle, B.ptr(src), B.stride_bytes(src), B.ptr(du), B.ptr(dv),
                                    B.stride_bytes(du), rows, cols, B.ptr(dst), B.stride_bytes(dst)))
    return dst


def calcOpticalFlowPyr(prevImg, nextImg, winSize=21, levels=4, ctx=None):
    """lk::calcOpticalFlowPyr (OpticalFlow.cpp:122-167) -> (u, v); levels=4 is the reference."""
    B.check2d(prevImg, np.float32, name="prevImg")
    B.check2d(nextImg, np.float32, name="nextImg")
    if tuple(prevImg.shape) != tuple(nextImg.shape):
        raise ValueError("prevImg and nextImg differ in size")
    if B.stride_bytes(prevImg) != B.stride_bytes(nextImg):
        raise ValueError("prevImg and nextImg need the same row stride")
    rows, cols = prevImg.shape
    u = B.empty_like_shape(prevImg, (rows, cols))
    v = B.empty_like_shape(prevImg, (rows, cols))
    c = _ctx_for(prevImg, ctx)
    if B.is_dev(prevImg):
        check(lib.micv_lk_flow_pyr_dev(c.handle, B.ptr(prevImg), B.ptr(nextImg), rows, cols,
                                       B.stride_bytes(prevImg), int(winSize), int(levels),
                                       B.ptr(u), B.ptr(v), B.stride_bytes(u),
                                       B.stream_of(prevImg)))
    else:
        check(lib.micv_lk_flow_pyr_host(c.handle, B.ptr(prevImg), B.ptr(nextImg), rows, cols,
                                        B.stride_bytes(prevImg), int(winSize), int(levels),
                                        B.ptr(u), B.ptr(v), B.stride_bytes(u)))
    return u, v


def calcOpticalFlowPyrFrames(prevImg, nextImg, winSize=21, levels=4, ctx=None):
    """lk::calcOpticalFlowPyr as the unchanged ps5 caller uses it (Solution.cpp:63): the frames may be
    colour ([rows, cols, 3|4]) and 8-bit; makeGaussianPyramid's conversion (Pyramids.cpp:9-15) runs on
    the device after ONE upload.  Host (numpy) frames only -> (u, v)."""
    a, b = np.ascontiguousarray(prevImg), np.ascontiguousarray(nextImg)
    if a.shape != b.shape or a.dtype != b.dtype or a.dtype not in (np.uint8, np.float32) or a.ndim not in (2, 3):
        raise ValueError("prevImg / nextImg: equal-shape uint8 or float32 frames expected")
    cn = 1 if a.ndim == 2 else a.shape[2]
    rows, cols = a.shape[:2]
    u = np.empty((rows, cols), np.float32)
    v = np.empty((rows, cols), np.float32)
    c = ctx if ctx is not None else default_context(0)
    check(lib.micv_lk_flow_pyr_frames_host(c.handle, a.ctypes.data, b.ctypes.data, rows, cols,
                                           cols * cn * a.dtype.itemsize, cn, 0 if a.dtype == np.uint8 else 5,
                                           int(winSize), int(levels), u.ctypes.data, v.ctypes.data, cols * 4))
    return u, v


def calcOpticalFlowPyrSequence(frames, winSize=21, levels=4, ctx=None, out=None):
    """lk::calcOpticalFlowPyr over consecutive frames -- pairs (0, 1), (1, 2), ... -- as the ps5 driver walks a directory
    of frames (ps5_cpp/src/Solution.cpp:255-285).  `frames`: a list of equal-shape host (numpy) frames, uint8 or float32,
    grey or colour.  Every frame is uploaded once; upload, pyramid chain and download of consecutive pairs overlap
    (micv_lk_flow_seq_host).  Returns (u, v) as [len(frames) - 1, rows, cols] float32 arrays (`out`, when given),
    byte-identical to the per-pair calcOpticalFlowPyrFrames results."""
    import ctypes as C
    fs = [np.ascontiguousarray(f) for f in frames]
    if len(fs) < 2:
        raise ValueError("frames: at least two frames expected")
    a = fs[0]
    if a.dtype not in (np.uint8, np.float32) or a.ndim not in (2, 3) or any(f.shape != a.shape or f.dtype != a.dtype for f in fs):
        raise ValueError("frames: equal-shape uint8 or float32 frames expected")
    cn = 1 if a.ndim == 2 else a.shape[2]
    rows, cols = a.shape[:2]
    n = len(fs) - 1
    if out is None:
        u = np.empty((n, rows, cols), np.float32)
        v = np.empty((n, rows, cols), np.float32)
    else:  # the caller's buffers (already touched pages: a fresh allocation is faulted in page by page during the download)
        u, v = out
        for a_, name in ((u, "out[0]"), (v, "out[1]")):
            if not (isinstance(a_, np.ndarray) and a_.dtype == np.float32 and a_.shape == (n, rows, cols) and a_.flags.c_contiguous):
                raise ValueError(f"{name}: a C-contiguous float32 array of shape {(n, rows, cols)} expected")
    fp = (C.c_void_p * len(fs))(*[f.ctypes.data for f in fs])
    up = (C.c_void_p * n)(*[u[i].ctypes.data for i in range(n)])
    vp_ = (C.c_void_p * n)(*[v[i].ctypes.data for i in range(n)])
    c = ctx if ctx is not None else default_context(0)
    check(lib.micv_lk_flow_seq_host(c.handle, fp, len(fs), rows, cols, cols * cn * a.dtype.itemsize, cn,
                                    0 if a.dtype == np.uint8 else 5, int(winSize), int(levels), up, vp_, cols * 4))
    return u, v


def calcOpticalFlowPyrBatch(prev, next_, winSize=21, levels=4, ctx=None, out=None, stream=None):
    """Batched device form: prev/next are [B, rows, cols] f32 CUDA tensors (contiguous).
    Returns (u, v) of the same shape.  One set of launches for the whole batch."""
    import torch
    if not (B.is_dev(prev) and prev.is_cuda and prev.dim() == 3 and prev.is_contiguous()
            and prev.dtype == torch.float32):
        raise ValueError("prev: need a contiguous [B, rows, cols] float32 CUDA tensor")
    if not (isinstance(next_, torch.Tensor) and next_.is_cuda and next_.device == prev.device
            and tuple(next_.shape) == tuple(prev.shape) and next_.is_contiguous() and next_.dtype == prev.dtype):
        raise ValueError("next: must be a contiguous CUDA tensor matching prev (shape, dtype, device)")
    nb, rows, cols = prev.shape
    if out is None:
        u = torch.empty_like(prev)
        v = torch.empty_like(prev)
    else:
        u, v = out
        for t, name in ((u, "out[0]"), (v, "out[1]")):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.device == prev.device and t.dtype == torch.float32
                    and tuple(t.shape) == tuple(prev.shape) and t.is_contiguous()):
                raise ValueError(f"{name}: must be a contiguous float32 CUDA tensor of prev's shape on prev's device")
        if u.data_ptr() == v.data_ptr() or u.data_ptr() in (prev.data_ptr(), next_.data_ptr()) \
                or v.data_ptr() in (prev.data_ptr(), next_.data_ptr()):
            raise ValueError("out: u, v, prev and next must be four different buffers")
    if stream is None:
        stream = torch.cuda.current_stream(prev.device).cuda_stream
    c = ctx if ctx is not None else default_context(prev.device.index or 0, stream)
    check(lib.micv_lk_flow_pyr_batch_dev(c.handle, prev.data_ptr(), next_.data_ptr(), nb,
                                         rows * cols * 4, rows, cols, cols * 4, int(winSize),
                                         int(levels), u.data_ptr(), v.data_ptr(), rows * cols * 4,
                                         cols * 4, stream))
    return u, v
