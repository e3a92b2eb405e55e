"""`mhi::` namespace of the reference's ps7 (ProblemSets/ps7_cpp/include/MotionHistory.h:8-28):
frame differencing and motion-history images, device tensors (uint8 CUDA, single channel)."""
from ._capi import check, lib
from .lk import _ctx_for
from .match import _host_ctx


def _chk(t, name):
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dim() == 2 and t.dtype == torch.uint8
            and t.stride(1) == 1):
        raise ValueError(f"{name}: need a 2-D uint8 CUDA tensor with unit column stride")


def _stream(t):
    import torch
    return torch.cuda.current_stream(t.device).cuda_stream


def _blur_wh(blurSize):
    """cv::Size(width, height) as (w, h); a bare int means a square."""
    if isinstance(blurSize, (tuple, list)):
        w, h = blurSize
        return int(w), int(h)
    return int(blurSize), int(blurSize)


def frameDifference(f1, f2, thresh, blurSize=(3, 3), blurSigma=1.0, ctx=None):
    """mhi::frameDifference (MotionHistory.cpp:26-77) -> {0,1} uint8 mask.  blurSize is the
    reference's cv::Size as (width, height) (MotionHistory.h:14; default cv::Size(3, 3))."""
    import numpy as np
    bw, bh = _blur_wh(blurSize)
    if isinstance(f1, np.ndarray):  # host-pointer entry point
        a1 = np.ascontiguousarray(f1, np.uint8)
        a2 = np.ascontiguousarray(f2, np.uint8)
        if a1.ndim != 2 or a1.shape != a2.shape:
            raise ValueError("f1 / f2: need 2-D uint8 arrays of equal shape")
        out = np.empty_like(a1)
        check(lib.micv_mhi_frame_difference_host((ctx or _host_ctx()).handle, a1.ctypes.data, a2.ctypes.data,
                                                 a1.shape[0], a1.shape[1], a1.strides[0], float(thresh),
                                                 bw, bh, float(blurSigma), out.ctypes.data,
                                                 out.strides[0]))
        return out
    import torch
    _chk(f1, "f1")
    _chk(f2, "f2")
    if tuple(f1.shape) != tuple(f2.shape) or f1.stride(0) != f2.stride(0):
        raise ValueError("f1 and f2 differ in size / stride")
    rows, cols = f1.shape
    diff = torch.empty((rows, cols), dtype=torch.uint8, device=f1.device)
    check(lib.micv_mhi_frame_difference_dev(_ctx_for(f1, ctx).handle, f1.data_ptr(), f2.data_ptr(), rows,
                                            cols, f1.stride(0), float(thresh), bw, bh,
                                            float(blurSigma), diff.data_ptr(), cols, _stream(f1)))
    return diff


def thresholdDifference(src, thresh, ctx=None):
    """thresholdDifference (MotionHistory.cu:27-48) -> {0,1} uint8."""
    import numpy as np
    if isinstance(src, np.ndarray):
        a1 = np.ascontiguousarray(src, np.uint8)
        out = np.empty_like(a1)
        check(lib.micv_mhi_threshold_host((ctx or _host_ctx()).handle, a1.ctypes.data, a1.shape[0], a1.shape[1],
                                          a1.strides[0], float(thresh), out.ctypes.data, out.strides[0]))
        return out
    import torch
    _chk(src, "src")
    rows, cols = src.shape
    dst = torch.empty((rows, cols), dtype=torch.uint8, device=src.device)
    check(lib.micv_mhi_threshold_dev(_ctx_for(src, ctx).handle, src.data_ptr(), rows, cols, src.stride(0),
                                     float(thresh), dst.data_ptr(), cols, _stream(src)))
    return dst


def calcMotionHistory(history, binaryMask, tau, ctx=None):
    """mhi::calcMotionHistory (MotionHistory.cpp:79-96): updates `history` in place."""
    import numpy as np
    if isinstance(history, np.ndarray):
        if history.dtype != np.uint8 or history.ndim != 2 or not history.flags.c_contiguous:
            raise ValueError("history: need a contiguous 2-D uint8 array (updated in place)")
        m = np.ascontiguousarray(binaryMask, np.uint8)
        if m.shape != history.shape:
            raise ValueError("history and binaryMask differ in size")
        check(lib.micv_mhi_update_host((ctx or _host_ctx()).handle, history.ctypes.data, history.strides[0],
                                       m.ctypes.data, m.strides[0], history.shape[0], history.shape[1], int(tau)))
        return history
    _chk(history, "history")
    _chk(binaryMask, "binaryMask")
    if tuple(history.shape) != tuple(binaryMask.shape):
        raise ValueError("history and binaryMask differ in size")
    rows, cols = history.shape
    check(lib.micv_mhi_update_dev(_ctx_for(history, ctx).handle, history.data_ptr(), history.stride(0),
                                  binaryMask.data_ptr(), binaryMask.stride(0), rows, cols, int(tau),
                                  _stream(history)))
    return history


def energyFromHistory(mhi, ctx=None):
    """mhi::energyFromHistory (MotionHistory.cpp:98-112): any nonzero history value -> 1.  A list of
    histories gives a list of energies (the vector overload, :107-112)."""
    import numpy as np
    if isinstance(mhi, (list, tuple)):
        return [energyFromHistory(m, ctx) for m in mhi]
    if isinstance(mhi, np.ndarray):
        a = np.ascontiguousarray(mhi, np.uint8)
        if a.ndim != 2:
            raise ValueError("mhi: need a 2-D uint8 array")
        out = np.empty_like(a)
        check(lib.micv_mhi_energy_host((ctx or _host_ctx()).handle, a.ctypes.data, a.shape[0], a.shape[1],
                                       a.strides[0], out.ctypes.data, out.strides[0]))
        return out
    import torch
    _chk(mhi, "mhi")
    rows, cols = mhi.shape
    mei = torch.empty((rows, cols), dtype=torch.uint8, device=mhi.device)
    check(lib.micv_mhi_energy_dev(_ctx_for(mhi, ctx).handle, mhi.data_ptr(), rows, cols, mhi.stride(0),
                                  mei.data_ptr(), cols, _stream(mhi)))
    return mei
