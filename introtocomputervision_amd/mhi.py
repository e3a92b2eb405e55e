"""`mhi::` namespace of the reference's ps7 (ProblemSets/ps7_cpp/include/MotionHistory.h:8-28):
frame differencing and motion-history images, device tensors (uint8 CUDA, single channel)."""
from ._capi import check, lib
from .lk import _ctx_for


def _chk(t, name):
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dim() == 2 and t.dtype == torch.uint8
            and t.stride(1) == 1):
        raise ValueError(f"{name}: need a 2-D uint8 CUDA tensor with unit column stride")


def _stream(t):
    import torch
    return torch.cuda.current_stream(t.device).cuda_stream


def frameDifference(f1, f2, thresh, blurSize=3, blurSigma=1.0, ctx=None):
    """mhi::frameDifference (MotionHistory.cpp:26-77) -> {0,1} uint8 mask (blurSize = the side of the
    reference's square cv::Size)."""
    import torch
    _chk(f1, "f1")
    _chk(f2, "f2")
    if tuple(f1.shape) != tuple(f2.shape) or f1.stride(0) != f2.stride(0):
        raise ValueError("f1 and f2 differ in size / stride")
    rows, cols = f1.shape
    diff = torch.empty((rows, cols), dtype=torch.uint8, device=f1.device)
    check(lib.micv_mhi_frame_difference_dev(_ctx_for(f1, ctx).handle, f1.data_ptr(), f2.data_ptr(), rows,
                                            cols, f1.stride(0), float(thresh), int(blurSize),
                                            float(blurSigma), diff.data_ptr(), cols, _stream(f1)))
    return diff


def thresholdDifference(src, thresh, ctx=None):
    """thresholdDifference (MotionHistory.cu:27-48) -> {0,1} uint8."""
    import torch
    _chk(src, "src")
    rows, cols = src.shape
    dst = torch.empty((rows, cols), dtype=torch.uint8, device=src.device)
    check(lib.micv_mhi_threshold_dev(_ctx_for(src, ctx).handle, src.data_ptr(), rows, cols, src.stride(0),
                                     float(thresh), dst.data_ptr(), cols, _stream(src)))
    return dst


def calcMotionHistory(history, binaryMask, tau, ctx=None):
    """mhi::calcMotionHistory (MotionHistory.cpp:79-96): updates `history` in place."""
    _chk(history, "history")
    _chk(binaryMask, "binaryMask")
    if tuple(history.shape) != tuple(binaryMask.shape):
        raise ValueError("history and binaryMask differ in size")
    rows, cols = history.shape
    check(lib.micv_mhi_update_dev(_ctx_for(history, ctx).handle, history.data_ptr(), history.stride(0),
                                  binaryMask.data_ptr(), binaryMask.stride(0), rows, cols, int(tau),
                                  _stream(history)))
    return history


def energyFromHistory(mhi):
    """mhi::energyFromHistory (MotionHistory.cpp:98-105): elementwise `> 0` (plain tensor op)."""
    return (mhi > 0).to(mhi.dtype)
