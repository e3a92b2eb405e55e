"""Descriptor matching of the reference's ps4 driver (ProblemSets/ps4_cpp/src/Solution.cpp:172-184):
cv::BFMatcher::create()->knnMatch(d1, d2, raw, 2) followed by the 0.75 ratio test."""
from ._capi import check, lib
from .lk import _ctx_for

_HOST_CTX = []


def _host_ctx():
    if not _HOST_CTX:
        from ._capi import Context
        _HOST_CTX.append(Context(0))
    return _HOST_CTX[0]


def knnMatch2(query, train, ctx=None):
    """2 nearest train descriptors (L2) per query row -> (idx [nq, 2] int32, dist [nq, 2] float32).
    numpy arrays take the host-pointer entry point (upload, kernel, download), CUDA tensors the
    device one."""
    import numpy as np
    if isinstance(query, np.ndarray):
        from ._capi import Context
        q = np.ascontiguousarray(query, np.float32)
        t = np.ascontiguousarray(train, np.float32)
        if q.ndim != 2 or t.ndim != 2 or q.shape[1] != t.shape[1]:
            raise ValueError("query / train: need 2-D float32 arrays of equal width")
        idx = np.empty((q.shape[0], 2), np.int32)
        dist = np.empty((q.shape[0], 2), np.float32)
        c = ctx or _host_ctx()
        check(lib.micv_bf_knn2_host(c.handle, q.ctypes.data, q.shape[0], q.strides[0], t.ctypes.data,
                                    t.shape[0], t.strides[0], q.shape[1], idx.ctypes.data, dist.ctypes.data))
        return idx, dist
    import torch
    for t, n in ((query, "query"), (train, "train")):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dim() == 2 and t.dtype == torch.float32
                and t.stride(1) == 1):
            raise ValueError(f"{n}: need a 2-D float32 CUDA tensor with unit column stride")
    if query.shape[1] != train.shape[1]:
        raise ValueError("descriptor dimensions differ")
    nq, dim = query.shape
    idx = torch.empty((nq, 2), dtype=torch.int32, device=query.device)
    dist = torch.empty((nq, 2), dtype=torch.float32, device=query.device)
    check(lib.micv_bf_knn2_dev(_ctx_for(query, ctx).handle, query.data_ptr(), nq, query.stride(0) * 4,
                               train.data_ptr(), train.shape[0], train.stride(0) * 4, dim,
                               idx.data_ptr(), dist.data_ptr(),
                               torch.cuda.current_stream(query.device).cuda_stream))
    return idx, dist


def ratioTest(idx, dist, ratio=0.75, ctx=None):
    """Good matches: (matches [n, 2] int32 = (queryIdx, trainIdx), distances [n]) in query order."""
    import numpy as np
    if isinstance(idx, np.ndarray):
        import ctypes as C
        idx = np.ascontiguousarray(idx, np.int32)
        dist = np.ascontiguousarray(dist, np.float32)
        nq = idx.shape[0]
        matches = np.empty((nq, 2), np.int32)
        distances = np.empty((nq,), np.float32)
        cnt = C.c_int64(0)
        c = ctx or _host_ctx()
        check(lib.micv_bf_ratio_filter_host(c.handle, idx.ctypes.data, dist.ctypes.data, nq, float(ratio),
                                            matches.ctypes.data, distances.ctypes.data, nq, C.byref(cnt)))
        return matches[:cnt.value], distances[:cnt.value]
    import torch
    nq = idx.shape[0]
    matches = torch.empty((nq, 2), dtype=torch.int32, device=idx.device)
    distances = torch.empty((nq,), dtype=torch.float32, device=idx.device)
    cnt = torch.zeros((1,), dtype=torch.int64, device=idx.device)
    check(lib.micv_bf_ratio_filter_dev(_ctx_for(idx, ctx).handle, idx.data_ptr(), dist.data_ptr(), nq,
                                       float(ratio), matches.data_ptr(), distances.data_ptr(), nq,
                                       cnt.data_ptr(), torch.cuda.current_stream(idx.device).cuda_stream))
    n = int(cnt.item())
    return matches[:n], distances[:n]
