"""Descriptor matching of the reference's ps4 driver (ProblemSets/ps4_cpp/src/Solution.cpp:172-184):
cv::BFMatcher::create()->knnMatch(d1, d2, raw, 2) followed by the 0.75 ratio test."""
from ._capi import check, lib
from .lk import _ctx_for


def knnMatch2(query, train, ctx=None):
    """2 nearest train descriptors (L2) per query row -> (idx [nq, 2] int32, dist [nq, 2] float32)."""
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
