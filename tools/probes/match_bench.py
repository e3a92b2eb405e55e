#!/usr/bin/env python3
"""BFMatcher knn2 alone (N1): ms per call at a few set sizes, 128-dimensional descriptors.  MICV_LIB selects the build."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from introtocomputervision_amd import match, lk
from introtocomputervision_amd._capi import Timer
ctx = lk.default_context()
stream = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(3)
for nq, nt in ((2000, 2000), (5035, 5035), (8192, 8192), (500, 20000)):
    q = torch.rand((nq, 128), device="cuda", generator=g) * 255
    tr = torch.rand((nt, 128), device="cuda", generator=g) * 255
    for _ in range(3):
        match.knnMatch2(q, tr, ctx=ctx)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = Timer(); t.start(stream)
        for _ in range(10):
            match.knnMatch2(q, tr, ctx=ctx)
        t.stop(stream)
        best = min(best, t.elapsed_ms() / 10)
    print(json.dumps({"lib": os.path.basename(os.environ.get("MICV_LIB", "libmicv.so")), "nq": nq, "nt": nt, "ms": round(best, 4),
                      "Tflop_s_3_per_element": round(nq * nt * 128 * 3 / best / 1e9, 1)}), flush=True)
