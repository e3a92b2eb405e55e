#!/usr/bin/env python3
"""mhi::frameDifference at 1080p per blur size (3x3 = the reference's default argument, 5x5, 31x31 = config/ps7.yaml):
ms per call by HIP events."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from introtocomputervision_amd import mhi, _capi
ctx = _capi.Context(0)
rng = np.random.default_rng(1)
f1 = torch.from_numpy(rng.integers(0, 256, (1080, 1920)).astype(np.uint8)).cuda()
f2 = torch.from_numpy(rng.integers(0, 256, (1080, 1920)).astype(np.uint8)).cuda()
for ks, sg, th in ((3, 1.0, 20), (5, 1.5, 20), (31, 10.0, 1.7)):
    for _ in range(5): mhi.frameDifference(f1, f2, th, ks, sg, ctx=ctx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): mhi.frameDifference(f1, f2, th, ks, sg, ctx=ctx)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"blur": ks, "ms": round(e0.elapsed_time(e1) / 50, 4)}))
