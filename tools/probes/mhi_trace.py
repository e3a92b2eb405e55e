#!/usr/bin/env python3
"""mhi::frameDifference at 1080p (blur 5 and the reference's default 3x3) for a kernel trace (tools/trace_script.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from introtocomputervision_amd import mhi, _capi
ctx = _capi.Context(0)
rng = np.random.default_rng(1)
f1 = torch.from_numpy(rng.integers(0, 256, (1080, 1920)).astype(np.uint8)).cuda()
f2 = torch.from_numpy(rng.integers(0, 256, (1080, 1920)).astype(np.uint8)).cuda()
for _ in range(50):
    mhi.frameDifference(f1, f2, 20, 5, 1.5, ctx=ctx)
    mhi.frameDifference(f1, f2, 20, 3, 1.0, ctx=ctx)
    mhi.frameDifference(f1, f2, 1.7, 31, 10.0, ctx=ctx)
torch.cuda.synchronize()
