#!/usr/bin/env python3
"""The Harris chain at 4K and at C1's size, three separate calls and the one-call form (micv_harris_corners_dev), for a
kernel trace (tools/trace_script.sh) -- device times of sobel / response / image -> R / NMS / list kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from introtocomputervision_amd import harris, synth, _capi
ctx = _capi.Context(0)
for rows, cols in ((2160, 3840), (480, 640)):
    img = torch.from_numpy(synth.checkerboard(rows, cols, 40, seed=0x5EED0004)).cuda()
    for _ in range(30):
        gx, gy = harris.getGradients(img, 3, ctx=ctx)
        R = harris.getCornerResponse(gx, gy, 5, 1.5, 0.04, ctx=ctx)
        harris.refineCorners(R, 5e8, 5, capacity=1 << 16, ctx=ctx, lazy=True)
        harris.cornersFromImage(img, 3, 5, 1.5, 0.04, 5e8, 5, capacity=1 << 16, ctx=ctx, want_gradients=False, lazy=True)
        harris.cornersFromImage(img, 3, 5, 1.5, 0.04, 5e8, 5, capacity=1 << 16, ctx=ctx, want_gradients=True, lazy=True)
torch.cuda.synchronize()
