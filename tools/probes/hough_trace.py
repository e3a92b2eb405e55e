#!/usr/bin/env python3
"""ps1's Hough chain at 1080p for a kernel trace (tools/trace_script.sh): lines accumulator, circles accumulator, top-10 peaks of each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from introtocomputervision_amd import hough, _capi
ctx = _capi.Context(0)
rng = np.random.default_rng(5)
rows, cols = 1080, 1920
m = np.zeros((rows, cols), np.uint8)
for k in range(12):  # a dozen straight lines and circles worth of edge pixels (~1 % density)
    x0, y0, x1, y1 = rng.integers(0, cols), rng.integers(0, rows), rng.integers(0, cols), rng.integers(0, rows)
    t = np.linspace(0, 1, 2000)
    m[np.clip((y0 + t * (y1 - y0)).astype(int), 0, rows - 1), np.clip((x0 + t * (x1 - x0)).astype(int), 0, cols - 1)] = 255
    a = np.linspace(0, 2 * np.pi, 600)
    cx, cy = rng.integers(100, cols - 100), rng.integers(100, rows - 100)
    m[np.clip((cy + 40 * np.sin(a)).astype(int), 0, rows - 1), np.clip((cx + 40 * np.cos(a)).astype(int), 0, cols - 1)] = 255
mask = torch.from_numpy(m).cuda()
for _ in range(20):
    acc = hough.houghLinesAccumulate(mask, 1, 1, ctx=ctx)
    hough.findLocalMaxima(acc, 10, 100, ctx=ctx)
    accc = hough.houghCirclesAccumulate(mask, 40, ctx=ctx)
    hough.findLocalMaxima(accc, 10, 100, ctx=ctx)
torch.cuda.synchronize()
print("edge pixels", int((m > 0).sum()), "lines acc", tuple(acc.shape), "circles acc", tuple(accc.shape))
