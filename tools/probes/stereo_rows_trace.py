#!/usr/bin/env python3
"""r06: the exact-sum stereo search at 1920 columns and 1080 / 720 / 540 / 360 / 180 rows (all one round of waves or less),
for a kernel trace: does the launch's time follow the number of resident waves per SIMD (VALU-bound) or stay (latency)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from introtocomputervision_amd import stereo, synth, _capi
ctx = _capi.Context(0)
for rows in (1080, 720, 540, 360, 180):
    left, right, _ = synth.stereo_pair(0x5EED0002, rows, 1920)
    L, R = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    for _ in range(20):
        stereo.disparitySSD(L, R, 5, -127, 0, ctx=ctx)
torch.cuda.synchronize()
