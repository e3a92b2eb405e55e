#!/usr/bin/env python3
"""The standalone pyramid / warp entry points (a4-a6) at 1080p, for a kernel trace: the calls are shorter than the
Python call overhead, so their device time comes from `rocprofv3 --kernel-trace --stats -- python3 tools/probes/a456_trace.py`
(tools/profile_kernels.sh style), not from host-side timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from introtocomputervision_amd import lk, pyr, synth
from introtocomputervision_amd._capi import Context
ctx = Context(0)
prev, _ = synth.lk_pair(0x5EED0005, 1080, 1920)
P = torch.from_numpy(prev).cuda()
du = torch.full_like(P, 2.3); dv = torch.full_like(P, -1.7)
rng = torch.Generator(device="cuda"); rng.manual_seed(1)
du2 = (torch.rand(P.shape, device="cuda", generator=rng) - 0.5) * 12; dv2 = (torch.rand(P.shape, device="cuda", generator=rng) - 0.5) * 12
for _ in range(30):
    pyr.pyrDown(P, ctx=ctx)
    pyr.pyrUp(P, ctx=ctx)
    lk.warp(P, du, dv, ctx=ctx)
    lk.warp(P, du2, dv2, ctx=ctx)
    pyr.makeGaussianPyramid(P, 5, ctx=ctx)
torch.cuda.synchronize()
