#!/usr/bin/env python3
"""r06: 5-level LK on 8 x 1080p pairs at windows 15 and 21, for tools/pmc_script.sh / trace_script.sh: the level-0 launch of
each window side by side (why window 21 costs 1.6 x window 15 for 1.4 x the taps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from introtocomputervision_amd import lk, synth, _capi
ctx = _capi.Context(0)
B = 8
prev = np.empty((B, 1080, 1920), np.float32); nxt = np.empty_like(prev)
for i in range(B):
    prev[i], nxt[i] = synth.lk_pair(0x5EED0005 + i, 1080, 1920, 3, -2)
P, N = torch.from_numpy(prev).cuda(), torch.from_numpy(nxt).cuda()
for win in (15, 21):
    for _ in range(6):
        lk.calcOpticalFlowPyrBatch(P, N, win, 5, ctx=ctx)
torch.cuda.synchronize()
