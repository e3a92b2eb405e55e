#!/usr/bin/env python3
"""In-kernel phase shares of the fused level-0 LK kernel (diagnostic; see micv_profile_lk_phases).
Needs the diagnostic flavour of the library:
  MICV_OUT=libmicv_diag.so EXTRA_HIPCC_FLAGS=-DMICV_DIAG bash introtocomputervision_amd/csrc/build.sh
  MICV_LIB=$PWD/introtocomputervision_amd/libmicv_diag.so python tools/phase_profile.py [pairs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from introtocomputervision_amd import lk, synth
from introtocomputervision_amd._capi import Context

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p, n = synth.lk_pair(0x5EED0005, 1080, 1920, 3, -2)
prev = torch.from_numpy(np.stack([p] * B)).cuda()
nxt = torch.from_numpy(np.stack([n] * B)).cuda()
ctx = Context(0)
for _ in range(3):
    lk.calcOpticalFlowPyrBatch(prev, nxt, 15, 5, ctx=ctx)
torch.cuda.synchronize()
ctx.profile_lk_phases(True)
for _ in range(5):
    lk.calcOpticalFlowPyrBatch(prev, nxt, 15, 5, ctx=ctx)
t = ctx.profile_lk_phases(False)
names = ["stage", "pyrUp rows", "warp", "gradients", "window sums", "solve+store"]
for label, off in (("interior", 0), ("border", 8)):
    tot = sum(t[off:off + 6]) or 1
    print(label, "tiles: total ticks", tot)
    for i, nm in enumerate(names):
        print(f"   {nm:12s} {t[off + i]:14d}  {100.0 * t[off + i] / tot:5.1f} %")
