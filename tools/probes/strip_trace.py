#!/usr/bin/env python3
"""8 x 1080p, 5 levels, window 15 with MICV_OPT_LK_STRIP = argv[1] (blocks per segment; 0 = tile launch) for a kernel
trace / PMC pass.  argv[2] = pairs, argv[3] = steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from introtocomputervision_amd import lk, synth, _capi
strip = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
pn = [synth.lk_pair(0x5EED0005 + i, 1080, 1920, 3, -2) for i in range(B)]
dp = torch.from_numpy(np.stack([p for p, _ in pn])).cuda()
dn = torch.from_numpy(np.stack([n for _, n in pn])).cuda()
out = (torch.empty_like(dp), torch.empty_like(dp))
ctx = _capi.Context(0)
ctx.set_lk_groups(1)
ctx.set_option(_capi.OPT_LK_STRIP, strip)
for _ in range(steps):
    lk.calcOpticalFlowPyrBatch(dp, dn, 15, 5, ctx=ctx, out=out)
torch.cuda.synchronize()
print("median u", float(out[0][0, 64:-64, 64:-64].median()))
