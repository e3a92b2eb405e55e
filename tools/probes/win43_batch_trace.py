#!/usr/bin/env python3
"""Window 43 (config/ps5.yaml:11), 8 x 1080p pairs, 5 levels, the batch entry point: for a kernel trace (trace_script.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from introtocomputervision_amd import lk, synth, _capi
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
WIN = int(sys.argv[2]) if len(sys.argv) > 2 else 43
GEN = int(os.environ.get("FORCE_GENERIC", "0"))  # MICV_OPT_LK_FORCE_GENERIC for windows that have a fused kernel
ps = [synth.lk_pair(0x5EED0005 + i, 1080, 1920, 3, -2) for i in range(B)]
prev = torch.from_numpy(np.stack([p for p, _ in ps])).cuda()
nxt = torch.from_numpy(np.stack([n for _, n in ps])).cuda()
ctx = _capi.Context(0)
if GEN:
    ctx.set_option(_capi.OPT_LK_FORCE_GENERIC, GEN)
out = (torch.empty_like(prev), torch.empty_like(prev))
for _ in range(3):
    lk.calcOpticalFlowPyrBatch(prev, nxt, WIN, 5, ctx=ctx, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    lk.calcOpticalFlowPyrBatch(prev, nxt, WIN, 5, ctx=ctx, out=out)
torch.cuda.synchronize()
print({"pairs": B, "win": WIN, "force_generic": GEN, "ms_per_call": round((time.perf_counter() - t0) * 100, 4)})
