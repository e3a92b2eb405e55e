"""Streamed level launch (MICV_OPT_LK_STREAM) against the plain launch, same library, same box:
8 x 1080p pairs, 5 levels, window 15; prints ms per step and the per-level launch times."""
import json
import sys
import time

sys.path.insert(0, '.')
import numpy as np
import torch
from introtocomputervision_amd import lk, synth, _capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prev = np.stack([synth.lk_pair(0x5EED0005 + i, 1080, 1920, 3, -2)[0] for i in range(B)])
nxt = np.stack([synth.lk_pair(0x5EED0005 + i, 1080, 1920, 3, -2)[1] for i in range(B)])
dp, dn = torch.from_numpy(prev).cuda(), torch.from_numpy(nxt).cuda()
out = (torch.empty_like(dp), torch.empty_like(dp))
for rnd in range(3):
    for opt in (0, 1):
        ctx = _capi.Context(0)
        ctx.set_option(_capi.OPT_LK_STREAM, opt)
        for _ in range(20):
            lk.calcOpticalFlowPyrBatch(dp, dn, 15, 5, ctx=ctx, out=out)
        torch.cuda.synchronize()
        t = time.perf_counter()
        N = 200
        for _ in range(N):
            lk.calcOpticalFlowPyrBatch(dp, dn, 15, 5, ctx=ctx, out=out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / N * 1e3
        ctx.profile(True)
        ctx.profile_reset()
        for _ in range(20):
            lk.calcOpticalFlowPyrBatch(dp, dn, 15, 5, ctx=ctx, out=out)
        torch.cuda.synchronize()
        lv = [ctx.profile_lk_level(l) for l in range(5)]
        print(json.dumps({"stream": opt, "ms_per_step": round(ms, 4),
                          "level_ms": [round(a / max(n, 1), 4) for a, n in lv]}), flush=True)
