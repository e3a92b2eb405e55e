#!/usr/bin/env python3
"""One line of JSON: 8 x 1080p, 5 levels, win 15 (or argv[1]) on the library MICV_LIB points at -- step time
(one pass at a time, one stream group) and the per-level launch times by HIP events.  tools/ab.py runs
it against several builds, interleaved, on one box."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from introtocomputervision_amd import lk, synth, _capi
win = int(sys.argv[1]) if len(sys.argv) > 1 else 15
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
opts = [a.split("=") for a in sys.argv[3:]]
pn = [synth.lk_pair(0x5EED0005 + i, 1080, 1920, 3, -2) for i in range(B)]
dp = torch.from_numpy(np.stack([p for p, _ in pn])).cuda()
dn = torch.from_numpy(np.stack([n for _, n in pn])).cuda()
out = (torch.empty_like(dp), torch.empty_like(dp))
ctx = _capi.Context(0)
ctx.set_lk_groups(1)
for k, v in opts:
    ctx.set_option(getattr(_capi, k), int(v))
t_end = time.perf_counter() + 0.3   # clock pre-roll
while time.perf_counter() < t_end:
    for _ in range(8): lk.calcOpticalFlowPyrBatch(dp, dn, win, 5, ctx=ctx, out=out)
    torch.cuda.synchronize()
N = 100
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(N): lk.calcOpticalFlowPyrBatch(dp, dn, win, 5, ctx=ctx, out=out)
torch.cuda.synchronize(); ms = (time.perf_counter() - t) / N * 1e3
ctx.profile(True); ctx.profile_reset()
for _ in range(40): lk.calcOpticalFlowPyrBatch(dp, dn, win, 5, ctx=ctx, out=out)
torch.cuda.synchronize()
lv = [ctx.profile_lk_level(l) for l in range(5)]
chk = float(out[0][0, 64:-64, 64:-64].median())
print(json.dumps({"lib": os.path.basename(os.environ.get("MICV_LIB", "libmicv.so")), "win": win, "pairs": B,
                  "ms_per_step": round(ms, 4), "level_ms": [round(a / max(n, 1), 4) for a, n in lv], "median_u": round(chk, 3)}))
