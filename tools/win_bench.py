import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
from introtocomputervision_amd import lk, synth, _capi
p, n = synth.lk_pair(0x5EED0005, 1080, 1920, 3, -2)
dp, dn = torch.from_numpy(np.stack([p]*4)).cuda(), torch.from_numpy(np.stack([n]*4)).cuda()
out = (torch.empty_like(dp), torch.empty_like(dp))
ctx = _capi.Context(0)
import sys as _s
if len(_s.argv) > 1: ctx.set_option(_capi.OPT_LK_SHORT_TILES, int(_s.argv[1]))
for win in (7, 11, 15, 21, 43):
    for _ in range(3): lk.calcOpticalFlowPyrBatch(dp, dn, win, 5, ctx=ctx, out=out)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): lk.calcOpticalFlowPyrBatch(dp, dn, win, 5, ctx=ctx, out=out)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 20 * 1e3
    print(json.dumps({"win": win, "ms_per_4_pairs": round(ms, 3), "Gpix_s": round(4 * 1080 * 1920 / ms / 1e6, 1)}))
