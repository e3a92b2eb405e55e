"""Single-level window-43 LK through the generic kernels: two launches vs four, by image size (which form wins
where decides kTwoLaunchMinPixels in csrc/lk.hip)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from introtocomputervision_amd import lk, synth, _capi
for rows, cols in ((68, 120), (135, 240), (270, 480), (540, 960), (1080, 1920)):
    p, n = synth.lk_pair(5, rows, cols, 1, -1)
    dp, dn = torch.from_numpy(p).cuda(), torch.from_numpy(n).cuda()
    res = {}
    for name, opt in (("two", 3), ("four", 2)):
        ctx = _capi.Context(0)
        ctx.set_option(_capi.OPT_LK_FORCE_GENERIC, opt)
        for win in (43, 9):
            for _ in range(5): lk.calcOpticalFlow(dp, dn, win, ctx=ctx)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(200): lk.calcOpticalFlow(dp, dn, win, ctx=ctx)
            torch.cuda.synchronize(); res["%s_win%d_us" % (name, win)] = round((time.perf_counter() - t) / 200 * 1e6, 1)
    print(json.dumps({"rows": rows, "cols": cols, **res}), flush=True)
