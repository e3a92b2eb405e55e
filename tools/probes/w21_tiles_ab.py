import sys, time, json, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from introtocomputervision_amd import lk, synth, _capi
for B in (4, 1, 8):
    p, n = synth.lk_pair(0x5EED0005, 1080, 1920, 3, -2)
    dp, dn = torch.from_numpy(np.stack([p]*B)).cuda(), torch.from_numpy(np.stack([n]*B)).cuda()
    out = (torch.empty_like(dp), torch.empty_like(dp))
    ref = None
    for tall in (0, 1, 0, 1):
        ctx = _capi.Context(0)
        ctx.set_option(_capi.OPT_LK_TALL_TILES, tall)
        for _ in range(3): lk.calcOpticalFlowPyrBatch(dp, dn, 21, 5, ctx=ctx, out=out)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(30): lk.calcOpticalFlowPyrBatch(dp, dn, 21, 5, ctx=ctx, out=out)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 30 * 1e3
        ctx.profile(True); ctx.profile_reset()
        for _ in range(10): lk.calcOpticalFlowPyrBatch(dp, dn, 21, 5, ctx=ctx, out=out)
        torch.cuda.synchronize()
        lv = [ctx.profile_lk_level(l) for l in range(5)]
        if ref is None: ref = (out[0].clone(), out[1].clone())
        same = bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
        print(json.dumps({"pairs": B, "tall_tiles_opt": tall, "form": "64x64/1024" if tall == 0 else "64x32/1024", "ms": round(ms, 4),
                          "Gpix_s": round(B * 1080 * 1920 / ms / 1e6, 1), "level_ms": [round(a / max(k, 1), 4) for a, k in lv], "same_bits": same}), flush=True)
