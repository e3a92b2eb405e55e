#!/usr/bin/env python3
"""What a border tile costs: the level-0 launch (64x32 tiles, coarse-flow mode) of a 2-level call for frame sets with
about the same number of tiles and different border fractions.  ns per tile-slot vs the fraction gives the ratio."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from introtocomputervision_amd import lk, _capi
g = torch.Generator(device="cuda").manual_seed(5)
for rows, cols, B in ((1080, 1920, 8), (1088, 1920, 8), (2176, 3840, 2), (4352, 7680, 1), (544, 960, 32), (288, 512, 96), (1080, 1920, 16)):
    prev = torch.rand((B, rows, cols), device="cuda", generator=g) * 255
    nxt = torch.roll(prev, (1, 2), (1, 2)) + 0.5
    out = (torch.empty_like(prev), torch.empty_like(prev))
    ctx = _capi.Context(0); ctx.set_lk_groups(1); ctx.set_option(_capi.OPT_LK_CHAIN, 1)
    for _ in range(5): lk.calcOpticalFlowPyrBatch(prev, nxt, 15, 2, ctx=ctx, out=out)
    torch.cuda.synchronize()
    ctx.profile(True); ctx.profile_reset()
    for _ in range(30): lk.calcOpticalFlowPyrBatch(prev, nxt, 15, 2, ctx=ctx, out=out)
    torch.cuda.synchronize()
    a, n = ctx.profile_lk_level(0)
    tx, ty = -(-cols // 64), -(-rows // 32)
    tiles = tx * ty * B
    border = (2 * tx + 2 * ty - 4) * B
    print(json.dumps({"rows": rows, "cols": cols, "pairs": B, "tiles": tiles, "border_frac": round(border / tiles, 4),
                      "level0_ms": round(a / n, 4), "ns_per_tile": round(a / n * 1e6 / tiles, 2), "rounds": round(tiles / 512, 2)}), flush=True)
    del prev, nxt, out, ctx
