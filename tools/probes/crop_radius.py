#!/usr/bin/env python3
"""How far does a frame border reach into a pyramidal LK result?  A 4096^2 smooth pair on the device, the oracle on crops
with artificial borders: prints, per crop side, the depth of the deepest cell that differs from the frame's result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import _oracle as orc
from introtocomputervision_amd import lk
N, C = 4096, 1536
g = torch.Generator(device="cuda").manual_seed(0x5EED16)
low = torch.rand((1, 1, N // 8 + 2, N // 8 + 2), device="cuda", generator=g) * 255
prev = torch.nn.functional.interpolate(low, scale_factor=8, mode="bicubic", align_corners=False)[0, 0, 8:8 + N, 8:8 + N].contiguous()
nxt = (torch.roll(prev, (2, -3), (0, 1)) + 0.25).contiguous()
for levels in (1, 2, 3, 4, 5):
    u, v = lk.calcOpticalFlowPyr(prev, nxt, 15, levels)
    y = x = 1280
    eu, ev = orc.lk_flow_pyr(prev[y:y + C, x:x + C].cpu().numpy(), nxt[y:y + C, x:x + C].cpu().numpy(), 15, levels)
    bad = (u[y:y + C, x:x + C].cpu().numpy() != eu) | (v[y:y + C, x:x + C].cpu().numpy() != ev)
    # depth from each side, measured on the central band of the other axis (rows next to a side border differ anyway)
    half = C // 2
    rows, cols = np.nonzero(bad[:, half - 64:half + 64].any(1))[0], np.nonzero(bad[half - 64:half + 64, :].any(0))[0]
    top = max([r for r in rows if r < half], default=-1) + 1
    bot = C - min([r for r in rows if r >= half], default=C)
    left = max([c for c in cols if c < half], default=-1) + 1
    right = C - min([c for c in cols if c >= half], default=C)
    print({"levels": levels, "reach_top": top, "reach_bottom": bot, "reach_left": left, "reach_right": right,
           "max_abs_flow": round(float(torch.maximum(u.abs().max(), v.abs().max())), 2)}, flush=True)
