#!/usr/bin/env python3
"""Soak of the carried pyramid build (MICV_OPT_LK_BUILD_OVERLAP = 1) against the build launch (-1): random frame sizes
up to 1100 x 2000 (16-byte rows), batches 1-4, 3-6 levels; bytes must agree.
  python tools/probes/carry_soak.py [draws]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from introtocomputervision_amd import lk, _capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(0xC0FFEE)
g = torch.Generator(device="cuda").manual_seed(7)
ctx = _capi.Context(0)
bad = 0
for i in range(n):
    rows = int(rng.integers(64, 1100)); cols = 4 * int(rng.integers(16, 500)); batch = int(rng.integers(1, 5))
    levels = int(rng.integers(3, 7))
    while (min(rows, cols) >> (levels - 1)) < 4: levels -= 1
    if levels < 3: continue
    pad = 0  # (the batch entry point takes dense frames)
    prev = torch.rand((batch, rows, cols), device="cuda", generator=g) * 255
    nxt = torch.roll(prev, (1, -2), (1, 2)) + 0.25
    outs = []
    for opt in (1, -1):
        ctx.set_option(_capi.OPT_LK_BUILD_OVERLAP, opt)
        u, v = lk.calcOpticalFlowPyrBatch(prev, nxt, 15, levels, ctx=ctx)
        outs.append((u.cpu().numpy().tobytes(), v.cpu().numpy().tobytes()))
    if outs[0] != outs[1]:
        bad += 1
        print("MISMATCH", rows, cols, batch, levels, pad, flush=True)
print({"draws": n, "mismatches": bad})
sys.exit(1 if bad else 0)
