#!/usr/bin/env python3
"""SSD / NCC window stereo at the C3 configuration (1080p, r = 5, 128 disparities), HIP-event timing, one JSON line.
MICV_LIB selects the build (A/B: run once per library on the same box)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from introtocomputervision_amd import stereo, synth
from introtocomputervision_amd._capi import Context, Timer

ctx = Context(0)
stream = torch.cuda.current_stream().cuda_stream
rad = int(sys.argv[1]) if len(sys.argv) > 1 else 5
if len(sys.argv) > 2:  # MICV_OPT_STEREO_EXACT: -1 = the float kernels only
    from introtocomputervision_amd._capi import OPT_STEREO_EXACT
    ctx.set_option(OPT_STEREO_EXACT, int(sys.argv[2]))


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = Timer()
    t.start(stream)
    for _ in range(iters):
        fn()
    t.stop(stream)
    return t.elapsed_ms() / iters


left, right, _ = synth.stereo_pair(0x5EED0002, 1080, 1920)
L, R = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
ssd = min(timeit(lambda: stereo.disparitySSD(L, R, rad, -127, 0, ctx=ctx)) for _ in range(3))
ncc = min(timeit(lambda: stereo.disparityNCorr(L, R, rad, -127, 0, ctx=ctx)) for _ in range(3))
print(json.dumps({"lib": os.path.basename(os.environ.get("MICV_LIB", "libmicv.so")), "radius": rad, "ssd_ms": round(ssd, 4),
                  "ncc_ms": round(ncc, 4), "ncc_over_ssd": round(ncc / ssd, 3)}))
