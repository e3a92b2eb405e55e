#!/usr/bin/env python3
"""The exact-sum stereo kernels (stereo_exact.hip) against the float kernels (MICV_OPT_STEREO_EXACT = -1) on random
8-bit-valued images: every radius, flag set, ragged size and disparity range; then C3 timing of both.  Ad-hoc sweep --
the committed cases are in tests/test_ps124_gpu.py."""
import itertools
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from introtocomputervision_amd import stereo, synth
from introtocomputervision_amd._capi import Context, Timer, OPT_STEREO_EXACT

fast, slow = Context(0), Context(0)
slow.set_option(OPT_STEREO_EXACT, -1)
rng = np.random.default_rng(1)
bad = 0
n = 0
cases = []
for rad in (1, 2, 3, 4, 5, 6, 7):
    for (rows, cols) in ((9, 40), (33, 141), (64, 256), (70, 333)):
        for (lo, hi) in ((-20, 0), (-70, 5), (0, 127), (-128, 127), (-3, -3)):
            for flags in (0, 1, 2, 3, 4, 8, 11):
                if flags & 4 and rad > 5:
                    continue
                if flags & 1 and rad < 2:
                    continue
                cases.append((rad, rows, cols, lo, hi, flags))
for (rad, rows, cols, lo, hi, flags) in cases:
    kind = n % 3
    if kind == 0:
        left = rng.integers(0, 256, (rows, cols)).astype(np.float32)
        right = np.roll(left, -5, axis=1)
        right[::3] = rng.integers(0, 256, right[::3].shape)
    elif kind == 1:  # flat regions: ties everywhere
        left = (rng.integers(0, 3, (rows, cols)) * 100).astype(np.float32)
        right = (rng.integers(0, 3, (rows, cols)) * 100).astype(np.float32)
    else:
        left = rng.integers(200, 256, (rows, cols)).astype(np.float32)
        right = rng.integers(0, 30, (rows, cols)).astype(np.float32)
    L, R = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    a = stereo.disparitySSD(L, R, rad, lo, hi, flags, ctx=fast).cpu().numpy()
    b = stereo.disparitySSD(L, R, rad, lo, hi, flags, ctx=slow).cpu().numpy()
    n += 1
    if not np.array_equal(a, b):
        bad += 1
        if bad <= 40:
            w = np.argwhere(a != b)
            print("MISMATCH", dict(rad=rad, rows=rows, cols=cols, lo=lo, hi=hi, flags=flags, kind=kind), len(w), "px; first", w[:3].tolist(),
                  a[tuple(w[0])], b[tuple(w[0])])
print(json.dumps({"cases": n, "mismatching": bad}))

# one non-integer pixel: the float kernel must take over
left = rng.integers(0, 256, (64, 200)).astype(np.float32)
right = np.roll(left, -7, axis=1)
left[40, 100] += 0.5
L, R = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
a = stereo.disparitySSD(L, R, 5, -30, 0, 0, ctx=fast).cpu().numpy()
b = stereo.disparitySSD(L, R, 5, -30, 0, 0, ctx=slow).cpu().numpy()
print("fallback identical:", bool(np.array_equal(a, b)))

stream = torch.cuda.current_stream().cuda_stream


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
a = stereo.disparitySSD(L, R, 5, -127, 0, ctx=fast).cpu().numpy()
b = stereo.disparitySSD(L, R, 5, -127, 0, ctx=slow).cpu().numpy()
print("C3 identical:", bool(np.array_equal(a, b)))
for rad in (5, 3, 7):
    f = min(timeit(lambda: stereo.disparitySSD(L, R, rad, -127, 0, ctx=fast)) for _ in range(3))
    s = min(timeit(lambda: stereo.disparitySSD(L, R, rad, -127, 0, ctx=slow)) for _ in range(3))
    print(json.dumps({"radius": rad, "ssd_exact_ms": round(f, 4), "ssd_float_ms": round(s, 4)}))
# what a float pair costs with the exact-sum path enabled (pre-pass + the float tiles riding in the search launch) against the
# float kernels alone
fl = (left + 0.25).astype(np.float32)
fr = (right + 0.25).astype(np.float32)
FL, FR = torch.from_numpy(fl).cuda(), torch.from_numpy(fr).cuda()
a = stereo.disparitySSD(FL, FR, 5, -127, 0, ctx=fast).cpu().numpy()
b = stereo.disparitySSD(FL, FR, 5, -127, 0, ctx=slow).cpu().numpy()
f = min(timeit(lambda: stereo.disparitySSD(FL, FR, 5, -127, 0, ctx=fast)) for _ in range(3))
s = min(timeit(lambda: stereo.disparitySSD(FL, FR, 5, -127, 0, ctx=slow)) for _ in range(3))
print(json.dumps({"float_pair_identical": bool(np.array_equal(a, b)), "with_prepass_ms": round(f, 4), "float_kernels_only_ms": round(s, 4)}))
