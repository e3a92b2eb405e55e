#!/usr/bin/env python3
"""BASELINE C5's chain (bench.py `secondary.C5_4k_chain`) under a kernel trace: run as
    KEEP_TRACE=1 bash tools/trace_script.sh tools/probes/c5_timeline.py            (collect)
    python3 tools/probes/c5_timeline.py --timeline gpurun_out/trace_c5_timeline      (print one iteration's timeline)
The timeline lists every kernel of the LAST iteration with its start offset, duration and the idle gap before it."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--timeline":
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    marker = sys.argv[3] if len(sys.argv) > 3 else "harris_image_response"   # the chain's first kernel
    firsts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    a, b = firsts[-2], firsts[-1]          # the last complete iteration
    t0, prev_end = int(rows[a]["Start_Timestamp"]), None
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        print(f'{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {gap:7.1f}  {r["Kernel_Name"].split("(")[0][-70:]}')
        prev_end = max(e, prev_end or e)
    print(f"iteration: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us from first kernel to the next iteration's first")
    sys.exit(0)

import numpy as np
import torch

from introtocomputervision_amd import harris, lk, synth

ctx = lk.default_context()
WIN, LEVELS = 15, 5
chk = synth.checkerboard(2160, 3840, square=40)
tex = synth.smooth_noise(0x5EED0004, 2160, 3840)
p = np.round(tex * (chk / 192.0)).astype(np.float32)
P, N = torch.from_numpy(p).cuda(), torch.from_numpy(np.ascontiguousarray(np.roll(p, shift=(-2, 3), axis=(0, 1)))).cuda()
count_host = torch.empty(1, dtype=torch.int64).pin_memory()
overlap = os.environ.get("C5_OVERLAP", "1") == "1"


def c5():
    if overlap:
        h = harris.cornersFromImage(P, 3, 5, 1.5, 0.04, 5e8, 5, capacity=1 << 20, ctx=ctx, lazy=True)
        count_host.copy_(h["count"], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        u_, v_ = lk.calcOpticalFlowPyr(P, N, WIN, LEVELS, ctx=ctx)
        ev.synchronize()
        locs = h["locs"][:int(count_host[0])]
    else:
        h = harris.cornersFromImage(P, 3, 5, 1.5, 0.04, 5e8, 5, capacity=1 << 20, ctx=ctx)
        locs = h["locs"]
    kp = harris.getKeypoints(h["gx"], h["gy"], locs, 10, ctx=ctx)
    desc = harris.computeDescriptors(h["gx"], h["gy"], kp, ctx=ctx)
    if not overlap:
        u_, v_ = lk.calcOpticalFlowPyr(P, N, WIN, LEVELS, ctx=ctx)
    ll = locs.long()
    yy, xx = ll[:, 0], ll[:, 1]
    return locs, kp, desc, u_[yy, xx], v_[yy, xx]


import time
for _ in range(3):
    c5()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    c5()
torch.cuda.synchronize()
print({"overlap": overlap, "ms": round((time.perf_counter() - t0) * 100, 4), "corners": int(len(c5()[0]))})
