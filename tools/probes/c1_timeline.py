#!/usr/bin/env python3
"""BASELINE C1's chain (480x640 Harris: image -> R -> ordered corner list, count read back) for a kernel trace:
    KEEP_TRACE=1 bash tools/trace_script.sh tools/probes/c1_timeline.py
    python3 tools/probes/c5_timeline.py --timeline gpurun_out/trace_c1_timeline"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from introtocomputervision_amd import harris, lk, synth
ctx = lk.default_context()
img = torch.from_numpy(synth.checkerboard(480, 640, square=40, seed=0x5EED0001)).cuda()
def c1():
    return harris.cornersFromImage(img, 3, 5, 1.5, 0.04, 5e8, 5, ctx=ctx, want_gradients=False)["locs"]
for _ in range(5):
    c1()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    c1()
torch.cuda.synchronize()
print({"ms": round((time.perf_counter() - t0) * 20, 4), "corners": int(len(c1()))})
