"""Soak of the one-launch ordered compaction: thousands of calls at random sizes / densities (default rule and
forced one-launch), results against numpy.  Run under a timeout: a lost wake-up in the look-back would hang."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from introtocomputervision_amd import harris, _capi
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ctxs = [_capi.Context(0), _capi.Context(0)]
ctxs[1].set_option(_capi.OPT_COMPACT_3PASS, -1)
t0 = time.time(); n = 0
while time.time() - t0 < float(sys.argv[2] if len(sys.argv) > 2 else 20):
    rows, cols = int(rng.integers(1, 1500)), int(rng.integers(1, 2500))
    dens = float(rng.choice([0.0, 0.001, 0.05, 0.5, 0.99, 1.0]))
    R = torch.rand((rows, cols), device="cuda")
    thr = 1.0 - dens if 0 < dens < 1 else (2.0 if dens == 0 else -1.0)
    exp = torch.nonzero(R.double() >= thr).to(torch.int32)
    for ctx in ctxs:
        for _ in range(3):
            _, locs = harris.refineCorners(R, thr, 0, ctx=ctx)
            assert torch.equal(locs, exp), (rows, cols, dens)
            n += 1
print("calls", n, "ok")
