#!/usr/bin/env python3
"""One 1080p pair per call (BASELINE configs[1] read literally): the call enqueued from the host every time against the
same call captured once in a HIP graph and replayed -- does the C ABI capture, and what do the launch gaps cost?
Also 8 pairs.  One JSON line per batch."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from introtocomputervision_amd import lk, synth, _capi

def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

for B in (1, 8):
    pn = [synth.lk_pair(0x5EED0005 + i, 1080, 1920, 3, -2) for i in range(B)]
    dp = torch.from_numpy(np.stack([p for p, _ in pn])).cuda(); dn = torch.from_numpy(np.stack([n for _, n in pn])).cuda()
    out = (torch.empty_like(dp), torch.empty_like(dp)); gout = (torch.empty_like(dp), torch.empty_like(dp))
    ctx = _capi.Context(0)
    side = torch.cuda.Stream()
    call = lambda o, s: lk.calcOpticalFlowPyrBatch(dp, dn, 15, 5, ctx=ctx, out=o, stream=s)
    for _ in range(3): call(out, side.cuda_stream)   # arena grown, attributes set: nothing illegal left for the capture
    side.synchronize()
    plain = timeit(lambda: call(out, side.cuda_stream))
    res = {"pairs": B, "enqueued_ms": round(plain, 4)}
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            call(gout, side.cuda_stream)
        res["graph_replay_ms"] = round(timeit(g.replay), 4)
        torch.cuda.synchronize()
        res["bit_exact"] = bool(torch.equal(out[0], gout[0]) and torch.equal(out[1], gout[1]))
    except Exception as e:
        res["graph_error"] = str(e)[:300]
    print(json.dumps(res), flush=True)
