#!/usr/bin/env python3
"""The host-pointer path as a cv::Mat caller sees it: micv_lk_flow_pyr_host on one 1080p pair, pageable
inputs, PREALLOCATED (touched) outputs -- cv::Mat::create() is a no-op on a Mat of the right size
(OpticalFlow.cpp:53-54) -- against the same call on resident device buffers.  GPU box."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from introtocomputervision_amd import synth, _capi
from introtocomputervision_amd._capi import lib, check
rows, cols, win, levels = 1080, 1920, 15, 5
p, n = synth.lk_pair(0x5EED0005, rows, cols, 3, -2)
u = np.zeros((rows, cols), np.float32); v = np.zeros((rows, cols), np.float32)
ctx = _capi.Context(0)
def host_call():
    check(lib.micv_lk_flow_pyr_host(ctx.handle, p.ctypes.data, n.ctypes.data, rows, cols, cols * 4, win, levels,
                                    u.ctypes.data, v.ctypes.data, cols * 4))
def wall(fn, iters=40, warm=5):
    for _ in range(warm): fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3
host_ms = wall(host_call)
dp, dn = torch.from_numpy(p).cuda(), torch.from_numpy(n).cuda()
du, dv = torch.empty_like(dp), torch.empty_like(dp)
def dev_call():
    check(lib.micv_lk_flow_pyr_dev(ctx.handle, dp.data_ptr(), dn.data_ptr(), rows, cols, cols * 4, win, levels,
                                   du.data_ptr(), dv.data_ptr(), cols * 4, None))
    torch.cuda.synchronize()
dev_ms = wall(dev_call)
ok = bool(np.array_equal(u, du.cpu().numpy()) and np.array_equal(v, dv.cpu().numpy()))
print(json.dumps({"host_pair_ms": round(host_ms, 4), "device_pair_ms_with_sync": round(dev_ms, 4),
                  "bytes_over_pcie": 4 * rows * cols * 4, "pcie_GBps_incl_compute": round(4 * rows * cols * 4 / host_ms / 1e6, 1),
                  "same_result": ok}))
