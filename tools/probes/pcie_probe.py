#!/usr/bin/env python3
"""What the host-pointer path can get out of PCIe on this box: 8.3 MB (one 1080p f32 image) copies,
pageable vs pinned host memory, one stream vs two concurrent streams, and both directions at once.
Prints GB/s per case (median of 20).  GPU box: python tools/probes/pcie_probe.py"""
import ctypes as C, statistics, time
import numpy as np
hip = C.CDLL("libamdhip64.so")
vp = C.c_void_p
def chk(rc):
    if rc: raise RuntimeError(f"hip error {rc}")
N = 1080 * 1920 * 4
H2D, D2H = 1, 2
streams = []
for _ in range(4):
    s = vp(); chk(hip.hipStreamCreateWithFlags(C.byref(s), 1)); streams.append(s)
dev = []
for _ in range(4):
    d = vp(); chk(hip.hipMalloc(C.byref(d), C.c_size_t(N))); dev.append(d)
pageable = [np.ones(N // 4, np.float32) for _ in range(4)]
pinned = []
for _ in range(4):
    p = vp(); chk(hip.hipHostMalloc(C.byref(p), C.c_size_t(N), 0)); C.memset(p, 1, N); pinned.append(p)
def run(copies):
    """copies: list of (host_ptr, dev_ptr, kind, stream)"""
    ts = []
    for it in range(24):
        for s in streams: chk(hip.hipStreamSynchronize(s))
        t0 = time.perf_counter()
        for h, d, kind, s in copies:
            if kind == H2D: chk(hip.hipMemcpyAsync(d, h, C.c_size_t(N), kind, s))
            else: chk(hip.hipMemcpyAsync(h, d, C.c_size_t(N), kind, s))
        for s in streams: chk(hip.hipStreamSynchronize(s))
        ts.append(time.perf_counter() - t0)
    t = statistics.median(ts[4:])
    return len(copies) * N / t / 1e9, t * 1e3
pg = [vp(a.ctypes.data) for a in pageable]
cases = {
    "H2D pageable x1": [(pg[0], dev[0], H2D, streams[0])],
    "H2D pageable x2 one stream": [(pg[0], dev[0], H2D, streams[0]), (pg[1], dev[1], H2D, streams[0])],
    "H2D pageable x2 two streams": [(pg[0], dev[0], H2D, streams[0]), (pg[1], dev[1], H2D, streams[1])],
    "H2D pinned x1": [(pinned[0], dev[0], H2D, streams[0])],
    "H2D pinned x2 one stream": [(pinned[0], dev[0], H2D, streams[0]), (pinned[1], dev[1], H2D, streams[0])],
    "H2D pinned x2 two streams": [(pinned[0], dev[0], H2D, streams[0]), (pinned[1], dev[1], H2D, streams[1])],
    "D2H pageable x1": [(pg[0], dev[0], D2H, streams[0])],
    "D2H pageable x2 one stream": [(pg[0], dev[0], D2H, streams[0]), (pg[1], dev[1], D2H, streams[0])],
    "D2H pageable x2 two streams": [(pg[0], dev[0], D2H, streams[0]), (pg[1], dev[1], D2H, streams[1])],
    "D2H pinned x2 one stream": [(pinned[0], dev[0], D2H, streams[0]), (pinned[1], dev[1], D2H, streams[0])],
    "D2H pinned x2 two streams": [(pinned[0], dev[0], D2H, streams[0]), (pinned[1], dev[1], D2H, streams[1])],
    "duplex pageable (2 up + 2 down, 4 streams)": [(pg[0], dev[0], H2D, streams[0]), (pg[1], dev[1], H2D, streams[1]),
                                                   (pg[2], dev[2], D2H, streams[2]), (pg[3], dev[3], D2H, streams[3])],
    "duplex pinned (2 up + 2 down, 4 streams)": [(pinned[0], dev[0], H2D, streams[0]), (pinned[1], dev[1], H2D, streams[1]),
                                                 (pinned[2], dev[2], D2H, streams[2]), (pinned[3], dev[3], D2H, streams[3])],
    "serial pageable (2 up then 2 down, one stream) = the _host call's transfers": [
        (pg[0], dev[0], H2D, streams[0]), (pg[1], dev[1], H2D, streams[0]), (pg[2], dev[2], D2H, streams[0]), (pg[3], dev[3], D2H, streams[0])],
}
for name, c in cases.items():
    gbs, ms = run(c)
    print(f"{name:80s} {gbs:7.1f} GB/s  {ms:7.3f} ms")
# CPU memcpy rate (staging through a pinned ring would pay this)
a, b = pageable[0], pageable[1]
t0 = time.perf_counter()
for _ in range(20): np.copyto(b, a)
print(f"{'host memcpy 8.3 MB, one thread':80s} {20 * N / (time.perf_counter() - t0) / 1e9:7.1f} GB/s")
