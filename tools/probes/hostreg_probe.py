#!/usr/bin/env python3
"""hipHostRegister / hipHostUnregister of an 8.3 MB pageable image (one 1080p f32 frame), and copies from registered memory:
what pinning the caller's cv::Mat in place would cost the frame-sequence entry.  GPU box: python tools/probes/hostreg_probe.py"""
import ctypes as C, statistics, time
import numpy as np
hip = C.CDLL("libamdhip64.so")
vp = C.c_void_p
def chk(rc):
    if rc: raise RuntimeError(f"hip error {rc}")
N = 1080 * 1920 * 4
arrs = [np.ones(N // 4, np.float32) for _ in range(8)]
reg, unreg = [], []
for rep in range(5):
    for a in arrs:
        t0 = time.perf_counter(); chk(hip.hipHostRegister(vp(a.ctypes.data), C.c_size_t(N), 0)); reg.append(time.perf_counter() - t0)
    for a in arrs:
        t0 = time.perf_counter(); chk(hip.hipHostUnregister(vp(a.ctypes.data))); unreg.append(time.perf_counter() - t0)
print(f"hipHostRegister 8.3 MB: median {statistics.median(reg)*1e3:.3f} ms (min {min(reg)*1e3:.3f}); unregister {statistics.median(unreg)*1e3:.3f} ms")
# duplex from registered memory, two streams
s = [vp(), vp()]
for x in s: chk(hip.hipStreamCreateWithFlags(C.byref(x), 1))
d = [vp(), vp(), vp()]
for x in d: chk(hip.hipMalloc(C.byref(x), C.c_size_t(N)))
for a in arrs[:3]: chk(hip.hipHostRegister(vp(a.ctypes.data), C.c_size_t(N), 0))
ts = []
for it in range(24):
    for x in s: chk(hip.hipStreamSynchronize(x))
    t0 = time.perf_counter()
    chk(hip.hipMemcpyAsync(d[0], vp(arrs[0].ctypes.data), C.c_size_t(N), 1, s[0]))
    chk(hip.hipMemcpyAsync(vp(arrs[1].ctypes.data), d[1], C.c_size_t(N), 2, s[1]))
    chk(hip.hipMemcpyAsync(vp(arrs[2].ctypes.data), d[2], C.c_size_t(N), 2, s[1]))
    for x in s: chk(hip.hipStreamSynchronize(x))
    ts.append(time.perf_counter() - t0)
print(f"registered memory: 1 upload + 2 downloads of 8.3 MB on two streams: {statistics.median(ts[4:])*1e3:.3f} ms")
# first use of a fresh registration vs steady state (the sequence entry would register per call)
for rep in range(3):
    b = np.ones(N // 4, np.float32)
    t0 = time.perf_counter(); chk(hip.hipHostRegister(vp(b.ctypes.data), C.c_size_t(N), 0)); t1 = time.perf_counter()
    chk(hip.hipMemcpyAsync(d[0], vp(b.ctypes.data), C.c_size_t(N), 1, s[0])); chk(hip.hipStreamSynchronize(s[0])); t2 = time.perf_counter()
    chk(hip.hipMemcpyAsync(d[0], vp(b.ctypes.data), C.c_size_t(N), 1, s[0])); chk(hip.hipStreamSynchronize(s[0])); t3 = time.perf_counter()
    chk(hip.hipMemcpyAsync(vp(b.ctypes.data), d[0], C.c_size_t(N), 2, s[0])); chk(hip.hipStreamSynchronize(s[0])); t4 = time.perf_counter()
    chk(hip.hipHostUnregister(vp(b.ctypes.data))); t5 = time.perf_counter()
    print(f"fresh image: register {1e3*(t1-t0):.3f} ms, first upload {1e3*(t2-t1):.3f} ms, second upload {1e3*(t3-t2):.3f} ms, download {1e3*(t4-t3):.3f} ms, unregister {1e3*(t5-t4):.3f} ms")
big = np.ones(15 * N // 4, np.float32)
t0 = time.perf_counter(); chk(hip.hipHostRegister(vp(big.ctypes.data), C.c_size_t(15 * N), 0)); t1 = time.perf_counter()
for k in range(3):
    t2 = time.perf_counter(); chk(hip.hipMemcpyAsync(vp(big.ctypes.data + k * N), d[0], C.c_size_t(N), 2, s[0])); chk(hip.hipStreamSynchronize(s[0])); t3 = time.perf_counter()
    print(f"124 MB registration ({1e3*(t1-t0):.3f} ms): download of slice {k}: {1e3*(t3-t2):.3f} ms")
chk(hip.hipHostUnregister(vp(big.ctypes.data)))
