#!/usr/bin/env python3
"""Host-side copy rates into / out of hipHostMalloc'd memory (what a pinned staging ring costs), 8.3 MB images."""
import ctypes as C, time, statistics
import numpy as np
hip = C.CDLL("libamdhip64.so")
libc = C.CDLL("libc.so.6")
vp = C.c_void_p
N = 1080 * 1920 * 4
def chk(rc):
    if rc: raise RuntimeError(f"hip error {rc}")
for flags, name in ((0, "default"), (0x80000000, "non-coherent"), (0x40000000, "coherent"), (0x2, "mapped"), (0x4, "write-combined")):
    p = vp()
    if hip.hipHostMalloc(C.byref(p), C.c_size_t(4 * N), C.c_uint(flags)):
        print(name, "hipHostMalloc failed"); continue
    C.memset(p, 1, 4 * N)
    src = [np.ones(N // 4, np.float32) for _ in range(4)]
    dst = [np.zeros(N // 4, np.float32) for _ in range(4)]
    tin, tout = [], []
    for it in range(12):
        k = it % 4
        t0 = time.perf_counter(); libc.memcpy(vp(p.value + k * N), vp(src[k].ctypes.data), C.c_size_t(N)); tin.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); libc.memcpy(vp(dst[k].ctypes.data), vp(p.value + k * N), C.c_size_t(N)); tout.append(time.perf_counter() - t0)
    print(f"{name:14s} pageable -> pinned {N / statistics.median(tin[4:]) / 1e9:6.1f} GB/s   pinned -> pageable {N / statistics.median(tout[4:]) / 1e9:6.1f} GB/s")
    hip.hipHostFree(p)
a, b = np.ones(N // 4, np.float32), np.zeros(N // 4, np.float32)
ts = []
for _ in range(12):
    t0 = time.perf_counter(); libc.memcpy(vp(b.ctypes.data), vp(a.ctypes.data), C.c_size_t(N)); ts.append(time.perf_counter() - t0)
print(f"pageable -> pageable {N / statistics.median(ts[4:]) / 1e9:6.1f} GB/s")
