"""numpy front-end of oracle/liboracle.so (the CPU restatement; TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The library is (re)built with gcc when missing or older than its sources.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ODIR, "liboracle.so")


def build(force=False):
    srcs = [os.path.join(ODIR, f) for f in os.listdir(ODIR) if f.endswith((".c", ".h")) or f == "Makefile"]
    stale = force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", ODIR, "-B", "liboracle.so"], check=True, capture_output=True)
    return SO


def _load():
    build()
    try:
        return C.CDLL(SO)
    except OSError:
        build(force=True)  # e.g. built on a host with different ISA extensions
        return C.CDLL(SO)


L = _load()
fp = C.POINTER(C.c_float)
i32, sz, f32, f64, i64 = C.c_int, C.c_size_t, C.c_float, C.c_double, C.c_int64
vp = C.c_void_p


def _p(a):
    return a.ctypes.data_as(vp)


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _sig(name, res, args):
    fn = getattr(L, name)
    fn.restype = res
    fn.argtypes = args
    return fn


_reflect = _sig("orc_reflect101", i32, [i32, i32])
_gk = _sig("orc_gaussian_kernel", None, [i32, f64, vp])
_sep = _sig("orc_sep_filter", None, [vp, i32, i32, sz, vp, i32, vp, i32, vp, sz])
_sobel = _sig("orc_sobel", i32, [vp, i32, i32, sz, i32, f32, vp, vp, sz])
_lk = _sig("orc_lk_flow", i32, [vp, vp, i32, i32, sz, i32, vp, vp, sz])
_remap = _sig("orc_remap_linear", None, [vp, i32, i32, sz, vp, vp, sz, vp, i32, i32, sz])
_warp = _sig("orc_lk_warp", None, [vp, vp, vp, i32, i32, sz, vp])
_resize = _sig("orc_resize_linear", None, [vp, i32, i32, sz, vp, i32, i32, sz])
_pd = _sig("orc_pyr_down", None, [vp, i32, i32, sz, vp, sz])
_pu = _sig("orc_pyr_up", None, [vp, i32, i32, sz, vp, sz])
_lkp = _sig("orc_lk_flow_pyr", i32, [vp, vp, i32, i32, sz, i32, i32, vp, vp, sz])
_gray = _sig("orc_rgb8_to_gray_f32", None, [vp, i32, i32, sz, vp, sz])


def reflect101(p, n):
    return _reflect(p, n)


def gaussian_kernel(n, sigma):
    out = np.zeros(n, np.float32)
    _gk(n, float(sigma), _p(out))
    return out


def sep_filter(src, krow, kcol):
    src = _f(src); krow = _f(krow); kcol = _f(kcol)
    r, c = src.shape
    out = np.empty_like(src)
    _sep(_p(src), r, c, c, _p(krow), len(krow), _p(kcol), len(kcol), _p(out), c)
    return out


def sobel(src, ksize=3, scale=1.0):
    src = _f(src)
    r, c = src.shape
    gx = np.empty_like(src); gy = np.empty_like(src)
    rc = _sobel(_p(src), r, c, c, ksize, float(np.float32(scale)), _p(gx), _p(gy), c)
    if rc:
        raise ValueError(f"orc_sobel rc={rc}")
    return gx, gy


def lk_flow(prev, nxt, win):
    prev = _f(prev); nxt = _f(nxt)
    r, c = prev.shape
    u = np.empty_like(prev); v = np.empty_like(prev)
    rc = _lk(_p(prev), _p(nxt), r, c, c, win, _p(u), _p(v), c)
    if rc:
        raise ValueError(f"orc_lk_flow rc={rc}")
    return u, v


def remap_linear(src, mapx, mapy):
    src = _f(src); mapx = _f(mapx); mapy = _f(mapy)
    r, c = src.shape
    dr, dc = mapx.shape
    out = np.empty((dr, dc), np.float32)
    _remap(_p(src), r, c, c, _p(mapx), _p(mapy), dc, _p(out), dr, dc, dc)
    return out


def lk_warp(src, du, dv):
    src = _f(src); du = _f(du); dv = _f(dv)
    r, c = src.shape
    out = np.empty_like(src)
    _warp(_p(src), _p(du), _p(dv), r, c, c, _p(out))
    return out


def resize_linear(src, drows, dcols):
    src = _f(src)
    r, c = src.shape
    out = np.empty((drows, dcols), np.float32)
    _resize(_p(src), r, c, c, _p(out), drows, dcols, dcols)
    return out


def pyr_down(src):
    src = _f(src)
    r, c = src.shape
    out = np.empty((r // 2, c // 2), np.float32)
    if out.size:
        _pd(_p(src), r, c, c, _p(out), c // 2)
    return out


def pyr_up(src):
    src = _f(src)
    r, c = src.shape
    out = np.empty((2 * r, 2 * c), np.float32)
    _pu(_p(src), r, c, c, _p(out), 2 * c)
    return out


def gaussian_pyramid(src, levels):
    out = [_f(src).copy()]
    for _ in range(1, levels):
        out.append(pyr_down(out[-1]))
    return out


def lk_flow_pyr(prev, nxt, win, levels):
    prev = _f(prev); nxt = _f(nxt)
    r, c = prev.shape
    u = np.empty_like(prev); v = np.empty_like(prev)
    rc = _lkp(_p(prev), _p(nxt), r, c, c, win, levels, _p(u), _p(v), c)
    if rc:
        raise ValueError(f"orc_lk_flow_pyr rc={rc}")
    return u, v


def rgb8_to_gray(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    r, c, _ = rgb.shape
    out = np.empty((r, c), np.float32)
    _gray(_p(rgb), r, c, c * 3, _p(out), c)
    return out
