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
    alt = os.environ.get("ORACLE_SO")  # another build of the same checker, e.g. oracle/liboracle_asan.so
    if alt:
        return C.CDLL(alt if os.path.isabs(alt) else os.path.join(ROOT, alt))
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
_cvround = _sig("orc_cv_round", i32, [f32])
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
_togray = _sig("orc_to_gray_f32", i32, [vp, i32, i32, sz, i32, i32, vp, sz])


def reflect101(p, n):
    return _reflect(p, n)


def cv_round(v):
    return _cvround(float(np.float32(v)))


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


_lkpat = _sig("orc_lk_flow_pyr_at", i32, [vp, vp, i32, i32, sz, i32, i32, i32, i32, vp, vp, sz])


def lk_flow_pyr_at(prev, nxt, win, levels, oy, ox):
    """lk_flow_pyr on a crop whose pixel (0, 0) is pixel (oy, ox) of a larger frame (multiples of 2^(levels - 1))."""
    prev = _f(prev); nxt = _f(nxt)
    r, c = prev.shape
    u = np.empty_like(prev); v = np.empty_like(prev)
    rc = _lkpat(_p(prev), _p(nxt), r, c, c, win, levels, int(oy), int(ox), _p(u), _p(v), c)
    if rc:
        raise ValueError(f"orc_lk_flow_pyr_at rc={rc}")
    return u, v


def lk_flow_pyr(prev, nxt, win, levels):
    prev = _f(prev); nxt = _f(nxt)
    r, c = prev.shape
    u = np.empty_like(prev); v = np.empty_like(prev)
    rc = _lkp(_p(prev), _p(nxt), r, c, c, win, levels, _p(u), _p(v), c)
    if rc:
        raise ValueError(f"orc_lk_flow_pyr rc={rc}")
    return u, v


_sepcv = _sig("orc_sep_filter_cvcpu", None, [vp, i32, i32, sz, vp, i32, vp, i32, vp, sz, i32])
_lkex = _sig("orc_lk_flow_ex", i32, [vp, vp, i32, i32, sz, i32, i32, vp, vp, sz, vp])
_lkpex = _sig("orc_lk_flow_pyr_ex", i32, [vp, vp, i32, i32, sz, i32, i32, i32, vp, vp, sz, vp])
VAR_BLUR_CVCPU, VAR_BLUR_FUSED = 1, 2          # oracle.h: bounding variants of the window sums
HARRIS_GPU, HARRIS_CPU, HARRIS_GPU_FMAD = 0, 1, 2


def sep_filter_cvcpu(src, krow, kcol, fused=False):
    """OpenCV's CPU FilterEngine order (row left->right, column folded from the centre): a bounding variant."""
    src = _f(src); krow = _f(krow); kcol = _f(kcol)
    r, c = src.shape
    out = np.empty_like(src)
    _sepcv(_p(src), r, c, c, _p(krow), len(krow), _p(kcol), len(kcol), _p(out), c, int(bool(fused)))
    return out


def lk_flow_ex(prev, nxt, win, variant=0, want_det=False):
    prev = _f(prev); nxt = _f(nxt)
    r, c = prev.shape
    u = np.empty_like(prev); v = np.empty_like(prev)
    det = np.empty((r, c), np.float64) if want_det else None
    rc = _lkex(_p(prev), _p(nxt), r, c, c, win, variant, _p(u), _p(v), c, _p(det) if want_det else None)
    if rc:
        raise ValueError(f"orc_lk_flow_ex rc={rc}")
    return (u, v, det) if want_det else (u, v)


def lk_flow_pyr_ex(prev, nxt, win, levels, variant=0, want_det=False):
    prev = _f(prev); nxt = _f(nxt)
    r, c = prev.shape
    u = np.empty_like(prev); v = np.empty_like(prev)
    det = np.empty((r, c), np.float64) if want_det else None
    rc = _lkpex(_p(prev), _p(nxt), r, c, c, win, levels, variant, _p(u), _p(v), c, _p(det) if want_det else None)
    if rc:
        raise ValueError(f"orc_lk_flow_pyr_ex rc={rc}")
    return (u, v, det) if want_det else (u, v)


_hrespex = _sig("orc_harris_response_ex", i32, [vp, vp, i32, i32, sz, i32, f64, f32, i32, vp, sz])
_hresp = _sig("orc_harris_response", i32, [vp, vp, i32, i32, sz, i32, f64, f32, vp, sz])
_hrefine = _sig("orc_harris_refine", i64, [vp, i32, i32, sz, f64, i32, vp, sz, vp, i64])
_sang = _sig("orc_sift_angles", None, [vp, vp, i32, i32, sz, vp, sz])
_skp = _sig("orc_sift_keypoints", None, [vp, vp, i32, i32, sz, vp, i64, f32, vp])
_ssd = _sig("orc_disparity_ssd", i32, [vp, vp, i32, i32, sz, i32, i32, i32, i32, vp, sz])
_ssds = _sig("orc_disparity_ssd_serial", i32, [vp, vp, i32, i32, sz, i32, i32, i32, vp, sz])
_ncc = _sig("orc_disparity_ncorr", i32, [vp, vp, i32, i32, sz, i32, i32, i32, i32, vp, sz])
_hdims = _sig("orc_hough_lines_dims", None, [i32, i32, C.c_uint, C.c_uint, C.POINTER(i32), C.POINTER(i32)])
_hlines = _sig("orc_hough_lines", i32, [vp, i32, i32, sz, C.c_uint, C.c_uint, vp])
_hcirc = _sig("orc_hough_circles", i32, [vp, i32, i32, sz, C.c_uint, vp])
_hpeaks = _sig("orc_hough_peaks", i64, [vp, i32, i32, C.c_uint, i32, vp])


def harris_response(gx, gy, win, sigma, alpha):
    gx = _f(gx); gy = _f(gy)
    r, c = gx.shape
    out = np.empty_like(gx)
    rc = _hresp(_p(gx), _p(gy), r, c, c, win, float(sigma), float(np.float32(alpha)), _p(out), c)
    if rc:
        raise ValueError(f"orc_harris_response rc={rc}")
    return out


def harris_response_ex(gx, gy, win, sigma, alpha, mode):
    """mode: HARRIS_GPU (the contract), HARRIS_CPU (harris::cpu as written), HARRIS_GPU_FMAD (bounding variant)."""
    gx = _f(gx); gy = _f(gy)
    r, c = gx.shape
    out = np.empty_like(gx)
    rc = _hrespex(_p(gx), _p(gy), r, c, c, win, float(sigma), float(np.float32(alpha)), int(mode), _p(out), c)
    if rc:
        raise ValueError(f"orc_harris_response_ex rc={rc}")
    return out


def harris_refine(resp, threshold, min_distance):
    resp = _f(resp)
    r, c = resp.shape
    corners = np.empty_like(resp)
    locs = np.empty((r * c, 2), np.int32)
    n = _hrefine(_p(resp), r, c, c, float(threshold), int(min_distance), _p(corners), c, _p(locs), r * c)
    return corners, locs[:n].copy()


_sdesc = _sig("orc_sift_descriptors", i32, [vp, vp, i32, i32, sz, vp, i64, vp, sz])


def sift_descriptors(gx, gy, kps):
    """kps: [n, 4] float32 (x, y, size, angle_deg) -> [n, 128] float32 (8-bit values)."""
    gx = _f(gx); gy = _f(gy)
    kps = np.ascontiguousarray(kps, dtype=np.float32).reshape(-1, 4)
    r, c = gx.shape
    out = np.zeros((len(kps), 128), np.float32)
    rc = _sdesc(_p(gx), _p(gy), r, c, c, _p(kps), len(kps), _p(out), 128)
    if rc:
        raise ValueError(f"orc_sift_descriptors rc={rc}")
    return out


def sift_angles(gx, gy):
    gx = _f(gx); gy = _f(gy)
    r, c = gx.shape
    out = np.empty_like(gx)
    _sang(_p(gx), _p(gy), r, c, c, _p(out), c)
    return out


def sift_keypoints(gx, gy, locs, size):
    gx = _f(gx); gy = _f(gy)
    locs = np.ascontiguousarray(locs, dtype=np.int32)
    r, c = gx.shape
    kp = np.empty((len(locs), 4), np.float32)
    _skp(_p(gx), _p(gy), r, c, c, _p(locs), len(locs), float(size), _p(kp))
    return kp


def disparity_ssd(left, right, rad, min_d, max_d, flags=0):
    left = _f(left); right = _f(right)
    r, c = left.shape
    out = np.empty((r, c), np.int8)
    rc = _ssd(_p(left), _p(right), r, c, c, rad, min_d, max_d, flags, _p(out), c)
    if rc:
        raise ValueError(f"orc_disparity_ssd rc={rc}")
    return out


def disparity_ssd_serial(left, right, rad, min_d, max_d):
    left = _f(left); right = _f(right)
    r, c = left.shape
    out = np.empty((r, c), np.int8)
    rc = _ssds(_p(left), _p(right), r, c, c, rad, min_d, max_d, _p(out), c)
    if rc:
        raise ValueError(f"orc_disparity_ssd_serial rc={rc}")
    return out


def disparity_ncorr(left, right, rad, min_d, max_d, flags=0):
    left = _f(left); right = _f(right)
    r, c = left.shape
    out = np.empty((r, c), np.int8)
    rc = _ncc(_p(left), _p(right), r, c, c, rad, min_d, max_d, flags, _p(out), c)
    if rc:
        raise ValueError(f"orc_disparity_ncorr rc={rc}")
    return out


def hough_lines_dims(rows, cols, rho_bin=1, theta_bin=1):
    rb, tb = i32(), i32()
    _hdims(rows, cols, rho_bin, theta_bin, C.byref(rb), C.byref(tb))
    return rb.value, tb.value


def hough_lines(mask, rho_bin=1, theta_bin=1):
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    r, c = mask.shape
    rb, tb = hough_lines_dims(r, c, rho_bin, theta_bin)
    acc = np.empty((rb, tb), np.int32)
    rc = _hlines(_p(mask), r, c, c, rho_bin, theta_bin, _p(acc))
    if rc:
        raise ValueError(f"orc_hough_lines rc={rc}")
    return acc


def hough_circles(mask, radius):
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    r, c = mask.shape
    acc = np.empty((r, c), np.int32)
    _hcirc(_p(mask), r, c, c, radius, _p(acc))
    return acc


def hough_peaks(acc, num_peaks, threshold):
    acc = np.ascontiguousarray(acc, dtype=np.int32)
    r, c = acc.shape
    peaks = np.empty((max(num_peaks, 1), 2), np.uint32)
    n = _hpeaks(_p(acc), r, c, num_peaks, threshold, _p(peaks))
    return peaks[:n].copy()


_mhi_fd = _sig("orc_mhi_frame_difference", i32, [vp, vp, i32, i32, sz, f64, i32, i32, f64, vp, sz])
_mhi_en = _sig("orc_mhi_energy", None, [vp, sz, vp])
_mhi_thr = _sig("orc_mhi_threshold", None, [vp, sz, f64, vp])
_mhi_upd = _sig("orc_mhi_update", None, [vp, sz, vp, sz, i32, i32, i32])


def mhi_frame_difference(f1, f2, thresh, ksize=3, sigma=1.0):
    """ksize: an int (square) or the reference's cv::Size as (width, height)."""
    f1 = np.ascontiguousarray(f1, dtype=np.uint8); f2 = np.ascontiguousarray(f2, dtype=np.uint8)
    r, c = f1.shape
    out = np.empty((r, c), np.uint8)
    kw, kh = ksize if isinstance(ksize, (tuple, list)) else (ksize, ksize)
    rc = _mhi_fd(_p(f1), _p(f2), r, c, c, float(thresh), int(kw), int(kh), float(sigma), _p(out), c)
    if rc:
        raise ValueError(f"orc_mhi_frame_difference rc={rc}")
    return out


def mhi_energy(mhi):
    mhi = np.ascontiguousarray(mhi, dtype=np.uint8)
    out = np.empty_like(mhi)
    _mhi_en(_p(mhi), mhi.size, _p(out))
    return out


def mhi_threshold(src, thresh):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    out = np.empty_like(src)
    _mhi_thr(_p(src), src.size, float(thresh), _p(out))
    return out


def mhi_update(history, mask, tau):
    h = np.ascontiguousarray(history, dtype=np.uint8).copy()
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    r, c = h.shape
    _mhi_upd(_p(h), c, _p(m), c, r, c, tau)
    return h


def to_gray(img):
    """[rows, cols] or [rows, cols, 3|4], uint8 or float32 -> grey float32 (Pyramids.cpp:9-15)."""
    img = np.ascontiguousarray(img)
    assert img.dtype in (np.uint8, np.float32)
    cn = 1 if img.ndim == 2 else img.shape[2]
    r, c = img.shape[:2]
    out = np.empty((r, c), np.float32)
    rc = _togray(_p(img), r, c, c * cn * img.dtype.itemsize, cn, 0 if img.dtype == np.uint8 else 5, _p(out), c)
    assert rc == 0
    return out


def rgb8_to_gray(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    r, c, _ = rgb.shape
    out = np.empty((r, c), np.float32)
    _gray(_p(rgb), r, c, c * 3, _p(out), c)
    return out
