"""Driver-level plumbing of ps5 in numpy: the Python mirror of
shim/micv_viz.hpp -- PNM / BMP files, drawVelocityVectors (ps5_cpp/src/Solution.cpp:13-37), the min-max
normalisation + JET colour maps of denseLKWrapper (:66-79).  Host code, no kernels; same arithmetic as the
C++ header, statement by statement, so the two can be compared byte for byte (tests/test_viz.py).
OpenCV's own drawing / colour-map code is not available here: parity with it is unpinned (micv_viz.hpp)."""
import math

import numpy as np


def imread(path):
    """P5 / P6 (maxval <= 255) or uncompressed BMP -> uint8 [rows, cols] or [rows, cols, 3] in B, G, R order."""
    raw = open(path, "rb").read()
    if raw[:2] in (b"P5", b"P6"):
        toks, pos = [], 2
        while len(toks) < 3:
            while raw[pos:pos + 1].isspace():
                pos += 1
            if raw[pos:pos + 1] == b"#":
                pos = raw.index(b"\n", pos) + 1
                continue
            end = pos
            while not raw[end:end + 1].isspace():
                end += 1
            toks.append(int(raw[pos:end]))
            pos = end
        w, h, maxv = toks
        assert 0 < maxv <= 255
        cn = 3 if raw[:2] == b"P6" else 1
        a = np.frombuffer(raw, np.uint8, w * h * cn, pos + 1).reshape(h, w, cn)
        return np.ascontiguousarray(a[:, :, ::-1]) if cn == 3 else np.ascontiguousarray(a[:, :, 0])
    if raw[:2] == b"BM":
        from PIL import Image  # BMP decoding only
        img = Image.open(path)
        if img.mode in ("L", "P") and (img.mode == "L" or all(img.getpalette()[3 * i] == img.getpalette()[3 * i + 1] ==
                                                             img.getpalette()[3 * i + 2] for i in range(256))):
            return np.asarray(img.convert("L"))
        return np.ascontiguousarray(np.asarray(img.convert("RGB"))[:, :, ::-1])
    raise ValueError(f"{path}: neither P5 / P6 nor BMP")


def imwrite(path, img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    with open(path, "wb") as f:
        if img.ndim == 3:
            f.write(b"P6\n%d %d\n255\n" % (w, h))
            f.write(np.ascontiguousarray(img[:, :, ::-1]).tobytes())
        else:
            f.write(b"P5\n%d %d\n255\n" % (w, h))
            f.write(img.tobytes())


def _cv_round(v):
    return int(np.rint(v))  # half to even, like lrint


def _put(img, x, y, color):
    if 0 <= x < img.shape[1] and 0 <= y < img.shape[0]:
        img[y, x] = color


def line(img, p1, p2, color):
    """cv::line, thickness 1, LINE_8: cv::LineIterator's integer walk, left to right."""
    if p1[0] > p2[0]:
        p1, p2 = p2, p1
    dx, dy = p2[0] - p1[0], p2[1] - p1[1]
    sy = -1 if dy < 0 else 1
    dy = abs(dy)
    steep = dy > dx
    major, minor = (dy, dx) if steep else (dx, dy)
    err, x, y = major - 2 * minor, p1[0], p1[1]
    for _ in range(major + 1):
        _put(img, x, y, color)
        both = err < 0
        err += 2 * major - 2 * minor if both else -2 * minor
        if steep:
            y += sy
            x += 1 if both else 0
        else:
            x += 1
            y += sy if both else 0


def arrowed_line(img, x1, y1, x2, y2, color):
    p1 = (_cv_round(np.float32(x1)), _cv_round(np.float32(y1)))
    p2 = (_cv_round(np.float32(x2)), _cv_round(np.float32(y2)))
    ddx, ddy = float(p1[0] - p2[0]), float(p1[1] - p2[1])
    tip = math.sqrt(ddx * ddx + ddy * ddy) * 0.1
    line(img, p1, p2, color)
    angle, q = math.atan2(ddy, ddx), 3.14159265358979323846 / 4
    for s in (q, -q):
        p = (_cv_round(p2[0] + tip * math.cos(angle + s)), _cv_round(p2[1] + tip * math.sin(angle + s)))
        line(img, p, p2, color)


def drawVelocityVectors(inputImg, u, v, color=(0, 255, 0)):
    """Solution.cpp:13-37 -> a new [rows, cols, 3] uint8 image (B, G, R)."""
    img = np.ascontiguousarray(inputImg, np.uint8)
    if img.ndim == 2:
        img = np.repeat(img[:, :, None], 3, axis=2)
    img = img.copy()
    rows, cols = u.shape
    rs, cs = max(1, rows // 30), max(1, cols // 30)
    for y in range(0, rows, rs):
        for x in range(0, cols, cs):
            uv, vv = np.float32(u[y, x]), np.float32(v[y, x])
            if not (np.isfinite(uv) and np.isfinite(vv)) or abs(uv) > 1e6 or abs(vv) > 1e6:
                continue
            arrowed_line(img, np.float32(x), np.float32(y), np.float32(x) + uv, np.float32(y) + vv, color)
    return img


def normalize_minmax_u8(src):
    """cv::normalize(src, dst, 0, 255, NORM_MINMAX, CV_8U) of a float32 field."""
    src = np.asarray(src, np.float32)
    lo, hi = float(np.nanmin(src)), float(np.nanmax(src))
    scale = 255.0 * (1.0 / (hi - lo) if hi - lo > np.finfo(np.float64).eps else 0.0)
    a, b = np.float32(scale), np.float32(0.0 - lo * scale)
    with np.errstate(invalid="ignore", over="ignore"):
        t = src * a + b
    out = np.zeros(src.shape, np.uint8)
    ok = np.isfinite(t)
    out[ok] = np.clip(np.rint(t[ok]), 0, 255).astype(np.uint8)
    return out


def jet_lut():
    lut = np.zeros((256, 3), np.uint8)
    for i in range(256):
        x = i / 255.0
        ramp = lambda t: 0.0 if t < 0 else (1.0 if t > 1 else t)  # noqa: E731
        r, g, b = ramp(1.5 - abs(4 * x - 3)), ramp(1.5 - abs(4 * x - 2)), ramp(1.5 - abs(4 * x - 1))
        lut[i] = (_cv_round(b * 255), _cv_round(g * 255), _cv_round(r * 255))
    return lut


def apply_colormap_jet(src):
    return jet_lut()[np.asarray(src, np.uint8)]
