"""Synthetic inputs of SURVEY.md §8(d): integer-exact, language-independent.

splitmix64(seed) stream -> top 8 bits per pixel -> three passes of an integer 5x5 box sum
(/25 round-half-up, edge-clamped) -> float32.  Used by tests and bench.py (host side, numpy).
"""
import numpy as np

_M = (1 << 64) - 1


def splitmix64_u8(seed, n):
    """First n outputs of splitmix64(seed), top 8 bits each."""
    i = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & _M) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(56)).astype(np.uint8)


def _box5(a):
    p = np.pad(a.astype(np.int64), 2, mode="edge")
    rows, cols = a.shape
    s = np.zeros((rows + 4, cols), dtype=np.int64)
    for k in range(5):
        s += p[:, k:k + cols]
    t = np.zeros((rows, cols), dtype=np.int64)
    for k in range(5):
        t += s[k:k + rows, :]
    return (t + 12) // 25


def smooth_noise(seed, rows, cols, passes=3):
    """u8-valued float32 texture."""
    a = splitmix64_u8(seed, rows * cols).reshape(rows, cols).astype(np.int64)
    for _ in range(passes):
        a = _box5(a)
    return a.astype(np.float32)


def lk_pair(seed, rows, cols, dx=3, dy=-2):
    """C2/C4 frame pair: next = prev circularly shifted by (dx, dy) px -> flow u ~ dx, v ~ dy."""
    prev = smooth_noise(seed, rows, cols)
    nxt = np.roll(prev, shift=(dy, dx), axis=(0, 1))
    return prev, np.ascontiguousarray(nxt)


def stereo_pair(seed, rows, cols):
    """C3 rectified pair: right(y,x) = left(y, x + d(y)), d(y) = 8 + floor(96 y / rows)
    (columns past the edge clamp), so the left-reference disparity is -d(y)."""
    left = smooth_noise(seed, rows, cols)
    d = 8 + (96 * np.arange(rows)) // rows
    xs = np.clip(np.arange(cols)[None, :] + d[:, None], 0, cols - 1)
    right = np.take_along_axis(left, xs, axis=1)
    return left, np.ascontiguousarray(right), -d


def hough_mask(rows, cols, n_lines=12, radii=(20, 24, 28, 32, 36, 40), seed=0x5EED0001):
    """Binary (0/255) edge mask with `n_lines` straight lines at theta in {-60..75 step 15} deg
    (rho chosen by the splitmix stream) and one circle per radius.  Returns (mask, lines, circles)
    with lines = [(rho, theta_deg)], circles = [(cy, cx, r)]."""
    mask = np.zeros((rows, cols), np.uint8)
    rnd = splitmix64_u8(seed, 4 * (n_lines + len(radii)) + 8).astype(np.int64)
    lines, circles = [], []
    yy, xx = np.mgrid[0:rows, 0:cols]
    for i in range(n_lines):
        theta = -60 + 15 * (i % 10)
        t = np.deg2rad(theta)
        # a line through a pseudo-random interior point
        py = rows // 4 + (rnd[4 * i] * rows // 2) // 256
        px = cols // 4 + (rnd[4 * i + 1] * cols // 2) // 256
        rho = px * np.cos(t) + py * np.sin(t)
        d = np.abs(xx * np.cos(t) + yy * np.sin(t) - rho)
        mask[d < 0.5] = 255
        lines.append((float(rho), theta))
    for j, r in enumerate(radii):
        k = 4 * (n_lines + j)
        cy = r + 2 + (rnd[k] * (rows - 2 * r - 4)) // 256
        cx = r + 2 + (rnd[k + 1] * (cols - 2 * r - 4)) // 256
        d = np.abs(np.hypot(yy - cy, xx - cx) - r)
        mask[d < 0.5] = 255
        circles.append((int(cy), int(cx), int(r)))
    return mask, lines, circles


def checkerboard(rows, cols, square=40, lo=64, hi=192, seed=None, amp=4):
    """C1 checkerboard (+- amp of the same noise stream when seed is given)."""
    yy, xx = np.mgrid[0:rows, 0:cols]
    img = np.where(((yy // square) + (xx // square)) % 2 == 0, lo, hi).astype(np.int64)
    if seed is not None:
        n = splitmix64_u8(seed, rows * cols).reshape(rows, cols).astype(np.int64)
        img = img + (n % (2 * amp + 1)) - amp
    return img.astype(np.float32)
