"""Row-sharding of the single-frame kernels across ranks: Hough, Harris, stereo.

One frame is cut into `world` row bands (`row_cuts`).  What each path needs besides its own rows:

  * **Hough lines / circles** (a14, a15): nothing -- each rank votes for the edge points of its rows
    into a private full-size accumulator (`micv_hough_*_band_dev`: the band pointer plus its row
    offset), then ONE integer all-reduce (RCCL sum of int32; <= 3.2 MB at 1080p for lines) gives
    every rank the unsharded accumulator bit for bit (integer addition commutes).  Peaks are then
    found redundantly on every rank -- no second collective.
  * **stereo SSD / NCC** (a12, a13): `r` static halo rows of both images; rows are otherwise
    independent, so there is no exchange at all.  Clamp-to-edge addressing only ever acts at true
    image borders because the halo covers every tap of the band's windows.
  * **Harris response + NMS + corner list** (a8-a10): static image halo of
    `sobel//2 + window//2 + minDistance` rows (gradients, then the response window, then the NMS
    window), i.e. the `minDistance` rows of R the survey names are recomputed instead of
    exchanged; the per-rank corner lists, restricted to the band and offset to global rows,
    concatenate in rank order into the reference's row-major list (`gather_rows`, one all-gather of
    the variable-length lists).

The compute functions are injected (`*_fn` arguments): the HIP path on the GPU (`gpu_fns`), the
oracle in the CPU tests of the distributed logic (tests/test_shard_ops.py).
"""
import numpy as np


def row_cuts(rows, world):
    """Even row cuts: band of rank g = [cuts[g], cuts[g+1])."""
    if world < 1 or rows < world:
        raise ValueError(f"{world} ranks cannot split {rows} rows")
    return [(g * rows) // world for g in range(world)] + [rows]


def band_with_halo(rows, world, rank, halo):
    """((a, b), (lo, hi)): the rank's band and the rows it has to hold (band + static halo)."""
    cuts = row_cuts(rows, world)
    a, b = cuts[rank], cuts[rank + 1]
    return (a, b), (max(0, a - halo), min(rows, b + halo))


class TorchDist:
    """Collectives of torch.distributed (nccl = RCCL over xGMI on the GPU, gloo on CPU)."""

    def __init__(self, rank, world):
        import torch.distributed as dist
        self.dist, self.rank, self.world = dist, rank, world

    def all_reduce_sum(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t

    def all_gather_rows(self, t):
        """Concatenate per-rank [n_g, k] tensors in rank order (n_g differs per rank)."""
        import torch
        n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
        counts = [torch.zeros_like(n) for _ in range(self.world)]
        self.dist.all_gather(counts, n)
        counts = [int(c.item()) for c in counts]
        cap = max(max(counts), 1)
        pad = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        parts = [torch.empty_like(pad) for _ in range(self.world)]
        self.dist.all_gather(parts, pad)
        return torch.cat([p[:c] for p, c in zip(parts, counts)])


class LocalComm:
    """world == 1 (or tests that emulate the collective themselves)."""
    rank, world = 0, 1

    def all_reduce_sum(self, t):
        return t

    def all_gather_rows(self, t):
        return t


# ---- Hough ----------------------------------------------------------------------------------------

def hough_lines_sharded(mask_rows, band, rows, rho_bin, theta_bin, band_fn, comm):
    """mask_rows: this rank's rows [a, b) of the edge mask.  band_fn(mask_rows, row0, rows, rho_bin,
    theta_bin) -> private int32 accumulator of the full image's shape.  Returns the summed one."""
    a, b = band
    assert mask_rows.shape[0] == b - a
    return comm.all_reduce_sum(band_fn(mask_rows, a, rows, rho_bin, theta_bin))


def hough_circles_sharded(mask_rows, band, rows, radius, band_fn, comm):
    a, b = band
    assert mask_rows.shape[0] == b - a
    return comm.all_reduce_sum(band_fn(mask_rows, a, rows, radius))


# ---- stereo ---------------------------------------------------------------------------------------

def stereo_sharded(left_rows, right_rows, band, held, disparity_fn):
    """left_rows/right_rows: rows [lo, hi) = `held` (band + windowRad halo, clipped to the image).
    disparity_fn(left, right) -> int8 disparity of the same shape.  Returns rows [a, b)."""
    (a, b), (lo, hi) = band, held
    assert left_rows.shape[0] == hi - lo
    d = disparity_fn(left_rows, right_rows)
    return d[a - lo:b - lo]


# ---- Harris ---------------------------------------------------------------------------------------

def harris_halo(sobel_size, window_size, min_distance):
    return sobel_size // 2 + window_size // 2 + min_distance


def harris_sharded(img_rows, band, held, sobel_size, window_size, sigma, alpha, threshold,
                   min_distance, grad_fn, response_fn, refine_fn, comm):
    """img_rows: rows [lo, hi) = `held` of the f32 image (band + harris_halo rows).  Returns
    (R_band, corners_band, locs): the rank's rows of the response and of the sparse corner map, and
    the GLOBAL (y, x) corner list of the whole frame in row-major order (all-gathered)."""
    (a, b), (lo, hi) = band, held
    assert img_rows.shape[0] == hi - lo
    gx, gy = grad_fn(img_rows, sobel_size)
    resp = response_fn(gx, gy, window_size, sigma, alpha)
    corners, locs = refine_fn(resp, threshold, min_distance)
    ys = locs[:, 0] + lo
    keep = (ys >= a) & (ys < b)
    mine = locs[keep]
    if mine.shape[0]:
        mine = mine.clone() if hasattr(mine, "clone") else mine.copy()
        mine[:, 0] += lo
    return resp[a - lo:b - lo], corners[a - lo:b - lo], comm.all_gather_rows(mine)


# ---- HIP path ---------------------------------------------------------------------------------------

class gpu_fns:  # noqa: N801
    """The injected compute functions on the HIP path (torch CUDA tensors)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def hough_lines_band(self, mask_rows, row0, rows, rho_bin, theta_bin):
        import torch

        from . import _buf as B
        from ._capi import check, lib
        from .hough import linesAccumulatorShape
        cols = mask_rows.shape[1]
        rb, tb = linesAccumulatorShape(rows, cols, rho_bin, theta_bin)
        acc = torch.empty((rb, tb), dtype=torch.int32, device=mask_rows.device)
        check(lib.micv_hough_lines_band_dev(self.ctx.handle, mask_rows.data_ptr(), mask_rows.shape[0], cols,
                                            B.stride_bytes(mask_rows), int(row0), int(rows), int(rho_bin),
                                            int(theta_bin), acc.data_ptr(), B.stream_of(mask_rows)))
        return acc

    def hough_circles_band(self, mask_rows, row0, rows, radius):
        import torch

        from . import _buf as B
        from ._capi import check, lib
        cols = mask_rows.shape[1]
        acc = torch.empty((rows, cols), dtype=torch.int32, device=mask_rows.device)
        check(lib.micv_hough_circles_band_dev(self.ctx.handle, mask_rows.data_ptr(), mask_rows.shape[0], cols,
                                              B.stride_bytes(mask_rows), int(row0), int(rows), int(radius),
                                              acc.data_ptr(), B.stream_of(mask_rows)))
        return acc

    def grad(self, img, ksize):
        from . import harris
        return harris.getGradients(img, ksize, ctx=self.ctx)

    def response(self, gx, gy, win, sigma, alpha):
        from . import harris
        return harris.getCornerResponse(gx, gy, win, sigma, alpha, ctx=self.ctx)

    def refine(self, resp, thr, min_dist):
        from . import harris
        return harris.refineCorners(resp, thr, min_dist, ctx=self.ctx)

    def ssd(self, rad, min_d, max_d, flags=0):
        from . import stereo
        return lambda l, r: stereo.disparitySSD(l, r, rad, min_d, max_d, flags, ctx=self.ctx)

    def ncorr(self, rad, min_d, max_d, flags=0):
        from . import stereo
        return lambda l, r: stereo.disparityNCorr(l, r, rad, min_d, max_d, flags, ctx=self.ctx)
