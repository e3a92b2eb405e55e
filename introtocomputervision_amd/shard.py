"""Row-sharded pyramidal Lucas-Kanade: one frame pair split across ranks by rows.

The north star sketches "frames shard row-wise across the 8 GPUs with a one-row halo exchanged over
RCCL/xGMI".  What exact parity needs is more than one row, and only part of it is
dynamic:

  * images are static inputs: every rank keeps the (replicated or over-read) level images it
    touches -- prev rows +-(1 + win//2), next rows further out by the warp displacement -- so
    there is NO per-level image exchange (here each rank simply holds the full 8 MB frames);
  * the flow is produced level by level, so the only dynamic exchange is the coarse flow: to
    compute rows [a, b) of level l a rank needs 2*pyrUp(flow_{l+1}) on rows [a-8, b+8) (Sobel 1 +
    window 7), i.e. coarse rows [(a-8-2)/2, (b+8+2)/2]: up to 5 rows beyond its own coarse band, 6-7
    with the cv::resize of odd-sized levels.  The halo follows the window: halo_rows(win) =
    (win//2 + 2)//2 + 3 coarse rows (7 for win 15, 11 for win 31) x cols x 2 fields x 4 B per
    neighbour per level (level 1 -> 0 at 1080p: 54 KB): latency-bound point-to-point, no collective.

Cuts: the coarsest level is split evenly; finer levels double the cut (a_l = 2 a_{l+1}), the last
rank absorbs the odd remainder, so a rank's band at level l is exactly the pyrUp image of its band at
level l+1.  Bands are exchanged with torch.distributed point-to-point ops (backend nccl = RCCL over
xGMI on the GPU, gloo in the CPU tests); a band smaller than the halo simply receives from more than
one neighbour.  Results are bit-identical to the unsharded path (tests/test_shard.py).
"""
import numpy as np

def halo_rows(win):
    """Coarse-flow rows a band needs beyond its own coarse band to compute the next finer level with
    window `win`: the fine rows [a - r - 1, b + r + 1) (Sobel 1 + window radius r) read 2*pyrUp on
    +-2 fine rows, i.e. coarse rows a/2 - ceil((r+3)/2) .. b/2 + (r+2)//2: (r+2)//2 + 1 rows, + 2 for
    the cv::resize of odd-sized levels (source row within 1 of the target row, uneven last band)."""
    r = int(win) // 2
    return (r + 2) // 2 + 3


class RowShardPlan:
    def __init__(self, rows, cols, levels, world, win=15):
        self.levels, self.world, self.win = levels, world, int(win)
        self.halo = halo_rows(win)
        self.dims = [(rows >> l, cols >> l) for l in range(levels)]
        if self.dims[-1][0] < world:
            raise ValueError(f"{world} ranks cannot split the {self.dims[-1][0]}-row coarsest level")
        top = self.dims[-1][0]
        cuts = [[(g * top) // world for g in range(world)] + [top]]
        for l in range(levels - 2, -1, -1):
            finer = [2 * c for c in cuts[0][:-1]] + [self.dims[l][0]]
            cuts.insert(0, finer)
        self.cuts = cuts  # cuts[l][g] .. cuts[l][g+1] = band of rank g at level l

    def band(self, level, rank):
        return self.cuts[level][rank], self.cuts[level][rank + 1]

    def needed(self, level, rank):
        """Rows of level `level` (a coarse level) rank needs to compute its band one level finer."""
        a, b = self.band(level, rank)
        return max(0, a - self.halo), min(self.dims[level][0], b + self.halo)

    def transfers(self, level):
        """[(src, dst, row0, row1)]: rows of level `level` that dst needs and src owns."""
        out = []
        for dst in range(self.world):
            n0, n1 = self.needed(level, dst)
            for src in range(self.world):
                if src == dst:
                    continue
                a, b = self.band(level, src)
                r0, r1 = max(a, n0), min(b, n1)
                if r0 < r1:
                    out.append((src, dst, r0, r1))
        return out

    def halo_bytes(self, level, rank):
        cols = self.dims[level][1]
        return sum((r1 - r0) * cols * 8 for s, d, r0, r1 in self.transfers(level) if d == rank)


class DistComm:
    """Halo exchange over torch.distributed point-to-point (nccl = RCCL on GPU, gloo on CPU)."""

    def __init__(self, rank, world):
        import torch.distributed as dist
        self.dist, self.rank, self.world = dist, rank, world

    def exchange(self, plan, level, fu, fv):
        import torch
        dist = self.dist
        ops, keep = [], []
        for src, dst, r0, r1 in plan.transfers(level):
            if src == self.rank:
                buf = torch.stack([fu[r0:r1], fv[r0:r1]]).contiguous()
                keep.append(buf)
                ops.append(dist.P2POp(dist.isend, buf, dst))
            elif dst == self.rank:
                buf = torch.empty((2, r1 - r0, fu.shape[1]), dtype=fu.dtype, device=fu.device)
                keep.append((buf, r0, r1))
                ops.append(dist.P2POp(dist.irecv, buf, src))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for k in keep:
            if isinstance(k, tuple):
                buf, r0, r1 = k
                fu[r0:r1] = buf[0]
                fv[r0:r1] = buf[1]


def lk_pyr_band(prev_pyr, next_pyr, plan, rank, win, level_fn, comm, poison=None):
    """Coarse-to-fine LK for the row band of `rank`.

    prev_pyr / next_pyr: per-level images (level 0 first).  level_fn(prev_l, next_l, fu, fv, a, b)
    returns (u, v) full-size arrays/tensors whose rows [a, b) are valid (fu/fv None at the coarsest
    level).  comm.exchange(plan, level, fu, fv) fills the halo rows of level `level` in place.
    Returns (u, v) of level 0 with rows plan.band(0, rank) valid."""
    if halo_rows(win) > plan.halo:
        raise ValueError(f"plan was built for window {plan.win} (halo {plan.halo} rows); window {win} "
                         f"needs {halo_rows(win)}: pass win to RowShardPlan")
    fu = fv = None
    for l in range(plan.levels - 1, -1, -1):
        a, b = plan.band(l, rank)
        if fu is not None:
            comm.exchange(plan, l + 1, fu, fv)
        u, v = level_fn(prev_pyr[l], next_pyr[l], fu, fv, a, b)
        if poison is not None:  # tests: make any read of a row we do not own visible
            u[:a] = poison
            u[b:] = poison
            v[:a] = poison
            v[b:] = poison
        fu, fv = u, v
    return fu, fv


def lk_pyr_virtual(prev_pyr, next_pyr, plan, win, level_fn, poison=12345.0):
    """All ranks of `plan` executed in one process ("virtual shards"): the same band
    arithmetic and the same transfer list, with the exchange done by row copies.  Rows a rank does
    not own are overwritten with `poison` after every level, so a missing halo row shows up as a
    wrong result.  Returns the assembled level-0 (u, v)."""
    if halo_rows(win) > plan.halo:
        raise ValueError(f"plan was built for window {plan.win}; window {win} needs a wider halo")
    state = [None] * plan.world  # per rank: (fu, fv) of the previous (coarser) level
    for l in range(plan.levels - 1, -1, -1):
        if state[0] is not None:
            for src, dst, r0, r1 in plan.transfers(l + 1):
                state[dst][0][r0:r1] = state[src][0][r0:r1]
                state[dst][1][r0:r1] = state[src][1][r0:r1]
        new = []
        for g in range(plan.world):
            a, b = plan.band(l, g)
            fu, fv = state[g] if state[g] is not None else (None, None)
            u, v = level_fn(prev_pyr[l], next_pyr[l], fu, fv, a, b)
            u[:a] = poison
            u[b:] = poison
            v[:a] = poison
            v[b:] = poison
            new.append((u, v))
        state = new
    u = state[0][0].clone() if hasattr(state[0][0], "clone") else state[0][0].copy()
    v = state[0][1].clone() if hasattr(state[0][1], "clone") else state[0][1].copy()
    for g in range(plan.world):
        a, b = plan.band(0, g)
        u[a:b] = state[g][0][a:b]
        v[a:b] = state[g][1][a:b]
    return u, v


def gpu_level_fn(ctx, win, stream=None):
    """level_fn on the HIP path (micv_lk_level_dev) for torch CUDA tensors."""
    import torch

    from ._capi import check, lib

    def fn(prev_l, next_l, fu, fv, a, b):
        rows, cols = prev_l.shape
        u = torch.empty_like(prev_l)
        v = torch.empty_like(prev_l)
        s = stream if stream is not None else torch.cuda.current_stream(prev_l.device).cuda_stream
        if fu is None:
            check(lib.micv_lk_level_dev(ctx.handle, prev_l.data_ptr(), next_l.data_ptr(), rows, cols,
                                        cols * 4, win, None, None, 0, 0, a, b, u.data_ptr(),
                                        v.data_ptr(), cols * 4, s))
        else:
            fu = fu.contiguous()
            fv = fv.contiguous()
            check(lib.micv_lk_level_dev(ctx.handle, prev_l.data_ptr(), next_l.data_ptr(), rows, cols,
                                        cols * 4, win, fu.data_ptr(), fv.data_ptr(), fu.shape[0],
                                        fu.shape[1], a, b, u.data_ptr(), v.data_ptr(), cols * 4, s))
        return u, v

    return fn


def lk_pyr_row_sharded_gpu(prev, nxt, win, levels, ctx, rank, world):
    """One frame pair, row-sharded over `world` ranks (torch.distributed initialised, backend nccl).
    prev/nxt: full [rows, cols] float32 CUDA tensors on every rank.  Returns this rank's band
    (u_band, v_band, (a, b))."""
    from . import pyr
    plan = RowShardPlan(prev.shape[0], prev.shape[1], levels, world, win)
    pp = pyr.makeGaussianPyramid(prev, levels, ctx=ctx)
    npyr = pyr.makeGaussianPyramid(nxt, levels, ctx=ctx)
    u, v = lk_pyr_band(pp, npyr, plan, rank, win, gpu_level_fn(ctx, win), DistComm(rank, world))
    a, b = plan.band(0, rank)
    return u[a:b], v[a:b], (a, b)


class _NoComm:
    def exchange(self, *a):
        pass


class RowShardBatch:
    """A batch of frame pairs, every pair split by rows over `world` ranks: what `bench.py --mode
    rowshard` runs.  Inputs are replicated ([B, rows, cols] on every rank); run() writes this rank's
    band of every pair into u / v."""

    def __init__(self, ctx, rows, cols, levels, win, batch, rank, world, comm=None):
        self.ctx, self.levels, self.win, self.batch, self.rank = ctx, levels, win, batch, rank
        self.plan = RowShardPlan(rows, cols, levels, world, win)
        self.comm = comm if comm is not None else _NoComm()
        self.band0 = self.plan.band(0, rank)
        self.level_fn = gpu_level_fn(ctx, win)

    def run(self, prev, nxt, u, v, stream=None):
        from . import pyr
        a0, b0 = self.band0
        for i in range(self.batch):
            pp = pyr.makeGaussianPyramid(prev[i], self.levels, ctx=self.ctx)
            npyr = pyr.makeGaussianPyramid(nxt[i], self.levels, ctx=self.ctx)
            bu, bv = lk_pyr_band(pp, npyr, self.plan, self.rank, self.win, self.level_fn, self.comm)
            u[i, a0:b0] = bu[a0:b0]
            v[i, a0:b0] = bv[a0:b0]
