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
        self._slabs = {}

    def exchange_batch(self, plan, level, flow):
        """Halo rows of level `level` for ALL pairs of a batch in one batch_isend_irecv: `flow` is the
        [B, 2, rows, cols] (u, v stacked) tensor of that level.  Send / receive slabs are allocated once
        per (level, peer) and reused; packing and unpacking are one strided copy each."""
        dist = self.dist
        ops, recvs = [], []
        for src, dst, r0, r1 in plan.transfers(level):
            if src != self.rank and dst != self.rank:
                continue
            key = (level, src, dst, r0, r1, tuple(flow.shape), flow.device)
            buf = self._slabs.get(key)
            if buf is None:
                buf = flow.new_empty((flow.shape[0], 2, r1 - r0, flow.shape[3]))
                self._slabs[key] = buf
            if src == self.rank:
                buf.copy_(flow[:, :, r0:r1])
                ops.append(dist.P2POp(dist.isend, buf, dst))
            else:
                recvs.append((buf, r0, r1))
                ops.append(dist.P2POp(dist.irecv, buf, src))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for buf, r0, r1 in recvs:
            flow[:, :, r0:r1].copy_(buf)

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

    def exchange_batch(self, *a):
        pass


IMAGE_MARGIN = 16  # level rows of `prev` a band reads beyond itself: Sobel 1 + window radius (<= 15 here)


class RowShardBatch:
    """A batch of frame pairs, every pair split by rows over `world` ranks: what `bench.py --mode
    rowshard` runs.  Per step and rank: ONE pyramid launch per image set restricted to the rows the
    band touches, ONE level launch for the band of all pairs (micv_lk_level_batch_dev), ONE batched
    halo exchange per level.  All buffers are allocated once.  Inputs are [B, rows, cols] tensors
    addressed by absolute row; run() writes this rank's band of every pair into u / v.

    Static halos: of `prev` a band reads only IMAGE_MARGIN rows beyond itself (a deployment ships
    band + margin, and the rank builds only those pyramid rows).  `next` is read at (y + dv) by the
    warp, and LK's flow is unbounded where the window is nearly degenerate (isolated pixels with
    |dv| of hundreds occur on the synthetic pairs).  Two ways to hold it:
      * next_margin=None (default): the whole `next` level on every rank -- exact, no check needed;
      * next_margin=m (level-0 rows, SURVEY.md section 8e's "declared bound on max |dv|"): the rank builds
        `next` for its band + IMAGE_MARGIN + ceil(m / 2^l) + 2 rows per level only, and a device-side check
        (micv_flow_bound_check_dev, no host sync) raises a flag when the coarse flow a level is about to
        expand exceeds what those rows cover.  run_checked() reads the flag after the step and, when it is
        set, repeats the step on the whole frame: the result is exact either way, the margin only decides
        how often the cheap path suffices."""

    def __init__(self, ctx, rows, cols, levels, win, batch, rank, world, comm=None, device=None, next_margin=None):
        import torch
        self.ctx, self.levels, self.win, self.batch, self.rank, self.world = ctx, levels, win, batch, rank, world
        if win // 2 + 1 > IMAGE_MARGIN:
            raise ValueError(f"window {win} reads more than IMAGE_MARGIN = {IMAGE_MARGIN} rows of prev beyond a band")
        self.plan = RowShardPlan(rows, cols, levels, world, win)
        self.comm = comm if comm is not None else _NoComm()
        self.band0 = self.plan.band(0, rank)
        dev = device if device is not None else torch.device("cuda", ctx.device)
        dims = self.plan.dims
        self.ppyr = [None] + [torch.empty((batch,) + dims[l], device=dev) for l in range(1, levels)]
        self.npyr = [None] + [torch.empty((batch,) + dims[l], device=dev) for l in range(1, levels)]
        self.flow = [None] + [torch.zeros((batch, 2) + dims[l], device=dev) for l in range(1, levels)]
        lo, hi = [], []
        for l in range(levels):
            a, b = self.plan.band(l, rank)
            lo.append(max(0, a - IMAGE_MARGIN))
            hi.append(min(dims[l][0], b + IMAGE_MARGIN))
        self.img_rows = (lo, hi)
        self.next_margin = next_margin
        self.next_rows = None
        self.flag = None
        if next_margin is not None:
            if next_margin < 0:
                raise ValueError("next_margin must be >= 0")
            nlo, nhi, self.level_margin = [], [], []
            for l in range(levels):
                a, b = self.plan.band(l, rank)
                m = -(-int(next_margin) // (1 << l)) + 2
                self.level_margin.append(m)
                nlo.append(max(0, a - IMAGE_MARGIN - m))
                nhi.append(min(dims[l][0], b + IMAGE_MARGIN + m))
            self.next_rows = (nlo, nhi)
            self.flag = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.full_next = next_margin is None  # which rows of `next` the pyramid holds right now

    def _pyr_ptrs(self, pyr):
        import ctypes as C
        arr = (C.c_void_p * self.levels)()
        arr[0] = None
        for l in range(1, self.levels):
            arr[l] = pyr[l].data_ptr()
        return arr

    def build_pyramids(self, prev, nxt, stream, restrict=True):
        import ctypes as C
        from ._capi import check, lib
        if self.levels < 2:
            return
        B, rows, cols = prev.shape
        lo = hi = None
        if restrict:
            lo = (C.c_int * self.levels)(*self.img_rows[0])
            hi = (C.c_int * self.levels)(*self.img_rows[1])
        nlo = nhi = None
        if restrict and self.next_rows is not None and not self.full_next:
            nlo = (C.c_int * self.levels)(*self.next_rows[0])
            nhi = (C.c_int * self.levels)(*self.next_rows[1])
        for src, pyr, a, b in ((prev, self.ppyr, lo, hi), (nxt, self.npyr, nlo, nhi)):
            check(lib.micv_gaussian_pyramid_batch_dev(self.ctx.handle, src.data_ptr(), B, rows * cols * 4, rows, cols,
                                                      cols * 4, self.levels, self._pyr_ptrs(pyr), a, b, stream))

    def level(self, l, prev, nxt, u, v, stream):
        """The band of level l for all pairs: one launch (+ one for the base flow on odd-sized levels)."""
        from ._capi import check, lib
        rows, cols = self.plan.dims[l]
        a, b = self.plan.band(l, self.rank)
        p = prev if l == 0 else self.ppyr[l]
        n = nxt if l == 0 else self.npyr[l]
        if l == 0:
            ou, ov, ostride = u.data_ptr(), v.data_ptr(), rows * cols * 4
        else:
            f = self.flow[l]
            ou, ov, ostride = f.data_ptr(), f.data_ptr() + rows * cols * 4, 2 * rows * cols * 4
        if l == self.levels - 1:
            fu = fv = None
            fr = fc = fstride = 0
        else:
            c = self.flow[l + 1]
            fr, fc = self.plan.dims[l + 1]
            fu, fv, fstride = c.data_ptr(), c.data_ptr() + fr * fc * 4, 2 * fr * fc * 4
        check(lib.micv_lk_level_batch_dev(self.ctx.handle, p.data_ptr(), n.data_ptr(), self.batch, rows * cols * 4,
                                          rows, cols, cols * 4, self.win, fu, fv, fr, fc, fstride, a, b, ou, ov,
                                          ostride, cols * 4, stream))

    def run(self, prev, nxt, u, v, stream=None):
        """Kernels go to `stream` (a raw HIP stream handle; default: torch's current stream).  The halo
        exchange is torch work -- slab copies and the RCCL point-to-point ops order themselves against
        torch's CURRENT stream -- so it is issued with `stream` made current: launches and exchange are
        then one in-order sequence whatever stream the caller passed (r02 raced when they differed)."""
        s, on_s = _launch_stream(stream, prev.device)
        self.build_pyramids(prev, nxt, s)
        for l in range(self.levels - 1, -1, -1):
            if l < self.levels - 1:
                with on_s:
                    self.comm.exchange_batch(self.plan, l + 1, self.flow[l + 1])
                if self.next_rows is not None and not self.full_next:
                    self._check_bound(l, s)
            self.level(l, prev, nxt, u, v, s)

    def _check_bound(self, l, stream):
        """Level l is about to warp `next` by 2 * pyrUp(v_{l+1}), a convex combination of the coarse rows this
        rank holds (its band + halo): |dv_l| <= 2 max |v_{l+1}| over them.  Flag when that exceeds the rows of
        `next` built for level l."""
        from ._capi import check, lib
        f = self.flow[l + 1]
        fr, fc = self.plan.dims[l + 1]
        r0, r1 = self.plan.needed(l + 1, self.rank)
        check(lib.micv_flow_bound_check_dev(self.ctx.handle, f.data_ptr() + fr * fc * 4, self.batch, 2 * fr * fc * 4,
                                            fr, fc, fc * 4, r0, r1, float(self.level_margin[l]) / 2.0,
                                            self.flag.data_ptr(), stream))

    def violated(self, stream=None):
        """True when the last run() met a vertical flow beyond the declared margin (host synchronisation).
        `stream`: the stream that run() was given -- the flag is written by a kernel on it, so it is read with
        that stream current (a read on another, non-blocking stream could come before the level kernels finish)."""
        if self.flag is None:
            return False
        _, on_s = _launch_stream(stream, self.flag.device)
        with on_s:
            return bool(self.flag.item())

    def run_checked(self, prev, nxt, u, v, stream=None):
        """run() with the declared margin; when the bound check fires, the same step again on the whole
        `next` frame.  Returns True when the margin sufficed.  Every rank must take the same branch (the
        exchange is collective in spirit): the flag is OR-reduced over the ranks when a process group exists."""
        if self.next_rows is None:
            self.run(prev, nxt, u, v, stream)
            return True
        # The flag is zeroed, written (micv_flow_bound_check_dev on the launch stream), reduced and read in ONE
        # in-order sequence: all of it is issued with the launch stream current (on torch's current
        # stream the zero could land after a check and .item() could read before the level kernels had run).
        _, on_s = _launch_stream(stream, prev.device)
        with on_s:
            self.flag.zero_()
        self.full_next = False
        self.run(prev, nxt, u, v, stream)
        with on_s:
            flag = self.flag
            dist = getattr(self.comm, "dist", None)
            if dist is not None and self.world > 1:
                flag = self.flag.clone()
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            ok = not bool(flag.item())
        if ok:
            return True
        self.full_next = True  # fall back: the whole `next` pyramid, no bound to check
        try:
            self.run(prev, nxt, u, v, stream)
        finally:
            self.full_next = False
        return False


def _launch_stream(stream, device):
    """(raw HIP stream handle, context manager that makes it torch's current stream).  Kernels of the C ABI go
    to the raw handle; torch work that must stay ordered with them (slab copies, collectives, flag reads) is
    issued inside the context."""
    import contextlib

    import torch
    cur = torch.cuda.current_stream(device)
    s = stream if stream is not None else cur.cuda_stream
    same = int(s or 0) == int(cur.cuda_stream or 0)
    return s, (contextlib.nullcontext() if same else torch.cuda.stream(torch.cuda.ExternalStream(int(s), device=device)))


def run_virtual_batch(runners, prev, nxt, u, v, stream=None, poison=None, shared_pyramids=True):
    """All ranks of a row-sharded batch on ONE device ("virtual shards"): the same
    band launches and the same transfer list as the distributed run, the exchange done by row copies
    between the ranks' private flow buffers.  The pyramids are built once and shared (the virtual
    ranks share the device's memory) unless shared_pyramids=False; `poison` overwrites rows a rank does not own after every level."""
    s, on_s = _launch_stream(stream, prev.device)
    r0 = runners[0]
    if shared_pyramids:
        r0.build_pyramids(prev, nxt, s, restrict=False)
        for r in runners[1:]:
            r.ppyr, r.npyr = r0.ppyr, r0.npyr
    else:  # as distributed ranks do: every rank builds the rows its band touches, into its own buffers
        for r in runners:
            r.build_pyramids(prev, nxt, s, restrict=True)
    with on_s:  # the row copies and poison fills are torch work: same stream as the launches
        for l in range(r0.levels - 1, -1, -1):
            if l < r0.levels - 1:
                for src, dst, a, b in r0.plan.transfers(l + 1):
                    runners[dst].flow[l + 1][:, :, a:b].copy_(runners[src].flow[l + 1][:, :, a:b])
            for r in runners:
                r.level(l, prev, nxt, u, v, s)
                if poison is not None and l > 0:
                    a, b = r.plan.band(l, r.rank)
                    r.flow[l][:, :, :a] = poison
                    r.flow[l][:, :, b:] = poison


def run_virtual_batch_checked(runners, prev, nxt, u, v, stream=None, poison=None):
    """run_checked() for virtual ranks: every rank builds only the rows of `next` its declared margin covers
    (runners built with next_margin=...), the bound checks run per rank and level, and when any rank's flag
    is raised ALL ranks repeat the step on the whole frame -- the decision a distributed run takes after
    OR-reducing the flags.  Returns True when the margin sufficed."""
    s, on_s = _launch_stream(stream, prev.device)
    r0 = runners[0]

    def one_pass(full):
        with on_s:  # flag fills, row copies, poison fills and the flag read: one in-order sequence with the launches
            return _one_pass(full)

    def _one_pass(full):
        for r in runners:
            r.full_next = full
            if r.flag is not None:
                r.flag.zero_()
            r.build_pyramids(prev, nxt, s, restrict=True)
        for l in range(r0.levels - 1, -1, -1):
            if l < r0.levels - 1:
                for src, dst, a, b in r0.plan.transfers(l + 1):
                    runners[dst].flow[l + 1][:, :, a:b].copy_(runners[src].flow[l + 1][:, :, a:b])
                if not full:
                    for r in runners:
                        r._check_bound(l, s)
            for r in runners:
                r.level(l, prev, nxt, u, v, s)
                if poison is not None and l > 0:
                    a, b = r.plan.band(l, r.rank)
                    r.flow[l][:, :, :a] = poison
                    r.flow[l][:, :, b:] = poison
        return any(r.violated(s) for r in runners) if not full else False

    try:
        if not one_pass(False):
            return True
        one_pass(True)
        return False
    finally:
        for r in runners:
            r.full_next = False


# ---- the C ABI's own multi-GPU path (csrc/comm.hip): what a C++ caller links, and what Python runs too --------

class MicvComm:
    """A micv_comm: the library's communicator over RCCL (loaded at run time by libmicv.so; the copy PyTorch already
    holds is shared).  Created from an RCCL unique id that rank 0 makes; `dist` (torch.distributed, any backend,
    initialised) carries the 128 bytes to the other ranks -- or pass `unique_id` yourself.  Collective, like
    ncclCommInitRank: every rank constructs it."""

    def __init__(self, ctx, rank=0, world=1, dist=None, unique_id=None):
        import ctypes as C

        from ._capi import MICV_COMM_ID_BYTES, check, lib
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        if unique_id is None:
            buf = (C.c_char * MICV_COMM_ID_BYTES)()
            if world > 1:
                if dist is None:
                    raise ValueError("world > 1 needs torch.distributed (or a unique_id) to share the RCCL unique id")
                # rank 0 ALWAYS broadcasts -- the id, or why there is none -- so that a failure on rank 0 does not leave
                # the other ranks waiting in the broadcast
                box = [None]
                if rank == 0:
                    try:
                        check(lib.micv_comm_unique_id(buf))
                        box = [bytes(buf.raw)]
                    except Exception as e:  # noqa: BLE001
                        box = [f"rank 0 could not make an RCCL unique id: {e}"]
                dist.broadcast_object_list(box, src=0)
                if not isinstance(box[0], (bytes, bytearray)):
                    raise RuntimeError(str(box[0]))
                unique_id = bytes(box[0])
            else:
                check(lib.micv_comm_unique_id(buf))
                unique_id = bytes(buf.raw)
        if len(unique_id) != MICV_COMM_ID_BYTES:
            raise ValueError(f"an RCCL unique id has {MICV_COMM_ID_BYTES} bytes")
        self.unique_id = bytes(unique_id)
        h = C.c_void_p()
        check(lib.micv_comm_create(ctx.handle, None, self.unique_id, self.rank, self.world, C.byref(h)))
        self._h = h

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("MicvComm used after close()")
        return self._h

    def close(self):
        from ._capi import lib
        if self._h:
            lib.micv_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def selftest(self, stream=None):
        """micv_comm_selftest: ring send / receive of a rank-stamped slab + an int32 all-reduce, verified on the device.
        Collective; raises MicvError naming rank, peer and step when the fabric or the call ordering is broken."""
        import torch

        from ._capi import check, lib
        s = stream if stream is not None else torch.cuda.current_stream(self.ctx.device).cuda_stream
        check(lib.micv_comm_selftest(self.ctx.handle, self.handle, s))

    def allreduce_sum_i32(self, t, stream=None):
        """In-place int32 sum over the ranks (the Hough accumulator merge) on `stream` / the current stream."""
        import torch

        from ._capi import check, lib
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise ValueError("a contiguous int32 tensor is expected")
        s = stream if stream is not None else torch.cuda.current_stream(t.device).cuda_stream
        check(lib.micv_allreduce_sum_i32_dev(self.ctx.handle, self.handle, t.data_ptr(), t.numel(), s))
        return t


def native_band(rows, cols, levels, world, win, rank, level=0, needed=False):
    """micv_rowshard_band: the C ABI's row plan (host only) -- equal to RowShardPlan by construction, and by test."""
    import ctypes as C

    from ._capi import check, lib
    a, b, n0, n1 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    check(lib.micv_rowshard_band(rows, cols, levels, world, win, rank, level, C.byref(a), C.byref(b), C.byref(n0), C.byref(n1)))
    return ((a.value, b.value), (n0.value, n1.value)) if needed else (a.value, b.value)


class RowShardNative:
    """RowShardBatch's job done by the library itself: micv_lk_flow_pyr_rowshard_dev builds the pyramids, launches
    the bands and exchanges the coarse-flow halo with ncclSend / ncclRecv on the launch stream -- no Python between
    the levels, and the same entry point a C++ caller (shim lk::calcOpticalFlowPyr with a communicator) uses.
    Whole `next` frames on every rank (the exact default); the declared-margin variant stays in RowShardBatch."""

    def __init__(self, ctx, rows, cols, levels, win, batch, comm):
        self.ctx, self.rows, self.cols, self.levels, self.win, self.batch, self.comm = ctx, rows, cols, levels, win, batch, comm
        self.band0 = native_band(rows, cols, levels, comm.world, win, comm.rank, 0)

    def run(self, prev, nxt, u, v, stream=None):
        import torch

        from ._capi import check, lib
        B, rows, cols = prev.shape
        if (B, rows, cols) != (self.batch, self.rows, self.cols):
            raise ValueError("shape differs from the runner's")
        s = stream if stream is not None else torch.cuda.current_stream(prev.device).cuda_stream
        check(lib.micv_lk_flow_pyr_rowshard_dev(self.ctx.handle, self.comm.handle, prev.data_ptr(), nxt.data_ptr(), B,
                                                rows * cols * 4, rows, cols, cols * 4, self.win, self.levels, u.data_ptr(),
                                                v.data_ptr(), rows * cols * 4, cols * 4, s))


def run_virtual_native(ctx, world, prev, nxt, win, levels, poison=True, stream=None):
    """micv_lk_flow_pyr_rowshard_virtual_dev: `world` virtual ranks of the C ABI's driver on one device (same plan,
    packing and band launches; the transport is a copy between the ranks' slabs).  Returns whole (u, v)."""
    import torch

    from ._capi import check, lib
    B, rows, cols = prev.shape
    u = torch.full_like(prev, float("nan"))
    v = torch.full_like(prev, float("nan"))
    s = stream if stream is not None else torch.cuda.current_stream(prev.device).cuda_stream
    check(lib.micv_lk_flow_pyr_rowshard_virtual_dev(ctx.handle, int(world), prev.data_ptr(), nxt.data_ptr(), B, rows * cols * 4,
                                                    rows, cols, cols * 4, int(win), int(levels), u.data_ptr(), v.data_ptr(),
                                                    rows * cols * 4, cols * 4, 1 if poison else 0, s))
    return u, v
