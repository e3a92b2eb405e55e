"""Row-sharded pyramidal LK (introtocomputervision_amd/shard.py): bit-identical to the unsharded
result.  CPU: the sharding / halo logic with the oracle as the per-band compute, in one process
("virtual shards") and across 2 real ranks over gloo.  GPU: virtual shards on one device through
micv_lk_level_dev (the same entry point the RCCL path uses)."""
import os
import socket
import sys

import numpy as np
import pytest

import _oracle as orc
from introtocomputervision_amd import shard, synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def oracle_level_fn(win):
    """One iteration of OpticalFlow.cpp:135-163 on full level arrays (rows outside the band valid
    only where their inputs were)."""
    def fn(prev_l, next_l, fu, fv, a, b):
        rows, cols = prev_l.shape
        if fu is None:
            du = np.zeros((rows, cols), np.float32)
            dv = np.zeros((rows, cols), np.float32)
        else:
            du = 2.0 * orc.pyr_up(fu)
            dv = 2.0 * orc.pyr_up(fv)
            if du.shape != (rows, cols):
                du = orc.resize_linear(du, rows, cols)
                dv = orc.resize_linear(dv, rows, cols)
        warped = orc.lk_warp(next_l, du, dv)
        dx, dy = orc.lk_flow(prev_l, warped, win)
        return du + dx, dv + dy
    return fn


def test_plan_geometry():
    p = shard.RowShardPlan(1080, 1920, 5, 8)
    assert p.dims == [(1080, 1920), (540, 960), (270, 480), (135, 240), (67, 120)]
    for l in range(5):
        assert p.cuts[l][0] == 0 and p.cuts[l][-1] == p.dims[l][0]
        assert all(p.cuts[l][g] < p.cuts[l][g + 1] for g in range(8))
    for l in range(4):  # a band is the pyrUp image of the coarser band
        assert all(p.cuts[l][g] == 2 * p.cuts[l + 1][g] for g in range(8))
    # level 1 -> 0 halo at win 15: 7 rows x 960 cols x 2 fields x 4 B from each neighbour
    assert p.halo == shard.halo_rows(15) == 7
    assert p.halo_bytes(1, 3) == 2 * 7 * 960 * 8
    assert p.halo_bytes(1, 0) == 7 * 960 * 8
    assert shard.halo_rows(31) == 11 and shard.halo_rows(7) == 5
    with pytest.raises(ValueError):
        shard.RowShardPlan(64, 64, 5, 8)


@pytest.mark.parametrize("rows,cols,levels,world", [(135, 120, 3, 2), (135, 96, 4, 3), (270, 96, 5, 8), (101, 77, 3, 4)])
def test_virtual_shards_match_unsharded_oracle(rows, cols, levels, world):
    prev, nxt = synth.lk_pair(31 + world, rows, cols, 3, -2)
    eu, ev = orc.lk_flow_pyr(prev, nxt, 7, levels)
    plan = shard.RowShardPlan(rows, cols, levels, world)
    u, v = shard.lk_pyr_virtual(orc.gaussian_pyramid(prev, levels), orc.gaussian_pyramid(nxt, levels),
                                plan, 7, oracle_level_fn(7))
    assert np.array_equal(u, eu) and np.array_equal(v, ev)


def test_native_plan_equals_python_plan():
    """micv_rowshard_band (csrc/comm.hip, the plan the C ABI's row-shard driver walks) against RowShardPlan: bands
    and needed rows of every rank and level, for sizes with odd levels, uneven cuts and several windows.  Host only."""
    for rows, cols, levels, world, win in [(1080, 1920, 5, 8, 15), (1080, 1920, 5, 3, 21), (270, 480, 4, 2, 15), (333, 517, 3, 5, 7),
                                           (2160, 3840, 5, 8, 15), (67, 120, 1, 4, 15), (135, 240, 2, 7, 31), (100, 100, 3, 1, 43)]:
        plan = shard.RowShardPlan(rows, cols, levels, world, win)
        for l in range(levels):
            for g in range(world):
                band, need = shard.native_band(rows, cols, levels, world, win, g, l, needed=True)
                assert band == plan.band(l, g) and need == plan.needed(l, g), (rows, cols, levels, world, win, l, g)
    from introtocomputervision_amd._capi import MicvError
    with pytest.raises(MicvError):
        shard.native_band(270, 480, 5, 17, 15, 0)  # a 16-row coarsest level cannot be cut 17 ways
    with pytest.raises(MicvError):
        shard.native_band(270, 480, 3, 2, 15, 2)   # rank out of range


@pytest.mark.parametrize("win,rows,cols,levels,world", [(21, 135, 96, 3, 3), (31, 135, 64, 3, 2), (31, 101, 77, 3, 4), (43, 160, 48, 2, 3)])
def test_halo_follows_the_window(win, rows, cols, levels, world):
    """Wide windows need more coarse rows than the 7 of win 15: the plan derives the halo from win
    (rows a rank does not own are poisoned after every level, so a short halo shows)."""
    prev, nxt = synth.lk_pair(7 + win, rows, cols, 2, -1)
    eu, ev = orc.lk_flow_pyr(prev, nxt, win, levels)
    plan = shard.RowShardPlan(rows, cols, levels, world, win)
    u, v = shard.lk_pyr_virtual(orc.gaussian_pyramid(prev, levels), orc.gaussian_pyramid(nxt, levels),
                                plan, win, oracle_level_fn(win))
    assert np.array_equal(u, eu) and np.array_equal(v, ev)
    with pytest.raises(ValueError):  # a plan built for a narrower window refuses the wide one
        shard.lk_pyr_virtual(orc.gaussian_pyramid(prev, levels), orc.gaussian_pyramid(nxt, levels),
                             shard.RowShardPlan(rows, cols, levels, world, 7), win, oracle_level_fn(win))


def test_halo_is_actually_needed():
    """With the halo exchange removed the bands no longer match: the test above is not vacuous."""
    prev, nxt = synth.lk_pair(5, 135, 96, 3, -2)
    eu, _ = orc.lk_flow_pyr(prev, nxt, 7, 3)
    plan = shard.RowShardPlan(135, 96, 3, 2)
    plan.transfers = lambda level: []
    u, _ = shard.lk_pyr_virtual(orc.gaussian_pyramid(prev, 3), orc.gaussian_pyramid(nxt, 3), plan, 7,
                                oracle_level_fn(7))
    assert not np.array_equal(u, eu)


def _gloo_worker(rank, world, port, rows, cols, levels, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prev, nxt = synth.lk_pair(77, rows, cols, 3, -2)
    plan = shard.RowShardPlan(rows, cols, levels, world)
    base = oracle_level_fn(7)

    def level_fn(p, n, fu, fv, a, b):  # torch tensors <-> oracle (numpy)
        u, v = base(p, n, None if fu is None else fu.numpy(), None if fv is None else fv.numpy(), a, b)
        return torch.from_numpy(u), torch.from_numpy(v)

    u, v = shard.lk_pyr_band(orc.gaussian_pyramid(prev, levels), orc.gaussian_pyramid(nxt, levels), plan,
                             rank, 7, level_fn, shard.DistComm(rank, world), poison=12345.0)
    a, b = plan.band(0, rank)
    np.save(os.path.join(out_dir, f"u{rank}.npy"), u[a:b].numpy())
    np.save(os.path.join(out_dir, f"v{rank}.npy"), v[a:b].numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_over_gloo(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    rows, cols, levels, world = 135, 96, 3, 2
    mp.spawn(_gloo_worker, args=(world, port, rows, cols, levels, str(tmp_path)), nprocs=world, join=True)
    prev, nxt = synth.lk_pair(77, rows, cols, 3, -2)
    eu, ev = orc.lk_flow_pyr(prev, nxt, 7, levels)
    u = np.concatenate([np.load(tmp_path / f"u{r}.npy") for r in range(world)])
    v = np.concatenate([np.load(tmp_path / f"v{r}.npy") for r in range(world)])
    assert np.array_equal(u, eu) and np.array_equal(v, ev)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,levels,world,win", [(270, 480, 5, 8, 15), (540, 960, 5, 4, 15), (135, 240, 4, 3, 9),
                                                       (1080, 1920, 5, 8, 15)])  # the last one is BASELINE C4's geometry
def test_virtual_shards_on_gpu(rows, cols, levels, world, win):
    import torch
    from introtocomputervision_amd import lk, pyr
    from introtocomputervision_amd._capi import Context
    ctx = Context(0)
    prev, nxt = synth.lk_pair(1234, rows, cols, 3, -2)
    dp, dn = torch.from_numpy(prev).cuda(), torch.from_numpy(nxt).cuda()
    gu, gv = lk.calcOpticalFlowPyr(dp, dn, win, levels, ctx=ctx)
    plan = shard.RowShardPlan(rows, cols, levels, world, win)
    u, v = shard.lk_pyr_virtual(pyr.makeGaussianPyramid(dp, levels, ctx=ctx),
                                pyr.makeGaussianPyramid(dn, levels, ctx=ctx), plan, win,
                                shard.gpu_level_fn(ctx, win))
    assert torch.equal(u, gu) and torch.equal(v, gv)
    eu, ev = orc.lk_flow_pyr(prev, nxt, win, levels)
    assert np.array_equal(u.cpu().numpy(), eu) and np.array_equal(v.cpu().numpy(), ev)


def _gloo_batch_worker(rank, world, port, out_dir):
    """exchange_batch over gloo: every rank fills its own band of a [B, 2, rows, cols] field with a
    known pattern and poison elsewhere; after the exchange the rows it needs hold the owners' values."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = shard.RowShardPlan(270, 96, 4, world, 15)
    comm = shard.DistComm(rank, world)
    ok = True
    for rep in range(2):  # second round reuses the slabs
        for level in (1, 2, 3):
            rows, cols = plan.dims[level]
            truth = (torch.arange(3 * 2 * rows * cols, dtype=torch.float32).reshape(3, 2, rows, cols) + 1000 * level + rep)
            f = torch.full_like(truth, -777.0)
            a, b = plan.band(level, rank)
            f[:, :, a:b] = truth[:, :, a:b]
            comm.exchange_batch(plan, level, f)
            n0, n1 = plan.needed(level, rank)
            ok = ok and torch.equal(f[:, :, n0:n1], truth[:, :, n0:n1])
            ok = ok and bool((f[:, :, :n0] == -777.0).all()) and bool((f[:, :, n1:] == -777.0).all())
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def test_batched_halo_exchange_over_gloo(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 3
    mp.spawn(_gloo_batch_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert [open(tmp_path / f"ok{r}").read() for r in range(world)] == ["1"] * world


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,levels,world,batch,shared", [(1080, 1920, 5, 8, 2, True), (1080, 1920, 5, 8, 2, False),
                                                               (540, 960, 5, 4, 3, False), (270, 480, 4, 3, 2, False),
                                                               (300, 500, 3, 2, 1, False)])
def test_row_sharded_batch_matches_unsharded(rows, cols, levels, world, batch, shared):
    """RowShardBatch (one band launch per level for all pairs, batched halo rows, pyramids restricted to
    the rows a band touches) as virtual shards on one GPU: bit-identical to the unsharded batch and to
    the oracle.  Foreign flow rows are poisoned after every level; with shared=False every rank builds
    its own pyramids into NaN-prefilled buffers (only the rows of `prev` its band touches)."""
    import torch
    from introtocomputervision_amd import lk
    from introtocomputervision_amd._capi import Context
    ctx = Context(0)
    pairs = [synth.lk_pair(600 + i, rows, cols, 3, -2) for i in range(batch)]
    prev = torch.from_numpy(np.stack([p for p, _ in pairs])).cuda()
    nxt = torch.from_numpy(np.stack([n for _, n in pairs])).cuda()
    gu, gv = lk.calcOpticalFlowPyrBatch(prev, nxt, 15, levels, ctx=ctx)
    runners = [shard.RowShardBatch(ctx, rows, cols, levels, 15, batch, g, world) for g in range(world)]
    if not shared:
        for r in runners:
            for t in r.ppyr[1:] + r.npyr[1:]:
                t.fill_(float("nan"))
    u = torch.full_like(prev, float("nan"))
    v = torch.full_like(prev, float("nan"))
    for _ in range(2):
        shard.run_virtual_batch(runners, prev, nxt, u, v, poison=float("nan"), shared_pyramids=shared)
    torch.cuda.synchronize()
    assert torch.equal(u, gu) and torch.equal(v, gv)
    eu, ev = orc.lk_flow_pyr(pairs[0][0], pairs[0][1], 15, levels)
    assert np.array_equal(u[0].cpu().numpy(), eu) and np.array_equal(v[0].cpu().numpy(), ev)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,margin,expect_ok", [("textured", 64, True), ("textured", 0, False), ("smooth", 24, False)])
def test_declared_next_margin_with_fallback(kind, margin, expect_ok):
    """SURVEY.md section 8e's "declared bound on max |dv|": ranks build `next` only for band + margin rows (the
    rest of their pyramid buffers is NaN here), a device-side check watches the coarse flow, and a raised flag
    makes every rank repeat the step on the whole frame.  Exact either way.  `textured`: white-noise frames,
    every window well conditioned, flow (3, -2) everywhere -> a 64-row margin suffices (the largest |dv| is 29), 0 rows do not;
    `smooth`: the bench's frames, whose nearly degenerate windows throw isolated flows of hundreds of pixels
    -> the fallback runs."""
    import torch
    from introtocomputervision_amd import lk
    from introtocomputervision_amd._capi import Context
    rows, cols, levels, world, batch = 540, 960, 5, 4, 2
    ctx = Context(0)
    if kind == "textured":
        rng = np.random.default_rng(3)
        pairs = []
        for i in range(batch):
            big = (rng.random((rows + 8, cols + 8)) * 255).astype(np.float32)
            k = np.array([1, 2, 1], np.float32) / 4  # a little smoothing so that sub-sampled levels keep texture
            big = np.apply_along_axis(lambda m: np.convolve(m, k, mode="same"), 0, big)
            big = np.apply_along_axis(lambda m: np.convolve(m, k, mode="same"), 1, big).astype(np.float32)
            pairs.append((np.ascontiguousarray(big[4:4 + rows, 4:4 + cols]), np.ascontiguousarray(big[2:2 + rows, 7:7 + cols])))
    else:
        pairs = [synth.lk_pair(0x5EED0005 + i, rows, cols, 3, -2) for i in range(batch)]
    prev = torch.from_numpy(np.stack([p for p, _ in pairs])).cuda()
    nxt = torch.from_numpy(np.stack([n for _, n in pairs])).cuda()
    gu, gv = lk.calcOpticalFlowPyrBatch(prev, nxt, 15, levels, ctx=ctx)
    runners = [shard.RowShardBatch(ctx, rows, cols, levels, 15, batch, g, world, next_margin=margin) for g in range(world)]
    for r in runners:
        for t in r.ppyr[1:] + r.npyr[1:]:
            t.fill_(float("nan"))
    u = torch.full_like(prev, float("nan"))
    v = torch.full_like(prev, float("nan"))
    ok = shard.run_virtual_batch_checked(runners, prev, nxt, u, v, poison=float("nan"))
    torch.cuda.synchronize()
    assert ok == expect_ok, (kind, margin, float(gv.abs().max()))
    assert torch.equal(u, gu) and torch.equal(v, gv)
    if expect_ok:  # the cheap path really skipped rows: some level's `next` buffer still holds NaN rows
        assert any(bool(torch.isnan(t).any()) for t in runners[1].npyr[1:])


@pytest.mark.gpu
@pytest.mark.parametrize("kind,margin,expect_ok", [("textured", 64, True), ("smooth", 24, False)])
def test_declared_margin_on_a_side_stream(kind, margin, expect_ok):
    """run_checked / run_virtual_batch_checked given a NON-current, non-blocking stream.  The flag fill,
    the bound checks, the row copies and the flag read must all follow the launch stream; a busy kernel queued on that
    stream first makes any work left on torch's current stream overtake it (stale flag -> missed violation, or rows
    copied before they were computed)."""
    import torch
    from introtocomputervision_amd import lk
    from introtocomputervision_amd._capi import Context
    rows, cols, levels, world, batch = 540, 960, 5, 4, 2
    ctx = Context(0)
    if kind == "textured":
        rng = np.random.default_rng(3)
        pairs = []
        for i in range(batch):
            big = (rng.random((rows + 8, cols + 8)) * 255).astype(np.float32)
            k = np.array([1, 2, 1], np.float32) / 4
            big = np.apply_along_axis(lambda m: np.convolve(m, k, mode="same"), 0, big)
            big = np.apply_along_axis(lambda m: np.convolve(m, k, mode="same"), 1, big).astype(np.float32)
            pairs.append((np.ascontiguousarray(big[4:4 + rows, 4:4 + cols]), np.ascontiguousarray(big[2:2 + rows, 7:7 + cols])))
    else:
        pairs = [synth.lk_pair(0x5EED0005 + i, rows, cols, 3, -2) for i in range(batch)]
    prev = torch.from_numpy(np.stack([p for p, _ in pairs])).cuda()
    nxt = torch.from_numpy(np.stack([n for _, n in pairs])).cuda()
    gu, gv = lk.calcOpticalFlowPyrBatch(prev, nxt, 15, levels, ctx=ctx)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()  # non-blocking with respect to the null stream, and not current
    runners = [shard.RowShardBatch(ctx, rows, cols, levels, 15, batch, g, world, next_margin=margin) for g in range(world)]
    busy = torch.empty(1 << 26, device="cuda")
    for rep in range(3):
        u = torch.full_like(prev, float("nan"))
        v = torch.full_like(prev, float("nan"))
        torch.cuda.synchronize()
        with torch.cuda.stream(side):  # ~ms of work ahead of the step on the side stream
            for _ in range(20):
                busy.mul_(1.0001)
        ok = shard.run_virtual_batch_checked(runners, prev, nxt, u, v, stream=side.cuda_stream, poison=float("nan"))
        side.synchronize()
        assert ok == expect_ok
        assert torch.equal(u, gu) and torch.equal(v, gv)
    # the single-rank distributed form takes the same path (no process group: the exchange list is empty)
    r1 = shard.RowShardBatch(ctx, rows, cols, levels, 15, batch, 0, 1, next_margin=margin)
    u = torch.full_like(prev, float("nan"))
    v = torch.full_like(prev, float("nan"))
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(20):
            busy.mul_(1.0001)
    r1.run_checked(prev, nxt, u, v, side.cuda_stream)
    side.synchronize()
    assert torch.equal(u, gu) and torch.equal(v, gv)
