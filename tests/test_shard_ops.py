"""Row-sharded Hough / stereo / Harris: the distributed logic on CPU (oracle as the
compute function, gloo world_size 2) and the HIP band entry points on the GPU ("virtual shards":
every rank's call made in one process, collectives emulated by a sum / concatenation)."""
import os
import socket
import sys

import numpy as np
import pytest

import _oracle as orc
from introtocomputervision_amd import shard_ops as so
from introtocomputervision_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _mask(rows, cols, seed=5):
    rng = np.random.default_rng(seed)
    m = (rng.random((rows, cols)) < 0.02).astype(np.uint8) * 255
    m[rows // 3, :] = 255
    m[:, cols // 4] = 255
    return m


def _orc_lines_band(mask_rows, row0, rows, rho_bin, theta_bin):
    full = np.zeros((rows, mask_rows.shape[1]), np.uint8)
    full[row0:row0 + mask_rows.shape[0]] = mask_rows
    return orc.hough_lines(full, rho_bin, theta_bin)


def _orc_circles_band(mask_rows, row0, rows, radius):
    full = np.zeros((rows, mask_rows.shape[1]), np.uint8)
    full[row0:row0 + mask_rows.shape[0]] = mask_rows
    return orc.hough_circles(full, radius)


def test_row_cuts():
    assert so.row_cuts(1080, 8) == [0, 135, 270, 405, 540, 675, 810, 945, 1080]
    assert so.row_cuts(10, 3) == [0, 3, 6, 10]
    with pytest.raises(ValueError):
        so.row_cuts(3, 4)
    assert so.band_with_halo(100, 4, 0, 7) == ((0, 25), (0, 32))
    assert so.band_with_halo(100, 4, 3, 7) == ((75, 100), (68, 100))


@pytest.mark.parametrize("world", [2, 3, 5])
def test_hough_virtual_shards_oracle(world):
    rows, cols = 61, 83
    m = _mask(rows, cols)
    cuts = so.row_cuts(rows, world)
    lines = sum(so.hough_lines_sharded(m[cuts[g]:cuts[g + 1]], (cuts[g], cuts[g + 1]), rows, 2, 3,
                                       _orc_lines_band, so.LocalComm()) for g in range(world))
    assert np.array_equal(lines, orc.hough_lines(m, 2, 3))
    circ = sum(so.hough_circles_sharded(m[cuts[g]:cuts[g + 1]], (cuts[g], cuts[g + 1]), rows, 9,
                                        _orc_circles_band, so.LocalComm()) for g in range(world))
    assert np.array_equal(circ, orc.hough_circles(m, 9))


@pytest.mark.parametrize("world", [2, 4])
def test_stereo_virtual_shards_oracle(world):
    rows, cols, rad = 48, 96, 3
    left, right, _ = synth.stereo_pair(3, rows, cols)
    full = orc.disparity_ssd(left, right, rad, -15, 0)
    fulln = orc.disparity_ncorr(left, right, rad, -15, 0)
    for fn, want in ((lambda l, r: orc.disparity_ssd(l, r, rad, -15, 0), full),
                     (lambda l, r: orc.disparity_ncorr(l, r, rad, -15, 0), fulln)):
        parts = []
        for g in range(world):
            band, held = so.band_with_halo(rows, world, g, rad)
            parts.append(so.stereo_sharded(left[held[0]:held[1]], right[held[0]:held[1]], band, held, fn))
        assert np.array_equal(np.concatenate(parts), want)


def _harris_cfg():
    return dict(sobel_size=3, window_size=5, sigma=1.5, alpha=0.04, threshold=5e8, min_distance=5)


def _orc_harris_fns():
    return dict(grad_fn=lambda img, k: orc.sobel(img, k, 1.0),
                response_fn=orc.harris_response, refine_fn=orc.harris_refine)


@pytest.mark.parametrize("world", [2, 3])
def test_harris_virtual_shards_oracle(world):
    rows, cols = 120, 160
    img = synth.checkerboard(rows, cols, 20, seed=11)
    cfg = _harris_cfg()
    gx, gy = orc.sobel(img, 3, 1.0)
    resp = orc.harris_response(gx, gy, 5, 1.5, 0.04)
    corners, locs = orc.harris_refine(resp, 5e8, 5)
    assert len(locs) > 10
    halo = so.harris_halo(3, 5, 5)
    rs, cs, ls = [], [], []
    for g in range(world):
        band, held = so.band_with_halo(rows, world, g, halo)
        r, c, l = so.harris_sharded(img[held[0]:held[1]], band, held, comm=so.LocalComm(), **cfg,
                                    **_orc_harris_fns())
        rs.append(r), cs.append(c), ls.append(l)
    assert np.array_equal(np.concatenate(rs), resp)
    assert np.array_equal(np.concatenate(cs), corners)
    assert np.array_equal(np.concatenate(ls), locs)


def test_harris_halo_is_needed():
    rows, cols, world = 120, 160, 3
    img = synth.checkerboard(rows, cols, 20, seed=11)
    gx, gy = orc.sobel(img, 3, 1.0)
    resp = orc.harris_response(gx, gy, 5, 1.5, 0.04)
    band, held = so.band_with_halo(rows, world, 1, 1)  # too small
    r, _, _ = so.harris_sharded(img[held[0]:held[1]], band, held, comm=so.LocalComm(), **_harris_cfg(),
                                **_orc_harris_fns())
    assert not np.array_equal(r, resp[band[0]:band[1]])


def _gloo_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = so.TorchDist(rank, world)
    rows, cols = 61, 83
    m = _mask(rows, cols)
    cuts = so.row_cuts(rows, world)
    a, b = cuts[rank], cuts[rank + 1]
    t = lambda f: (lambda *args: torch.from_numpy(f(*[x.numpy() if hasattr(x, "numpy") else x for x in args])))
    lines = so.hough_lines_sharded(torch.from_numpy(m[a:b]), (a, b), rows, 1, 1, t(_orc_lines_band), comm)
    circ = so.hough_circles_sharded(torch.from_numpy(m[a:b]), (a, b), rows, 7, t(_orc_circles_band), comm)
    np.save(os.path.join(out_dir, f"lines{rank}.npy"), lines.numpy())
    np.save(os.path.join(out_dir, f"circ{rank}.npy"), circ.numpy())
    # Harris: every rank ends with the whole corner list
    img = synth.checkerboard(120, 160, 20, seed=11)
    band, held = so.band_with_halo(120, world, rank, so.harris_halo(3, 5, 5))

    def refine(resp, thr, d):
        c, l = orc.harris_refine(resp.numpy(), thr, d)
        return torch.from_numpy(c), torch.from_numpy(l)

    def grad(x, k):
        gx, gy = orc.sobel(x.numpy(), k, 1.0)
        return torch.from_numpy(gx), torch.from_numpy(gy)

    r, c, locs = so.harris_sharded(torch.from_numpy(img[held[0]:held[1]]), band, held, comm=comm,
                                   grad_fn=grad, response_fn=t(orc.harris_response), refine_fn=refine,
                                   **_harris_cfg())
    np.save(os.path.join(out_dir, f"locs{rank}.npy"), locs.numpy())
    np.save(os.path.join(out_dir, f"resp{rank}.npy"), r.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_over_gloo(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    mp.spawn(_gloo_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    m = _mask(61, 83)
    for r in range(world):  # the all-reduce leaves the full accumulator on every rank
        assert np.array_equal(np.load(tmp_path / f"lines{r}.npy"), orc.hough_lines(m, 1, 1))
        assert np.array_equal(np.load(tmp_path / f"circ{r}.npy"), orc.hough_circles(m, 7))
    img = synth.checkerboard(120, 160, 20, seed=11)
    gx, gy = orc.sobel(img, 3, 1.0)
    resp = orc.harris_response(gx, gy, 5, 1.5, 0.04)
    _, locs = orc.harris_refine(resp, 5e8, 5)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"locs{r}.npy"), locs)
    assert np.array_equal(np.concatenate([np.load(tmp_path / f"resp{r}.npy") for r in range(world)]), resp)


# ---- HIP path --------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_hough_band_entry_points_gpu(world):
    import torch

    from introtocomputervision_amd import hough
    from introtocomputervision_amd._capi import Context
    rows, cols = 270, 480
    m = synth.hough_mask(rows, cols)[0]
    ctx = Context(0)
    fns = so.gpu_fns(ctx)
    dm = torch.from_numpy(m).cuda()
    cuts = so.row_cuts(rows, world)
    lines = sum(so.hough_lines_sharded(dm[cuts[g]:cuts[g + 1]], (cuts[g], cuts[g + 1]), rows, 1, 1,
                                       fns.hough_lines_band, so.LocalComm()) for g in range(world))
    want = orc.hough_lines(m, 1, 1)
    assert np.array_equal(lines.cpu().numpy(), want)
    assert np.array_equal(hough.houghLinesAccumulate(dm, 1, 1, ctx=ctx).cpu().numpy(), want)
    circ = sum(so.hough_circles_sharded(dm[cuts[g]:cuts[g + 1]], (cuts[g], cuts[g + 1]), rows, 20,
                                        fns.hough_circles_band, so.LocalComm()) for g in range(world))
    assert np.array_equal(circ.cpu().numpy(), orc.hough_circles(m, 20))


@pytest.mark.gpu
def test_hough_band_rejects_bad_band():
    import torch

    from introtocomputervision_amd._capi import Context, MicvError
    fns = so.gpu_fns(Context(0))
    dm = torch.zeros((10, 32), dtype=torch.uint8, device="cuda")
    with pytest.raises(MicvError):
        fns.hough_lines_band(dm, 25, 30, 1, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_stereo_and_harris_shards_gpu(world):
    import torch

    from introtocomputervision_amd import harris, stereo
    from introtocomputervision_amd._capi import Context
    ctx = Context(0)
    fns = so.gpu_fns(ctx)
    rows, cols, rad = 128, 256, 5
    left, right, _ = synth.stereo_pair(3, rows, cols)
    dl, dr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    for fn, whole in ((fns.ssd(rad, -31, 0), stereo.disparitySSD(dl, dr, rad, -31, 0, ctx=ctx)),
                      (fns.ncorr(rad, -31, 0), stereo.disparityNCorr(dl, dr, rad, -31, 0, ctx=ctx))):
        parts = []
        for g in range(world):
            band, held = so.band_with_halo(rows, world, g, rad)
            parts.append(so.stereo_sharded(dl[held[0]:held[1]], dr[held[0]:held[1]], band, held, fn))
        assert torch.equal(torch.cat(parts), whole)
    assert np.array_equal(stereo.disparitySSD(dl, dr, rad, -31, 0, ctx=ctx).cpu().numpy(),
                          orc.disparity_ssd(left, right, rad, -31, 0))

    rows, cols = 480, 640
    img = synth.checkerboard(rows, cols, 40, seed=1)
    di = torch.from_numpy(img).cuda()
    gx, gy = harris.getGradients(di, 3, ctx=ctx)
    resp = harris.getCornerResponse(gx, gy, 5, 1.5, 0.04, ctx=ctx)
    corners, locs = harris.refineCorners(resp, 5e8, 5, ctx=ctx)
    assert locs.shape[0] > 50
    halo = so.harris_halo(3, 5, 5)
    rs, cs, ls = [], [], []
    for g in range(world):
        band, held = so.band_with_halo(rows, world, g, halo)
        r, c, l = so.harris_sharded(di[held[0]:held[1]], band, held, comm=so.LocalComm(),
                                    grad_fn=fns.grad, response_fn=fns.response, refine_fn=fns.refine,
                                    **_harris_cfg())
        rs.append(r), cs.append(c), ls.append(l)
    assert torch.equal(torch.cat(rs), resp)
    assert torch.equal(torch.cat(cs), corners)
    assert torch.equal(torch.cat(ls), locs)
