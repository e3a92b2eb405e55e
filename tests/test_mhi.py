"""ps7 motion-history path: oracle sanity on CPU, HIP parity on GPU
(byte outputs: bit-exact)."""
import numpy as np
import pytest

import _oracle as orc


def moving_square(rows, cols, x0, seed=3):
    rng = np.random.default_rng(seed)
    f = rng.integers(90, 110, (rows, cols)).astype(np.uint8)
    f[rows // 3: rows // 3 + 30, x0:x0 + 30] = 230
    return f


def test_oracle_frame_difference_finds_the_leading_edge():
    f1, f2 = moving_square(120, 160, 40), moving_square(120, 160, 52)
    d = orc.mhi_frame_difference(f1, f2, 20, 5, 1.5)
    assert set(np.unique(d)) <= {0, 1}
    ys, xs = np.nonzero(d)
    assert len(xs) > 100 and xs.min() >= 64 and xs.max() <= 88 and ys.min() >= 35 and ys.max() <= 75
    # saturating subtract: the trailing edge (f2 < f1) produces nothing
    assert not d[:, :60].any()
    h = np.zeros((120, 160), np.uint8)
    for _ in range(3):
        h = orc.mhi_update(h, d, 25)
    assert set(np.unique(h)) == {0, 25}
    h2 = orc.mhi_update(h, np.zeros_like(d), 25)
    assert set(np.unique(h2)) == {0, 24}
    assert set(np.unique(orc.mhi_energy(h2))) == {0, 1} and np.array_equal(orc.mhi_energy(h2) > 0, h2 > 0)
    # cv::Size(width, height): a non-square blur differs from its transpose on this input
    assert not np.array_equal(orc.mhi_frame_difference(f1, f2, 20, (9, 1), 2.0),
                              orc.mhi_frame_difference(f1, f2, 20, (1, 9), 2.0))
    assert np.array_equal(orc.mhi_threshold(np.arange(256, dtype=np.uint8).reshape(16, 16), 1.7).ravel(),
                          (np.arange(256) >= 2).astype(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,ksize,sigma,thr", [(120, 160, 5, 1.5, 20), (97, 131, 31, 10.0, 1.7),
                                                       (33, 40, 3, 1.0, 5), (480, 640, 31, 10.0, 1.7),
                                                       (90, 140, (7, 3), 2.0, 10), (64, 200, (1, 9), 1.2, 4)])
def test_mhi_gpu_matches_oracle(rows, cols, ksize, sigma, thr):
    import torch
    from introtocomputervision_amd import mhi
    f1, f2 = moving_square(rows, cols, cols // 4), moving_square(rows, cols, cols // 4 + 7, seed=4)
    exp = orc.mhi_frame_difference(f1, f2, thr, ksize, sigma)
    d1, d2 = torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()
    got = mhi.frameDifference(d1, d2, thr, ksize, sigma)
    assert np.array_equal(got.cpu().numpy(), exp)
    hist = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (rows, cols)).astype(np.uint8)).cuda()
    eh = orc.mhi_update(hist.cpu().numpy(), exp, 25)
    mhi.calcMotionHistory(hist, got, 25)
    assert np.array_equal(hist.cpu().numpy(), eh)
    assert np.array_equal(mhi.thresholdDifference(d1, thr).cpu().numpy(), orc.mhi_threshold(f1, thr))
    assert np.array_equal(mhi.energyFromHistory(hist).cpu().numpy(), orc.mhi_energy(eh))
    assert np.array_equal(orc.mhi_energy(eh), (eh > 0).astype(np.uint8))
    meis = mhi.energyFromHistory([eh, exp])  # the vector overload (MotionHistory.cpp:107-112), host flavour
    assert np.array_equal(meis[0], orc.mhi_energy(eh)) and np.array_equal(meis[1], orc.mhi_energy(exp))
    # host-pointer flavours (numpy in, numpy out)
    assert np.array_equal(mhi.frameDifference(f1, f2, thr, ksize, sigma), exp)
    assert np.array_equal(mhi.thresholdDifference(f1, thr), orc.mhi_threshold(f1, thr))
    hh = np.random.default_rng(1).integers(0, 256, (rows, cols)).astype(np.uint8)
    mhi.calcMotionHistory(hh, exp, 25)
    assert np.array_equal(hh, eh)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols", [(1080, 1920), (53, 64), (52, 70), (1, 300), (300, 1), (5, 5), (7, 129), (105, 58), (200, 6), (64, 1000)])
@pytest.mark.parametrize("thr", [1, 0, 40])
def test_mhi_bit_plane_open_on_noise_and_edges(rows, cols, thr):
    """r05: frameDifference = a ballot-mask launch + a bit-plane 7x7 open (mhi.hip).  Random frames make masks that are
    dense, ragged and alive right at the borders -- where the reflected padding of BOTH morphology passes matters -- on
    widths that are not multiples of 64, images smaller than the structuring element, one-row / one-column images,
    a threshold of 0 (everything set) and pitched device views.  Byte-exact against the oracle."""
    import torch
    from introtocomputervision_amd import mhi
    rng = np.random.default_rng(rows * 7919 + cols + thr)
    f1 = rng.integers(0, 256, (rows, cols)).astype(np.uint8)
    f2 = np.clip(f1.astype(np.int32) + rng.integers(-60, 90, (rows, cols)) * (rng.random((rows, cols)) < 0.6), 0, 255).astype(np.uint8)
    for ksize, sigma in ((3, 1.0), ((5, 1), 1.5)):
        exp = orc.mhi_frame_difference(f1, f2, thr, ksize, sigma)
        got = mhi.frameDifference(torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda(), thr, ksize, sigma)
        assert np.array_equal(got.cpu().numpy(), exp), (ksize, int((got.cpu().numpy() != exp).sum()))
    big1 = torch.zeros((rows + 4, cols + 9), dtype=torch.uint8, device="cuda"); big2 = torch.zeros_like(big1)
    v1, v2 = big1[2:2 + rows, 5:5 + cols], big2[2:2 + rows, 5:5 + cols]
    v1.copy_(torch.from_numpy(f1)); v2.copy_(torch.from_numpy(f2))
    exp = orc.mhi_frame_difference(f1, f2, thr, 3, 1.0)
    assert np.array_equal(mhi.frameDifference(v1, v2, thr, 3, 1.0).cpu().numpy(), exp)
    if 0 < thr < 40 and rows * cols > 2000:
        assert 0 < exp.sum() < exp.size  # the case exercises both values
