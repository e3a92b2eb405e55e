"""HIP path (through the C ABI) against the COMMITTED golden vectors -- no oracle call at all."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
HERE = os.path.dirname(os.path.abspath(__file__))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(HERE, "golden", "golden_v1.npz"))


def test_lk_path_against_golden(G):
    from introtocomputervision_amd import harris, lk, pyr
    img = G["img"]
    for k, scale, a, b in ((3, 1.0, "sobel3_x", "sobel3_y"), (3, np.float32(1 / 9.0), "sobel3s_x", "sobel3s_y"),
                           (5, 1.0, "sobel5_x", "sobel5_y")):
        gx, gy = harris.getGradients(dev(img), k, scale)
        assert np.array_equal(host(gx), G[a]) and np.array_equal(host(gy), G[b])
    assert np.array_equal(host(pyr.pyrDown(dev(img))), G["pyr_down"])
    assert np.array_equal(host(pyr.pyrUp(dev(img[:20, :28]))), G["pyr_up"])
    assert np.array_equal(host(pyr.resizeLinear(dev(img[:20, :30]), 21, 31)), G["resize_21x31"])
    assert np.array_equal(host(lk.warp(dev(img), dev(G["warp_du"]), dev(G["warp_dv"]))), G["warp"])
    u, v = lk.calcOpticalFlow(dev(G["lk_prev"]), dev(G["lk_next"]), 15)
    assert np.array_equal(host(u), G["lk_u15"]) and np.array_equal(host(v), G["lk_v15"])
    u, v = lk.calcOpticalFlowPyr(dev(G["lk_prev"]), dev(G["lk_next"]), 15, 3)
    assert np.array_equal(host(u), G["lkpyr_u"]) and np.array_equal(host(v), G["lkpyr_v"])
    u, v = lk.calcOpticalFlowPyr(dev(G["lk2_prev"]), dev(G["lk2_next"]), 7, 3)
    assert np.array_equal(host(u), G["lkpyr2_u"]) and np.array_equal(host(v), G["lkpyr2_v"])


def test_harris_stereo_hough_against_golden(G):
    from introtocomputervision_amd import harris, hough, stereo
    gx, gy = harris.getGradients(dev(G["chk"]), 3)
    R = harris.getCornerResponse(gx, gy, 5, 1.5, 0.04)
    assert np.array_equal(host(R), G["harris_R"])
    c, l = harris.refineCorners(R, 5e8, 5)
    assert np.array_equal(host(c), G["harris_corners"]) and np.array_equal(host(l), G["harris_locs"])
    kp = host(harris.getKeypoints(gx, gy, l, 10))
    assert np.array_equal(kp[:, :3], G["sift_kp"][:, :3]) and np.allclose(kp[:, 3], G["sift_kp"][:, 3], atol=1e-3, rtol=0)
    L, Rr = dev(G["st_left"]), dev(G["st_right"])
    assert np.array_equal(host(stereo.disparitySSD(L, Rr, 3, -24, 0)), G["ssd_r3"])
    assert np.array_equal(host(stereo.disparitySSD(L, Rr, 3, -24, 0, stereo.AS_WRITTEN_CUDA)), G["ssd_r3_as_written"])
    assert np.array_equal(host(stereo.disparitySSD(L, Rr, 3, -24, 0, stereo.STEREO_SERIAL)), G["ssd_r3_serial"])
    assert np.array_equal(host(stereo.disparityNCorr(L + 1, Rr + 1, 3, -24, 0)), G["ncc_r3"])
    m = dev(G["hough_mask"])
    acc = hough.houghLinesAccumulate(m, 1, 1)
    assert np.array_equal(host(acc), G["hough_lines"])
    assert np.array_equal(host(hough.houghLinesAccumulate(m, 2, 3)), G["hough_lines_b23"])
    assert np.array_equal(host(hough.findLocalMaxima(acc, 8, 30)).astype(np.uint32), G["hough_peaks"])
    assert np.array_equal(host(hough.houghCirclesAccumulate(m, 12)), G["hough_circles_r12"])


def test_contract_corners_against_golden(G):
    """r03 entries: cvRound's INT_MIN for non-finite / far map entries, the rolling column sums."""
    from introtocomputervision_amd import lk, stereo
    w = host(lk.warp(dev(G["img"]), dev(G["warp_nonfinite_du"]), dev(G["warp_nonfinite_dv"])))
    assert w.tobytes() == G["warp_nonfinite"].tobytes()
    for y, x in ((3, 4), (5, 6), (7, 8), (9, 10), (11, 12)):
        assert G["warp_nonfinite"][y, x] == 0.0
    L, Rr = dev(G["st_left_f"]), dev(G["st_right"])
    assert np.array_equal(host(stereo.disparitySSD(L, Rr, 3, -24, 0, stereo.AS_WRITTEN_CUDA_ROLLING)), G["ssd_r3_rolling"])
    assert np.array_equal(host(stereo.disparityNCorr(L + 1, Rr + 1, 3, -24, 0, 1 | 8)), G["ncc_r3_rolling"])
