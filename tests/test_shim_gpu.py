"""The header-only C++ shim (the reference's namespaces and signatures on top of the C ABI):
compile tests/cpp/shim_demo.cpp with g++, run it the way a psN driver would call its library,
and compare every output with the CPU oracle (bit-exact; keypoint angles within 1e-3 deg)."""
import os
import subprocess

import numpy as np
import pytest

import _oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_demo(tmp):
    exe = os.path.join(tmp, "shim_demo")
    lib = os.path.join(ROOT, "introtocomputervision_amd")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cpp", "shim_demo.cpp"),
                    "-o", exe, "-L" + lib, "-lmicv", "-Wl,-rpath," + lib], check=True)
    return exe


def test_shim_compiles_on_cpu(tmp_path):
    build_demo(str(tmp_path))


@pytest.mark.gpu
def test_shim_matches_oracle(tmp_path):
    from introtocomputervision_amd import synth
    d = str(tmp_path)
    exe = build_demo(d)
    rows, cols = 96, 160
    prev, nxt = synth.lk_pair(0x5EED0005, rows, cols, 3, -2)
    chk = synth.checkerboard(rows, cols, square=20, seed=0x5EED0001)
    left, right, _ = synth.stereo_pair(0x5EED0002, rows, cols)
    mask, lines, _ = synth.hough_mask(rows, cols, n_lines=4, radii=())
    import test_canny as tc
    import test_match as tm
    img8 = tc.scene(rows, cols, seed=3)
    img8b = np.ascontiguousarray(np.roll(img8, 5, axis=1))
    hist = np.random.default_rng(2).integers(0, 256, (rows, cols)).astype(np.uint8)
    desc1, desc2 = tm.descriptors(60, 75, 128, 9)
    desc2[5] = desc2[3]
    desc1[0] = desc2[3]
    # colour frames as cv::imread would hand them to the unchanged ps5 driver
    rng = np.random.default_rng(11)
    def colourise(g):  # three different, clipped mixtures of the grey texture: R, G, B really differ
        c = np.stack([g * 0.9 + 10, g * 0.7 + 40, 255 - g * 0.8], axis=-1)
        return np.clip(c + rng.integers(-3, 4, c.shape), 0, 255)
    prev_rgb, next_rgb = colourise(prev).astype(np.uint8), colourise(nxt).astype(np.uint8)
    prev_rgbf, next_rgbf = (colourise(prev) / 3.0).astype(np.float32), (colourise(nxt) / 3.0).astype(np.float32)
    prev_rgba = np.concatenate([prev_rgb, rng.integers(0, 256, (rows, cols, 1)).astype(np.uint8)], axis=-1)
    prev_g8, next_g8 = prev.astype(np.uint8), nxt.astype(np.uint8)
    for name, a in (("prev_rgb.u8", prev_rgb), ("next_rgb.u8", next_rgb), ("prev_rgbf.f32", prev_rgbf),
                    ("next_rgbf.f32", next_rgbf), ("prev_rgba.u8", prev_rgba), ("prev_g8.u8", prev_g8),
                    ("next_g8.u8", next_g8)):
        np.ascontiguousarray(a).tofile(os.path.join(d, name))
    for name, a in (("prev.f32", prev), ("next.f32", nxt), ("chk.f32", chk), ("left.f32", left),
                    ("right.f32", right), ("mask.u8", mask), ("img8.u8", img8), ("img8b.u8", img8b),
                    ("hist.u8", hist), ("desc1.f32", desc1), ("desc2.f32", desc2)):
        a.tofile(os.path.join(d, name))
    r = subprocess.run([exe, d, str(rows), str(cols)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr

    def rd(name, dtype, shape=None):
        a = np.fromfile(os.path.join(d, name), dtype=dtype)
        return a.reshape(shape) if shape else a

    eu, ev = orc.lk_flow_pyr(prev, nxt, 15, 4)
    assert np.array_equal(rd("lkpyr_u.f32", np.float32, (rows, cols)), eu)
    assert np.array_equal(rd("lkpyr_v.f32", np.float32, (rows, cols)), ev)
    assert np.array_equal(rd("lk_u.f32", np.float32, (rows, cols)), orc.lk_flow(prev, nxt, 15)[0])
    assert np.array_equal(rd("warped.f32", np.float32, (rows, cols)), orc.lk_warp(nxt, eu, ev))
    p3 = orc.gaussian_pyramid(prev, 4)[3]
    assert np.array_equal(rd("pyr3.f32", np.float32, p3.shape), p3)
    assert np.array_equal(rd("pyr3_up.f32", np.float32, (2 * p3.shape[0], 2 * p3.shape[1])), orc.pyr_up(p3))
    # the colour branch: Pyramids.cpp:9-15 -> OpticalFlow.cpp:122-167, bit for bit
    pg, ng = orc.to_gray(prev_rgb), orc.to_gray(next_rgb)
    assert not np.array_equal(pg, prev) and pg.std() > 5  # a real conversion, not a pass-through
    assert np.array_equal(pg, orc.rgb8_to_gray(prev_rgb))
    ecu, ecv = orc.lk_flow_pyr(pg, ng, 15, 4)
    assert np.array_equal(rd("lkpyr_rgb_u.f32", np.float32, (rows, cols)), ecu)
    assert np.array_equal(rd("lkpyr_rgb_v.f32", np.float32, (rows, cols)), ecv)
    assert np.array_equal(rd("pyr_rgb0.f32", np.float32, (rows, cols)), pg)
    pg2 = orc.gaussian_pyramid(pg, 3)[2]
    assert np.array_equal(rd("pyr_rgb2.f32", np.float32, pg2.shape), pg2)
    assert np.array_equal(rd("lk_rgb_u.f32", np.float32, (rows, cols)), orc.lk_flow(pg, ng, 15)[0])
    efu, _ = orc.lk_flow_pyr(orc.to_gray(prev_rgbf), orc.to_gray(next_rgbf), 15, 4)
    assert np.array_equal(rd("lkpyr_rgbf_u.f32", np.float32, (rows, cols)), efu)
    assert np.array_equal(rd("pyr_rgba0.f32", np.float32, (rows, cols)), pg)  # alpha ignored
    e8u, _ = orc.lk_flow_pyr(prev_g8.astype(np.float32), next_g8.astype(np.float32), 15, 4)
    assert np.array_equal(rd("lkpyr_g8_u.f32", np.float32, (rows, cols)), e8u)
    gx, gy = orc.sobel(chk, 3, 1.0)
    R = orc.harris_response(gx, gy, 5, 1.5, 0.04)
    assert np.array_equal(rd("harris_R.f32", np.float32, (rows, cols)), R)
    _, locs = orc.harris_refine(R, 5e8, 5)
    assert len(locs) > 5 and np.array_equal(rd("harris_locs.i32", np.int32).reshape(-1, 2), locs)
    Rc = orc.harris_response_ex(gx, gy, 5, 1.5, 0.04, orc.HARRIS_CPU)  # harris::cpu:: keeps its own arithmetic
    assert rd("harris_cpu_R.f32", np.float32, (rows, cols)).tobytes() == Rc.tobytes() and not np.array_equal(Rc, R)
    _, locs_c = orc.harris_refine(Rc, 5e8, 5)
    assert np.array_equal(rd("harris_cpu_locs.i32", np.int32).reshape(-1, 2), np.vstack([[[-1, -1]], locs_c]))  # appended (Harris.cpp:138)
    kp = rd("kps.f32", np.float32).reshape(-1, 4)
    ekp = orc.sift_keypoints(gx, gy, locs, 10)
    assert np.array_equal(kp[:, :3], ekp[:, :3]) and np.allclose(kp[:, 3], ekp[:, 3], atol=1e-3, rtol=0)
    edesc = orc.sift_descriptors(gx, gy, ekp)
    # the shim hands the kernel the keypoint angles IT computed (device atan2f), the oracle its own:
    # compare on the shim's keypoints so the descriptor stage itself is bit for bit
    assert np.array_equal(rd("sift_desc.f32", np.float32, (len(kp), 128)), orc.sift_descriptors(gx, gy, kp))
    assert np.abs(rd("sift_desc.f32", np.float32, (len(kp), 128)) - edesc).max() <= 1
    sm = rd("sift_selfmatch.i32", np.int32).reshape(-1, 2)
    assert len(sm) > 0 and np.array_equal(sm[:, 0], sm[:, 1])  # every descriptor's nearest neighbour is itself
    assert np.array_equal(rd("disp_cuda.i8", np.int8, (rows, cols)), orc.disparity_ssd(left, right, 5, -30, 0, 3))
    assert np.array_equal(rd("disp_serial.i8", np.int8, (rows, cols)), orc.disparity_ssd_serial(left, right, 5, -30, 0))
    acc = orc.hough_lines(mask, 1, 1)
    assert np.array_equal(rd("acc.i32", np.int32, acc.shape), acc)
    assert np.array_equal(rd("peaks.u32", np.uint32).reshape(-1, 2), orc.hough_peaks(acc, 10, 40))
    # GpuMat overloads: same accumulators / peaks from device-resident inputs
    acc2 = orc.hough_lines(mask, 2, 3)
    assert np.array_equal(rd("acc_gpumat.i32", np.int32, acc2.shape), acc2)
    pk2 = rd("peaks_gpumat.u32", np.uint32).reshape(-1, 2)
    assert np.array_equal(pk2[0], [7, 7]) and np.array_equal(pk2[1:], orc.hough_peaks(acc2, 6, 20))
    assert np.array_equal(rd("circ_gpumat.i32", np.int32, (rows, cols)), orc.hough_circles(mask, 12))
    # next rows through the shim's host-pointer path
    assert np.array_equal(rd("edges.u8", np.uint8, (rows, cols)), tc.oracle_edges(img8, 5, 1.4, 30, 90))
    ediff = orc.mhi_frame_difference(img8, img8b, 20, 5, 1.5)
    assert np.array_equal(rd("mhi_diff.u8", np.uint8, (rows, cols)), ediff)
    ehist = orc.mhi_update(hist, ediff, 25)
    assert np.array_equal(rd("mhi_hist.u8", np.uint8, (rows, cols)), ehist)
    assert np.array_equal(rd("mhi_diff73.u8", np.uint8, (rows, cols)), orc.mhi_frame_difference(img8, img8b, 10, (7, 3), 2.0))
    assert np.array_equal(rd("mhi_diffdef.u8", np.uint8, (rows, cols)), orc.mhi_frame_difference(img8, img8b, 10, (3, 3), 1.0))
    assert np.array_equal(rd("mhi_mei.u8", np.uint8, (rows, cols)), orc.mhi_energy(ehist))
    assert np.array_equal(rd("mhi_mei1.u8", np.uint8, (rows, cols)), orc.mhi_energy(ediff))
    eidx, edist = tm.oracle_knn2(desc1, desc2)
    em, ed = tm.oracle_ratio(eidx, edist, 0.75)
    assert len(em) > 0 and np.array_equal(rd("good.i32", np.int32).reshape(-1, 2), em)
    assert np.array_equal(rd("good_dist.f32", np.float32), ed)
    # the reference's kernel-timing log lines, through micv_shim::log_kernel_times_to
    import re
    log = open(os.path.join(d, "kernel_log.txt")).read().splitlines()
    pat = re.compile(r"^(\w+Kernel) (execution )?took ([0-9.]+) ms$")
    seen = {}
    for line in log:
        m = pat.match(line)
        assert m, line
        assert (m.group(2) is None) == m.group(1).startswith("pyr"), line  # Pyramids.cu:69,123 say "took"
        assert 0.0 < float(m.group(3)) < 1000.0, line
        seen[m.group(1)] = seen.get(m.group(1), 0) + 1
    for k in ("cornerResponseKernel", "refineCornersKernel", "disparitySSDKernel", "houghLinesAccumulateKernel",
              "findLocalMaximaKernel", "pyrUpsampleKernel"):
        assert seen.get(k, 0) >= 1, (k, seen)


@pytest.mark.gpu
def test_kernel_log_callback_python():
    """micv_set_kernel_log through the ctypes mirror: names and plausible times for every timed `_host`
    call, nothing after the sink is removed."""
    from introtocomputervision_amd import _capi, harris, hough, pyr, stereo, synth
    got = []
    _capi.set_kernel_log(lambda name, ms: got.append((name, ms)))
    try:
        img = synth.checkerboard(120, 160, 20, seed=1)
        gx, gy = harris.getGradients(img, 3)
        R = harris.getCornerResponse(gx, gy, 5, 1.5, 0.04)
        harris.refineCorners(R, 5e8, 5)
        l, r, _ = synth.stereo_pair(1, 60, 90)
        stereo.disparitySSD(l, r, 3, -10, 0)
        stereo.disparityNCorr(l + 1, r + 1, 3, -10, 0)
        m = synth.hough_mask(90, 130, n_lines=3, radii=(12,))[0]
        acc = hough.houghLinesAccumulate(m, 1, 1)
        hough.houghCirclesAccumulate(m, 12)
        hough.findLocalMaxima(acc, 5, 20)
        pyr.pyrDown(img)
        pyr.pyrUp(img)
    finally:
        _capi.set_kernel_log(None)
    assert [n for n, _ in got] == ["cornerResponseKernel", "refineCornersKernel", "disparitySSDKernel",
                                   "disparityNCorrKernel", "houghLinesAccumulateKernel", "houghCirclesAccumulateKernel",
                                   "findLocalMaximaKernel", "pyrDownsampleKernel", "pyrUpsampleKernel"]
    assert all(0.0 < ms < 1000.0 for _, ms in got)
    n = len(got)
    pyr.pyrDown(img)
    assert len(got) == n


def build_seq_demo(tmp):
    exe = os.path.join(tmp, "seq_demo")
    lib = os.path.join(ROOT, "introtocomputervision_amd")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cpp", "seq_demo.cpp"),
                    "-o", exe, "-L" + lib, "-lmicv", "-Wl,-rpath," + lib], check=True)
    return exe


def test_sequence_demo_compiles_on_cpu(tmp_path):
    build_seq_demo(str(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("n,rows,cols,cn,f32", [(2, 40, 64, 1, 1), (3, 96, 160, 3, 0), (7, 120, 200, 1, 0), (9, 67, 131, 4, 1),
                                                 (5, 540, 960, 3, 0)])
def test_frame_sequence_entry_from_cpp(tmp_path, n, rows, cols, cn, f32):
    """micv_lk_flow_seq_host (every frame uploaded once; upload, chain and download of consecutive pairs overlapped) and
    the shim's lk::calcOpticalFlowPyrSequence, called from C++ through the C ABI: byte-identical to n - 1 per-pair
    micv_lk_flow_pyr_frames_host calls -- grey / colour, 8-bit / f32 frames, 2 to 9 frames (the rings hold three)."""
    exe = build_seq_demo(str(tmp_path))
    out = subprocess.run([exe, str(n), str(rows), str(cols), str(cn), str(f32)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
