#!/usr/bin/env python3
"""Regenerates tests/golden/golden_v1.npz from the CPU oracle (oracle/liboracle.so).

The reference ships no golden vectors and cannot be built here, so these
fixtures pin THIS repository's arithmetic contract: small seeded inputs and the oracle's
outputs.  Any change to the oracle that alters a bit shows up as a diff of this file.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _oracle as orc  # noqa: E402
from introtocomputervision_amd import synth  # noqa: E402


def build():
    g = {}
    img = synth.smooth_noise(0xC0FFEE, 40, 56)
    g["img"] = img
    g["gauss15"] = orc.gaussian_kernel(15, 5.0)
    g["gauss5_s15"] = orc.gaussian_kernel(5, 1.5)
    g["sobel3_x"], g["sobel3_y"] = orc.sobel(img, 3, 1.0)
    g["sobel3s_x"], g["sobel3s_y"] = orc.sobel(img, 3, np.float32(1.0 / 9.0))
    g["sobel5_x"], g["sobel5_y"] = orc.sobel(img, 5, 1.0)
    g["pyr_down"] = orc.pyr_down(img)
    g["pyr_up"] = orc.pyr_up(img[:20, :28])
    g["resize_21x31"] = orc.resize_linear(img[:20, :30], 21, 31)
    rng = np.random.default_rng(12345)
    du = (rng.standard_normal(img.shape) * 2.5).astype(np.float32)
    dv = (rng.standard_normal(img.shape) * 2.5).astype(np.float32)
    g["warp_du"], g["warp_dv"] = du, dv
    g["warp"] = orc.lk_warp(img, du, dv)
    prev, nxt = synth.lk_pair(0x5EED0005, 72, 96, 3, -2)
    g["lk_prev"], g["lk_next"] = prev, nxt
    g["lk_u15"], g["lk_v15"] = orc.lk_flow(prev, nxt, 15)
    g["lkpyr_u"], g["lkpyr_v"] = orc.lk_flow_pyr(prev, nxt, 15, 3)
    p2, n2 = synth.lk_pair(77, 67, 120, 2, 1)  # odd rows: exercises the cv::resize branch
    g["lk2_prev"], g["lk2_next"] = p2, n2
    g["lkpyr2_u"], g["lkpyr2_v"] = orc.lk_flow_pyr(p2, n2, 7, 3)
    chk = synth.checkerboard(80, 120, square=20, seed=0x5EED0001)
    g["chk"] = chk
    gx, gy = orc.sobel(chk, 3, 1.0)
    R = orc.harris_response(gx, gy, 5, 1.5, 0.04)
    g["harris_R"] = R
    g["harris_corners"], g["harris_locs"] = orc.harris_refine(R, 5e8, 5)
    g["sift_kp"] = orc.sift_keypoints(gx, gy, g["harris_locs"], 10)
    left, right, negd = synth.stereo_pair(0x5EED0002, 36, 96)
    g["st_left"], g["st_right"] = left, right
    g["ssd_r3"] = orc.disparity_ssd(left, right, 3, -24, 0)
    g["ssd_r3_as_written"] = orc.disparity_ssd(left, right, 3, -24, 0, 3)
    g["ssd_r3_serial"] = orc.disparity_ssd_serial(left, right, 3, -24, 0)
    g["ncc_r3"] = orc.disparity_ncorr(left + 1, right + 1, 3, -24, 0)
    mask, lines, circles = synth.hough_mask(90, 130, n_lines=4, radii=(12,))
    g["hough_mask"] = mask
    acc = orc.hough_lines(mask, 1, 1)
    g["hough_lines"] = acc
    g["hough_lines_b23"] = orc.hough_lines(mask, 2, 3)
    g["hough_peaks"] = orc.hough_peaks(acc, 8, 30)
    g["hough_circles_r12"] = orc.hough_circles(mask, 12)
    # r03: the contract corners -- cvRound's INT_MIN for non-finite / far map entries (OpticalFlow.cpp:119)
    # and the CUDA stereo kernels' rolling column sums (DisparitySSD.cu:97-138) on a non-integer image
    du2, dv2 = du.copy(), dv.copy()
    du2[3, 4], du2[5, 6], dv2[7, 8], dv2[9, 10], du2[11, 12] = np.nan, np.inf, -np.inf, 3e9, -(2.0 ** 26)
    g["warp_nonfinite_du"], g["warp_nonfinite_dv"] = du2, dv2
    g["warp_nonfinite"] = orc.lk_warp(img, du2, dv2)
    lf = (left * np.float32(0.37) + np.float32(0.11)).astype(np.float32)
    lf[5, 17], lf[20, 60] = 1e4, 1e4
    g["st_left_f"] = lf
    g["ssd_r3_rolling"] = orc.disparity_ssd(lf, right, 3, -24, 0, 1 | 2 | 8)
    g["ncc_r3_rolling"] = orc.disparity_ncorr(lf + 1, right + 1, 3, -24, 0, 1 | 8)
    return g


def exercise_rest():
    """The oracle functions the golden file does not hold (they have their own tests): run once each, so
    that a sanitizer build sees every translation unit."""
    import ctypes as C
    rng = np.random.default_rng(3)
    img8 = rng.integers(0, 256, (40, 56), dtype=np.uint8)
    edge = orc._sig("orc_generate_edge", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_double,
                                                  C.c_double, C.c_double, C.c_void_p, C.c_size_t])
    out = np.empty_like(img8)
    assert edge(img8.ctypes.data, 40, 56, 56, 5, 1.5, 40.0, 120.0, out.ctypes.data, 56) == 0
    chk = synth.checkerboard(80, 120, square=20, seed=0x5EED0001)
    gx, gy = orc.sobel(chk, 3, 1.0)
    _, locs = orc.harris_refine(orc.harris_response(gx, gy, 5, 1.5, 0.04), 5e8, 5)
    kp = orc.sift_keypoints(gx, gy, locs, 10)
    desc = orc.sift_descriptors(gx, gy, kp)
    assert desc.shape == (len(kp), 128)
    knn = orc._sig("orc_bf_knn2", None, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t, C.c_int,
                                         C.c_void_p, C.c_void_p])
    idx = np.empty((len(desc), 2), np.int32); dist = np.empty((len(desc), 2), np.float32)
    d2 = np.ascontiguousarray(desc[::-1])
    knn(desc.ctypes.data, len(desc), 128, d2.ctypes.data, len(d2), 128, 128, idx.ctypes.data, dist.ctypes.data)
    f1 = rng.integers(0, 256, (30, 44), dtype=np.uint8); f2 = rng.integers(0, 256, (30, 44), dtype=np.uint8)
    m = orc.mhi_frame_difference(f1, f2, 20.0, 3, 1.0)
    orc.mhi_update(orc.mhi_energy(m), orc.mhi_threshold(f1, 100.0), 30)
    orc.to_gray(rng.integers(0, 256, (9, 11, 3), dtype=np.uint8))
    orc.to_gray(rng.random((9, 11, 4)).astype(np.float32))


if __name__ == "__main__":
    g = build()
    if "--check" in sys.argv:  # regenerate and compare with the committed file (used under sanitizers)
        old = np.load(os.path.join(HERE, "golden_v1.npz"))
        assert set(old.files) == set(g), sorted(set(old.files) ^ set(g))
        for k in old.files:
            a, b = old[k], g[k]
            assert a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes(), k
        exercise_rest()
        print("golden_v1.npz regenerates bit for bit;", len(g), "entries")
    else:
        np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **g)
        print("wrote golden_v1.npz:", {k: v.shape for k, v in g.items()})
