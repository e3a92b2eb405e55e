#!/usr/bin/env python3
"""Times the non-LK kernels of the path on their BASELINE configs (device-resident inputs,
HIP-event timing through micv_timer) and prints one JSON line each with the algorithmic GB/s of
DESIGN.md §5.  Usage on the GPU box: python tools/bench_kernels.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from introtocomputervision_amd import harris, hough, lk, pyr, stereo, synth
from introtocomputervision_amd._capi import Context, Timer

ctx = Context(0)
stream = torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = Timer()
    t.start(stream)
    for _ in range(iters):
        fn()
    t.stop(stream)
    return t.elapsed_ms() / iters


def line(name, ms, px, bytes_per_px, extra=None):
    d = {"kernel": name, "ms": round(ms, 4), "Mpix_per_s": round(px / ms / 1e3, 1),
         "algorithmic_GBps": round(px * bytes_per_px / ms / 1e6, 1),
         "frac_of_8TBps": round(px * bytes_per_px / ms / 1e6 / 8000, 4)}
    if extra:
        d.update(extra)
    print(json.dumps(d))


# C3: stereo SSD, 1080p, 11x11 window, 128 disparities
left, right, _ = synth.stereo_pair(0x5EED0002, 1080, 1920)
L, R = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
ms = timeit(lambda: stereo.disparitySSD(L, R, 5, -127, 0, ctx=ctx), iters=10)
line("stereo SSD 1080p r=5 d=128 (C3)", ms, 1080 * 1920, 9, {"Gpixdisp_per_s": round(1080 * 1920 * 128 / ms / 1e6, 1)})
ms = timeit(lambda: stereo.disparityNCorr(L, R, 5, -127, 0, ctx=ctx), iters=10)
line("stereo NCC 1080p r=5 d=128", ms, 1080 * 1920, 9)
# reference's own published case: 640x511, r=7, 96 disparities (19.3 ms on a GTX 1080)
l2, r2, _ = synth.stereo_pair(3, 511, 640)
L2, R2 = torch.from_numpy(l2).cuda(), torch.from_numpy(r2).cuda()
ms = timeit(lambda: stereo.disparitySSD(L2, R2, 7, -95, 0, stereo.AS_WRITTEN_CUDA, ctx=ctx), iters=10)
line("stereo SSD 640x511 r=7 d=96 as-written (ref: 19.28 ms GTX1080)", ms, 640 * 511, 9)

# C5: 4K Harris (Sobel -> response -> NMS + list) and C1 size
for rows, cols in ((2160, 3840), (480, 640)):
    img = torch.from_numpy(synth.checkerboard(rows, cols, 40, seed=0x5EED0004)).cuda()
    gx, gy = harris.getGradients(img, 3, ctx=ctx)
    ms_s = timeit(lambda: harris.getGradients(img, 3, ctx=ctx))
    ms_r = timeit(lambda: harris.getCornerResponse(gx, gy, 5, 1.5, 0.04, ctx=ctx))
    Rr = harris.getCornerResponse(gx, gy, 5, 1.5, 0.04, ctx=ctx)
    ms_n = timeit(lambda: harris.refineCorners(Rr, 5e8, 5, capacity=1 << 16, ctx=ctx))
    line(f"sobel3 pair {cols}x{rows}", ms_s, rows * cols, 12)
    line(f"harris response 5x5 {cols}x{rows} (ref 480x640: 0.80 ms GTX1080)", ms_r, rows * cols, 12)
    line(f"harris NMS + ordered list {cols}x{rows} (ref 480x640: 0.59 ms)", ms_n, rows * cols, 8)
    # r05: image -> R in ONE launch (Sobel inside the response kernel's tile), and the whole chain as one call
    ms_f = timeit(lambda: harris.cornersFromImage(img, 3, 5, 1.5, 0.04, 3e38, 5, capacity=1 << 16, ctx=ctx, want_gradients=False, lazy=True))
    line(f"harris image -> R -> (empty) list, one call, no gradient outputs {cols}x{rows}", ms_f, rows * cols, 12)
    ms_fg = timeit(lambda: harris.cornersFromImage(img, 3, 5, 1.5, 0.04, 5e8, 5, capacity=1 << 16, ctx=ctx, want_gradients=True, lazy=True))
    line(f"harris image -> gradients + R -> list, one call {cols}x{rows}", ms_fg, rows * cols, 20)
    ms_l = timeit(lambda: harris.refineCorners(Rr, 5e8, 5, capacity=1 << 16, ctx=ctx, lazy=True))
    line(f"harris NMS + ordered list {cols}x{rows}, count left on the device (lazy)", ms_l, rows * cols, 8)
    kp_ = harris.getKeypoints(gx, gy, harris.refineCorners(Rr, 5e8, 5, capacity=1 << 16, ctx=ctx)[1], 10, ctx=ctx)
    if len(kp_):
        ms_d = timeit(lambda: harris.computeDescriptors(gx, gy, kp_, ctx=ctx))
        line(f"SIFT-style descriptors, {len(kp_)} keypoints of size 10 (107x107 windows) {cols}x{rows}", ms_d, len(kp_) * 107 * 107,
             8, {"keypoints": int(len(kp_)), "us_per_keypoint": round(ms_d * 1e3 / len(kp_), 3)})

# Hough on a 1080p mask
mask, lines_, circles = synth.hough_mask(1080, 1920)
Mk = torch.from_numpy(mask).cuda()
n_edge = int((mask > 0).sum())
ms = timeit(lambda: hough.houghLinesAccumulate(Mk, 1, 1, ctx=ctx))
acc = hough.houghLinesAccumulate(Mk, 1, 1, ctx=ctx)
line("hough lines 1080p 1x1 bins", ms, 1080 * 1920, 1, {"edge_points": n_edge, "Mvotes_per_s": round(n_edge * 180 / ms / 1e3, 1)})
ms = timeit(lambda: hough.houghCirclesAccumulate(Mk, 30, ctx=ctx))
line("hough circles 1080p r=30", ms, 1080 * 1920, 5, {"Mvotes_per_s": round(n_edge * 360 / ms / 1e3, 1)})
ms = timeit(lambda: hough.findLocalMaxima(acc, 10, 300, ctx=ctx))
line("hough peaks top-10 of 4406x180", ms, 4406 * 180, 4)
ms = timeit(lambda: hough.findLocalMaxima(acc, 10, 300, ctx=ctx, lazy=True))
line("hough peaks top-10, count left on the device (lazy)", ms, 4406 * 180, 4)

# single-level LK and pyramid pieces at 1080p
prev, nxt = synth.lk_pair(0x5EED0005, 1080, 1920)
P, N = torch.from_numpy(prev).cuda(), torch.from_numpy(nxt).cuda()
ms = timeit(lambda: lk.calcOpticalFlow(P, N, 15, ctx=ctx))
line("lk::calcOpticalFlow 1080p win 15 (fused)", ms, 1080 * 1920, 16)
ms = timeit(lambda: lk.calcOpticalFlow(P, N, 43, ctx=ctx), iters=5)
line("lk::calcOpticalFlow 1080p win 43 (two-launch generic path)", ms, 1080 * 1920, 16)
ms = timeit(lambda: pyr.makeGaussianPyramid(P, 5, ctx=ctx))
line("makeGaussianPyramid 1080p 5 levels", ms, 1080 * 1920, 4 + 4 * 0.333 + 4)

# Host-pointer flavours (the cv::Mat shim's path): wall clock including H2D / D2H over PCIe.
import time


def wall(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    return (time.perf_counter() - t0) * 1e3 / iters


ms = wall(lambda: lk.calcOpticalFlowPyr(prev, nxt, 15, 5, ctx=ctx))
line("HOST lk::calcOpticalFlowPyr 1080p 5 levels win 15 (33 MB over PCIe)", ms, 1080 * 1920, 32)
ms = wall(lambda: stereo.disparitySSD(left, right, 5, -127, 0, ctx=ctx))
line("HOST stereo SSD 1080p r=5 d=128", ms, 1080 * 1920, 9)
img480 = synth.checkerboard(480, 640, 40, seed=1)
ms = wall(lambda: harris.getGradients(img480, 3, ctx=ctx))
line("HOST sobel3 pair 640x480", ms, 480 * 640, 12)

# "next" rows: matcher, edge front-end, motion history at representative sizes
from introtocomputervision_amd import match, mhi
rng = np.random.default_rng(1)
for nq, nt in ((2000, 2000), (8192, 8192)):
    q = torch.from_numpy(rng.random((nq, 128), dtype=np.float32) * 255).cuda()
    tr = torch.from_numpy(rng.random((nt, 128), dtype=np.float32) * 255).cuda()
    ms = timeit(lambda: match.knnMatch2(q, tr, ctx=ctx), iters=10)
    print(json.dumps({"kernel": f"BFMatcher knn2 {nq}x{nt}x128", "ms": round(ms, 4),
                      "GFLOPs": round(3 * nq * nt * 128 / ms / 1e6, 1)}))
img8 = torch.from_numpy((synth.smooth_noise(9, 1080, 1920)).astype(np.uint8)).cuda()
ms = timeit(lambda: hough.generateEdge(img8, 5, 1.4, 30, 90, ctx=ctx), iters=10)
line("generateEdge (blur 5 + Canny) 1080p u8", ms, 1080 * 1920, 2)
f1 = img8
f2 = torch.roll(img8, 3, 1).contiguous()
ms = timeit(lambda: mhi.frameDifference(f1, f2, 20, 5, 1.5, ctx=ctx), iters=10)
line("mhi::frameDifference (blur 5, open 7x7) 1080p u8", ms, 1080 * 1920, 3)

# C5: the whole 4K chain on one GPU (Harris -> corner list -> keypoint angles -> 5-level LK sampled at
# the corners), device resident, wall clock including the one host read-back of the corner count
tex = synth.smooth_noise(0x5EED0004, 2160, 3840)
chk = synth.checkerboard(2160, 3840, square=40)
p4 = np.round(tex * (chk / 192.0)).astype(np.float32)
n4 = np.ascontiguousarray(np.roll(p4, shift=(-2, 3), axis=(0, 1)))
P4, N4 = torch.from_numpy(p4).cuda(), torch.from_numpy(n4).cuda()


def c5():
    gx_, gy_ = harris.getGradients(P4, 3, ctx=ctx)
    R_ = harris.getCornerResponse(gx_, gy_, 5, 1.5, 0.04, ctx=ctx)
    _, locs_ = harris.refineCorners(R_, 5e8, 5, capacity=1 << 20, ctx=ctx)
    kp_ = harris.getKeypoints(gx_, gy_, locs_, 10, ctx=ctx)
    u_, v_ = lk.calcOpticalFlowPyr(P4, N4, 15, 5, ctx=ctx)
    return u_[locs_[:, 0].long(), locs_[:, 1].long()], kp_


ms = wall(lambda: (c5(), torch.cuda.synchronize()), iters=10)
line("C5 chain 3840x2160: Harris + corner list + keypoints + LK(5 levels) at corners", ms, 2160 * 3840, 26.6 + 12 + 12 + 9)

# the standalone pyramid / warp entry points (a4-a6) at 1080p
ms = timeit(lambda: pyr.pyrDown(P, ctx=ctx))
line("pyr::pyrDown 1080p", ms, 1080 * 1920, 4 * 0.25 + 4 * 0.25)
small = pyr.pyrDown(P, ctx=ctx)
ms = timeit(lambda: pyr.pyrUp(P, ctx=ctx))
line("pyr::pyrUp 1080p -> 2160p", ms, 4 * 1080 * 1920, 4 * 0.25 + 4)
du = torch.full_like(P, 2.3); dv = torch.full_like(P, -1.7)
ms = timeit(lambda: lk.warp(P, du, dv, ctx=ctx))
line("lk::warp 1080p", ms, 1080 * 1920, 16)
