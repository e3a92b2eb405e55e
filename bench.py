#!/usr/bin/env python3
"""bench.py -- Mpix/s through 5-level pyramidal Lucas-Kanade on 1080p frame pairs.

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment this process is only a LAUNCHER: before torch
or the GPU is touched it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...`
as a child, relays rank 0's JSON line and exits with the child's return code (the driver's own
torchrun launch sets WORLD_SIZE and runs the ranks directly).

One "step" = one pass of the hot path (lk::calcOpticalFlowPyr, 5 levels, win 15) over one
batch of --pairs synthetic 1080p frame pairs that already sit in HBM.  Frame pairs are
independent, so N GPUs = N ranks each with its own batch (weak scaling, no data-path
collective; RCCL carries only the timing barrier / MAX).  `--mode rowshard` splits every pair
by rows over the ranks instead (coarse-flow halo exchange per level, strong scaling).
Rank 0 prints ONE JSON line.  See DESIGN.md "Measurement" for the byte model.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROWS, COLS, WIN, LEVELS = 1080, 1920, 15, 5
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
HBM_COPY_GBS = 6290.0      # same guide: 6.29 TB/s measured float4 copy
VALU_PEAK_TFLOPS = 157.3   # same guide: peak FP32 vector (packed); 78.65 unpacked
SIMDS, CLOCK_GHZ, CYC_PER_VALU = 1024, 2.4, 4   # 256 CUs x 4 SIMDs; one wave64 VALU instruction = 4 cycles
# Irreducible arithmetic of one level pixel: five 15-tap separable window sums
# (150 FMA), two Sobel pairs + averages + It (~22), five products, pyrUp of two fields (20),
# bilinear warp (~12), 2x2 solve (~20) = ~230 FMA-equivalents = 460 flop.
USEFUL_FLOP_PER_LEVEL_PX = 460.0


def level_dims(rows, cols, levels):
    return [(rows >> l, cols >> l) for l in range(levels)]


def algorithmic_bytes_pair(rows, cols, levels):
    """BASELINE.md §3 byte model, per frame pair (f32, one HBM round trip per level)."""
    px = [r * c for r, c in level_dims(rows, cols, levels)]
    p0, s_all, s_up = px[0], sum(px), sum(px[1:])
    return 8 * p0 + 8 * s_up + 8 * s_up + 8 * s_all + 8 * s_up


def level0_kernel_bytes_pair(rows, cols, levels):
    """Algorithmic bytes of ONE launch of the dominant kernel (fused level-0 LK) per pair:
    read prev0 + next0 (8 B/px), write du, dv (8 B/px), read the coarse flow once (8 B per
    level-1 px)."""
    px = [r * c for r, c in level_dims(rows, cols, levels)]
    return 16 * px[0] + (8 * px[1] if levels > 1 else 0)


def cpu_baseline(sample_pairs, hip_uv=None):
    """The CPU oracle (a plain-C port of the reference algorithm, single thread -- the
    reference's own loops are single-threaded, OpticalFlow.cpp:85-103) timed on this host.
    `hip_uv(i)` returns the HIP path's (u, v) of bench pair i as numpy arrays: the oracle's result
    on the same pair is compared bit for bit (the checker's only other use here)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import _oracle as orc
    from introtocomputervision_amd import synth
    pairs = [synth.lk_pair(0x5EED0005 + i, ROWS, COLS, 3, -2) for i in range(sample_pairs)]
    t0 = time.perf_counter()
    res = [orc.lk_flow_pyr(p, n, WIN, LEVELS) for p, n in pairs]
    dt = time.perf_counter() - t0
    out = {
        "value": sample_pairs * ROWS * COLS / dt / 1e6,
        "unit": "Mpix/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{sample_pairs} of the same 1080p pairs, 5 levels, win 15, oracle/liboracle.so, {dt:.1f} s",
    }
    parity = None
    if hip_uv is not None:
        parity = {"pairs_compared": 0, "bit_exact": True, "mismatching_values": 0}
        for i, (eu, ev) in enumerate(res):
            got = hip_uv(i)
            if got is None:
                break
            bad = int(np.count_nonzero(got[0] != eu) + np.count_nonzero(got[1] != ev))
            parity["pairs_compared"] += 1
            parity["mismatching_values"] += bad
            parity["bit_exact"] = parity["bit_exact"] and bad == 0
        if parity["pairs_compared"] == 0:
            parity = None
    # The same oracle over all host cores: one frame pair per thread -- the ctypes
    # call releases the GIL and pairs are independent, so this is the data-parallel CPU ceiling.
    ncores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count(), 32)
    if ncores > 1:
        from concurrent.futures import ThreadPoolExecutor
        work = [pairs[i % len(pairs)] for i in range(ncores)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(ncores) as ex:
            list(ex.map(lambda pn: orc.lk_flow_pyr(pn[0], pn[1], WIN, LEVELS), work))
        dt = time.perf_counter() - t0
        out["all_cores"] = {"value": ncores * ROWS * COLS / dt / 1e6, "unit": "Mpix/s", "cores": ncores,
                            "sample": f"{ncores} pairs, one per thread, {dt:.1f} s"}
    return out, parity


_PROFILER_ENV_PREFIXES = ("ROCP_", "ROCPROF", "ROCPROFILER", "ROCTRACER", "ROCTX", "HSA_TOOLS", "RPDT_")


def under_profiler():
    """True when this process already runs under rocprofv3 / rocprof (its tool library is preloaded or its
    environment is set): the live PMC pass must not start counter-collecting children from inside a traced
    process -- tracing and --pmc may not be mixed on this pool, and the nested passes can wedge."""
    if any(k.startswith(_PROFILER_ENV_PREFIXES) for k in os.environ):
        return True
    return "rocprof" in os.environ.get("LD_PRELOAD", "").lower()


def scrubbed_env():
    """The environment for a profiler child: nothing inherited from an enclosing profiler."""
    env = {k: v for k, v in os.environ.items() if not k.startswith(_PROFILER_ENV_PREFIXES)}
    pre = [x for x in env.get("LD_PRELOAD", "").replace(":", " ").split() if "rocprof" not in x.lower()]
    if pre:
        env["LD_PRELOAD"] = ":".join(pre)
    else:
        env.pop("LD_PRELOAD", None)
    env["TMPDIR"] = "/tmp"
    env.pop("MICV_BENCH_FORCE_DIST", None)
    return env


def measure_traffic_live(pairs, extra_opts):
    """HBM traffic and VALU instructions of the level-0 launch, measured NOW: three child runs of this
    bench under `rocprofv3 --pmc` (FETCH_SIZE, WRITE_SIZE and SQ_INSTS_VALU in separate passes, no tracing
    flags -- the pool's rule and the guide's recipe), one stream group and one pass at a time so that a
    level-0 dispatch covers the whole batch.  FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled
    (MI355X_MICROARCH.md section HBM: gfx950 tallies 64 B per 128-B request of a wide coalesced read).
    Returns None when rocprofv3 is missing or a pass fails (the caller then falls back to the committed
    profiles/traffic.json, labelled as such)."""
    import csv
    import glob
    import shutil
    import tempfile
    if not shutil.which("rocprofv3") or under_profiler():
        return None
    means = {}
    kernel = None
    tmp = tempfile.mkdtemp(prefix="micv_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out, "--",
                   "python3", os.path.abspath(__file__), "--cpu-pairs", "0", "--steps", "3", "--warmup", "1",
                   "--no-profile-pass", "--lk-groups", "1", "--inflight", "1", "--sustained-s", "0", "--preroll-s", "0",
                   "--pairs", str(pairs), "--no-pmc", "--no-secondary"] + [x for o in extra_opts for x in ("--opt", o)]
            r = subprocess.run(cmd, cwd="/tmp", env=scrubbed_env(), capture_output=True, text=True, timeout=240)
            if r.returncode != 0:
                return None
            per = {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "lk_level" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        per.setdefault((row["Kernel_Name"].split("(")[0].replace("void ", ""), int(row["Grid_Size"])),
                                       []).append(float(row["Counter_Value"]))
            if not per:
                return None
            key = max(per, key=lambda k: k[1])  # the largest grid = pyramid level 0
            means[counter] = sum(per[key]) / len(per[key])
            kernel = key[0]
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch, write = means["FETCH_SIZE"] * 1024, means["WRITE_SIZE"] * 1024
    return {"kernel": kernel, "pairs_per_launch": pairs, "fetch_bytes_raw": fetch, "fetch_bytes_x2_gfx950": 2 * fetch,
            "write_bytes": write, "level0_hbm_bytes_per_launch": 2 * fetch + write,
            "valu_insts_per_launch": means["SQ_INSTS_VALU"],
            "method": "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU, three child "
                      "passes of bench.py (3 steps, one stream group, one pass at a time); KiB -> bytes; FETCH_SIZE "
                      "doubled per MI355X_MICROARCH.md section HBM"}


def secondary_block(ctx, stream, torch, np):
    """BASELINE.json's other single-GPU configs in the driver-run line: C3 (ps2 stereo SSD + NCC,
    1080p, 11x11 window, 128 disparities; DisparitySSD.cu:143-207), C1 (ps4 Harris chain on 480x640 with the
    parameters of the reference's own config/ps4.yaml; Harris.cu:96-159,243-329) and C5 (4K Harris -> ordered corner
    list -> keypoints -> descriptors -> 5-level LK sampled at the corners).  Device-resident inputs, HIP-event time per
    call; every entry carries an in-run bit-exact check of the SAME functions against the CPU oracle at a size the
    oracle covers in about a second (the oracle is the checker here, never the thing timed)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as orc
    from introtocomputervision_amd import config, harris, lk, stereo, synth
    from introtocomputervision_amd._capi import Timer

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def timeit(fn, iters=10, warm=2, reps=3):  # the median of `reps` timed runs of `iters` calls each
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        runs = []
        for _ in range(reps):
            t = Timer()
            t.start(stream)
            for _ in range(iters):
                fn()
            t.stop(stream)
            runs.append(t.elapsed_ms() / iters)
        return sorted(runs)[reps // 2]

    def wall(fn, iters=10, warm=2, reps=3):  # chains with a host read-back inside (the corner count)
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        runs = []
        for _ in range(reps):
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
            runs.append((time.perf_counter() - t0) * 1e3 / iters)
        return sorted(runs)[reps // 2]

    out = {}
    # ---- C3: stereo, 1080p, r = 5, d in [-127, 0] (128 candidates fit int8) ------------------------------
    rows, cols, rad, nd = 1080, 1920, 5, 128
    left, right, _ = synth.stereo_pair(0x5EED0002, rows, cols)
    L, R = dev(left), dev(right)
    sl, sr, _ = synth.stereo_pair(0x5EED0002, 160, 320)   # the check: same generator, 160x320, 48 disparities
    SL, SR = dev(sl), dev(sr)
    # box-filter form of the window sum: per pixel and disparity 1 subtract + 1 multiply + (2r+1) column adds
    # + (2r+1) row adds + compare = 2 (2r+1) + 3 flop
    flop_pd = 2 * (2 * rad + 1) + 3
    from introtocomputervision_amd._capi import Context as _Ctx, OPT_STEREO_EXACT
    fctx = _Ctx(ctx.device)                  # the float kernels only (what every call took up to r05)
    fctx.set_option(OPT_STEREO_EXACT, -1)
    for key, fn, ofn in (("C3_ssd", stereo.disparitySSD, orc.disparity_ssd), ("C3_ncc", stereo.disparityNCorr, orc.disparity_ncorr)):
        ms = timeit(lambda: fn(L, R, rad, -(nd - 1), 0, ctx=ctx))
        exp = ofn(sl, sr, rad, -47, 0)
        got = fn(SL, SR, rad, -47, 0, ctx=ctx).cpu().numpy()
        px = rows * cols
        path = {}
        if key == "C3_ssd":
            # the pair is 8-bit-valued (as every plain ps2 call's, main.cpp:87-88): the pre-pass finds that on the device and
            # the exact-sum kernels (stereo_exact.hip) run; same disparities as the float kernels, timed beside them
            fms = timeit(lambda: fn(L, R, rad, -(nd - 1), 0, ctx=fctx))
            same = bool(np.array_equal(fn(L, R, rad, -(nd - 1), 0, ctx=ctx).cpu().numpy(), fn(L, R, rad, -(nd - 1), 0, ctx=fctx).cpu().numpy()))
            path = {"kernels": "exact-sum (8-bit-valued pair found on the device): pre-pass + search + the float kernel's empty launch",
                    "float_kernels_ms": fms, "same_as_float_kernels_at_full_size": same}
        out[key] = {
            **path,
            "workload": f"{cols}x{rows} rectified synthetic pair, {2 * rad + 1}x{2 * rad + 1} window, {nd} disparities, device-resident",
            "ms": ms, "Mpix_per_s": px / ms / 1e3, "Gpix_disp_per_s": px * nd / ms / 1e6,
            "algorithmic_bytes_per_px": 9, "frac_of_hbm_peak": px * 9 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "valu_flop_per_px_disp": flop_pd if key == "C3_ssd" else None,
            "frac_of_fp32_vector_peak": (px * nd * flop_pd / (ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS) if key == "C3_ssd" else None,
            "bound": "valu",
            "check": {"vs": "oracle, 320x160, 48 disparities, same window", "bit_exact": bool(np.array_equal(got, exp)),
                      "mismatching_values": int(np.count_nonzero(got != exp))},
        }
    # ---- C1: ps4 Harris chain, 480x640, parameters from the reference's config/ps4.yaml ------------------
    cfg = config.load(os.path.join(ROOT, "tests", "golden", "config", "ref", "ps4.yaml"))
    hp = config.harris_params(cfg, "harris_trans")
    img = synth.checkerboard(480, 640, square=40, seed=0x5EED0001)
    dimg = dev(img)

    def c1(cpu_arith=False):
        gx, gy = harris.getGradients(dimg, hp["sobel_kernel_size"], ctx=ctx)
        Rr = harris.getCornerResponse(gx, gy, hp["window_size"], hp["gaussian_sigma"], hp["alpha"], ctx=ctx, cpu_arithmetic=cpu_arith)
        _, locs = harris.refineCorners(Rr, hp["response_threshold"], hp["min_distance"], ctx=ctx)
        return Rr, locs
    ms3 = wall(c1)
    Rg, lg = c1()
    Rc, lc = c1(True)

    # the same chain as ONE call (micv_harris_corners_dev, r05): Sobel inside the response kernel's tile, R in context
    # scratch, only the list leaves the device -- image in (4 B), R out and back in (4 + 4 B)
    def c1_chain(cpu_arith=False, **kw):
        return harris.cornersFromImage(dimg, hp["sobel_kernel_size"], hp["window_size"], hp["gaussian_sigma"], hp["alpha"],
                                       hp["response_threshold"], hp["min_distance"], ctx=ctx, cpu_arithmetic=cpu_arith,
                                       want_gradients=False, **kw)
    ms = wall(c1_chain)
    f_g, f_c = c1_chain(want_response=True), c1_chain(True, want_response=True)
    chain_same = bool(torch.equal(f_g["response"], Rg) and torch.equal(f_g["locs"], lg) and
                      f_c["response"].cpu().numpy().tobytes() == Rc.cpu().numpy().tobytes() and torch.equal(f_c["locs"], lc))
    ogx, ogy = orc.sobel(img, hp["sobel_kernel_size"], 1.0)
    eR = orc.harris_response(ogx, ogy, hp["window_size"], hp["gaussian_sigma"], hp["alpha"])
    _, el = orc.harris_refine(eR, hp["response_threshold"], hp["min_distance"])
    eRc = orc.harris_response_ex(ogx, ogy, hp["window_size"], hp["gaussian_sigma"], hp["alpha"], orc.HARRIS_CPU)
    _, elc = orc.harris_refine(eRc, hp["response_threshold"], hp["min_distance"])
    out["C1_harris"] = {
        "workload": "640x480 greyscale checkerboard, config/ps4.yaml harris_trans (sobel 3, window 5, sigma 1.5, alpha 0.04, "
                    "threshold 5e8, minDistance 5): getGradients -> getCornerResponse -> refineCorners as one call "
                    "(micv_harris_corners_dev: image -> R -> ordered list), corner count read back",
        "ms": ms, "Mpix_per_s": 480 * 640 / ms / 1e3, "corners": int(len(lg)),
        "three_separate_calls_ms": ms3,
        "algorithmic_bytes_per_px": 12, "frac_of_hbm_peak": 480 * 640 * 12 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "bound": "latency (two launches + one read-back on 0.3 Mpx)",
        "reference_gtx1080_ms": {"cornerResponseKernel": 0.80, "refineCornersKernel": 0.59},
        "check": {"vs": "oracle at full size, gpu:: and cpu:: arithmetic",
                  "bit_exact": bool(np.array_equal(Rg.cpu().numpy(), eR) and np.array_equal(lg.cpu().numpy(), el)
                                    and Rc.cpu().numpy().tobytes() == eRc.tobytes() and np.array_equal(lc.cpu().numpy(), elc)
                                    and chain_same),
                  "one_call_equals_three_calls": chain_same},
    }
    # ---- C5: 4K Harris + descriptors + LK refine ------------------------------------------------------------
    def c5_frames(rows, cols):
        tex = synth.smooth_noise(0x5EED0004, rows, cols)
        chk = synth.checkerboard(rows, cols, square=40)
        p = np.round(tex * (chk / 192.0)).astype(np.float32)
        return p, np.ascontiguousarray(np.roll(p, shift=(-2, 3), axis=(0, 1)))

    count_host = torch.empty(1, dtype=torch.int64).pin_memory()

    def c5(P, N):
        # Harris leaves its corner count on the device (lazy); the flow of the pair does not depend on it and is queued
        # before the host reads the count, so the read-back's round trip hides behind the LK launches
        h = harris.cornersFromImage(P, 3, 5, 1.5, 0.04, 5e8, 5, capacity=1 << 20, ctx=ctx, lazy=True)  # gradients kept: the keypoints read them
        count_host.copy_(h["count"], non_blocking=True)
        counted = torch.cuda.Event()
        counted.record()
        u_, v_ = lk.calcOpticalFlowPyr(P, N, WIN, LEVELS, ctx=ctx)
        counted.synchronize()
        gx, gy, locs = h["gx"], h["gy"], h["locs"][:min(int(count_host[0]), 1 << 20)]
        kp = harris.getKeypoints(gx, gy, locs, 10, ctx=ctx)
        desc = harris.computeDescriptors(gx, gy, kp, ctx=ctx)
        ll = locs.long()  # one conversion kernel for both index columns
        yy, xx = ll[:, 0], ll[:, 1]
        return locs, kp, desc, u_[yy, xx], v_[yy, xx]
    p4, n4 = c5_frames(2160, 3840)
    P4, N4 = dev(p4), dev(n4)
    ms = wall(lambda: c5(P4, N4), iters=6)
    n_corners = int(len(c5(P4, N4)[0]))
    ps, ns = c5_frames(270, 480)
    locs, kp, desc, fu, fv = c5(dev(ps), dev(ns))
    ogx, ogy = orc.sobel(ps, 3, 1.0)
    eR = orc.harris_response(ogx, ogy, 5, 1.5, 0.04)
    _, el = orc.harris_refine(eR, 5e8, 5)
    ekp = orc.sift_keypoints(ogx, ogy, el, 10)
    edesc = orc.sift_descriptors(ogx, ogy, ekp)
    eu, ev = orc.lk_flow_pyr(ps, ns, WIN, LEVELS if 270 >> (LEVELS - 1) >= 1 else 4)
    same_list = np.array_equal(locs.cpu().numpy(), el)
    kpn = kp.cpu().numpy()
    ok = bool(same_list and np.array_equal(kpn[:, :3], ekp[:, :3]) and np.allclose(kpn[:, 3], ekp[:, 3], atol=1e-3, rtol=0)
              and np.array_equal(fu.cpu().numpy(), eu[el[:, 0], el[:, 1]]) and np.array_equal(fv.cpu().numpy(), ev[el[:, 0], el[:, 1]]))
    # descriptors depend on the keypoint angle (device atan2f vs libm, 1e-3 deg): compare on the oracle's keypoints
    gdesc = harris.computeDescriptors(dev(ogx), dev(ogy), dev(ekp), ctx=ctx).cpu().numpy() if len(ekp) else edesc
    ok = ok and bool(np.array_equal(gdesc, edesc))
    px = 2160 * 3840
    bpp = algorithmic_bytes_pair(2160, 3840, LEVELS) / px + 4 + 8 + 4 + 4   # LK + image in, gradients out, R out, R back in (NMS)
    out["C5_4k_chain"] = {
        "workload": "3840x2160 textured checkerboard pair: Harris (sobel 3, window 5; one call, image -> gradients + R -> ordered corner list) -> keypoints -> "
                    "4x4x8 descriptors; 5-level LK (win 15) of the pair queued behind Harris and sampled at the corners; one corner-count read-back, "
                    "hidden behind the LK launches",
        "ms": ms, "Mpix_per_s": px / ms / 1e3, "corners": n_corners,
        "algorithmic_bytes_per_px": bpp, "frac_of_hbm_peak": px * bpp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "bound": "valu (LK level 0) + latency (list read-back, descriptors on a short list)",
        "check": {"vs": "oracle on the same chain at 480x270 (corner list, keypoints, descriptors, flow at the corners)",
                  "bit_exact": ok, "corners_checked": int(len(el))},
    }
    return out



def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pairs", type=int, default=8, help="frame pairs per GPU per step")
    ap.add_argument("--cpu-pairs", type=int, default=6, help="pairs timed on the CPU baseline (0 = skip)")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--inflight", type=int, default=2, help="passes in flight (contexts/streams), pairs mode")
    ap.add_argument("--lk-groups", type=int, default=0,
                    help="stream groups the library splits a batch into (0 = its default, 1); the PMC "
                         "passes of tools/profile.sh use 1 so a level-0 dispatch covers the whole batch")
    ap.add_argument("--lk-chain", type=int, default=0,
                    help="MICV_OPT_LK_CHAIN of every context: 0 = the library's rule (pairs of tiles only for launches just over whole rounds), 1 = no tile chains, n = longest chain")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="micv_ctx_set_option on every context, e.g. --opt OPT_LK_TALL_TILES=1 (A/B and PMC runs)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure roofline.traffic / roofline.valu live (three short rocprofv3 --pmc child "
                         "runs, ~25 s); fall back to the committed profiles/traffic.json")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` block (BASELINE configs C1 / C3 / C5 timed and checked in this run, ~3 s of GPU)")
    ap.add_argument("--preroll-s", type=float, default=0.25,
                    help="seconds of untimed steps before the warm-up steps (clock pre-roll; 0 = none)")
    ap.add_argument("--sustained-s", type=float, default=2.0,
                    help="seconds of back-to-back steps for the `sustained` field (0 = skip)")
    ap.add_argument("--mode", choices=["pairs", "rowshard"], default="pairs",
                    help="pairs: every rank owns whole frame pairs (default, weak scaling, no data-path "
                         "collective); rowshard: every pair is split by rows over all ranks with a "
                         "coarse-flow halo exchange per level (strong scaling, RCCL point-to-point)")
    ap.add_argument("--next-margin", type=int, default=None,
                    help="rowshard mode: declared bound on |dv| in level-0 rows; ranks build `next` for band + margin only, "
                         "a device-side check makes a step fall back to the whole frame when the bound is exceeded "
                         "(default: whole frame on every rank)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the ranks (nccl = RCCL; gloo only with --dry-run)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no kernels and no GPU: exercises launcher, rendezvous, barrier, MAX over ranks "
                         "and the JSON line (value is null); what the CPU tests run")
    return ap.parse_args(argv)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """--gpus N > 1 without a torchrun environment: start the N ranks as a CHILD process (this
    process has not imported torch, so nothing here has touched a GPU), relay rank 0's JSON line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    relayed = False
    for line in proc.stdout.splitlines():
        if line.startswith("{") and '"metric"' in line:
            print(line)
            relayed = True
        else:
            print(line, file=sys.stderr)
    if proc.returncode == 0 and not relayed:
        print("[bench] the ranks exited 0 without a JSON line", file=sys.stderr)
        return 1
    return proc.returncode


def valu_roofline(lvl0_ms, pairs_per_launch, tj=None, source=None):
    """VALU axis of the dominant kernel: wave-VALU instructions per launch (SQ_INSTS_VALU, measured live
    by measure_traffic_live or taken from the committed profiles/traffic.json), the issue time they need
    on 1024 SIMDs at 4 cycles each, and the useful arithmetic rate against the FP32 vector peak."""
    if tj is None:
        path = os.path.join(ROOT, "profiles", "traffic.json")
        try:
            tj = json.load(open(path))
        except Exception:
            return None
        source = f"profiles/traffic.json ({tj.get('profile', 'rocprofv3 --pmc SQ_INSTS_VALU')}, not this run)"
    insts = tj.get("valu_insts_per_launch")
    if not insts or tj.get("pairs_per_launch") != pairs_per_launch:
        return None
    px = ROWS * COLS * pairs_per_launch
    issue_us = insts * CYC_PER_VALU / (SIMDS * CLOCK_GHZ * 1e3)
    useful_tflops = USEFUL_FLOP_PER_LEVEL_PX * px / (lvl0_ms * 1e-3) / 1e12
    return {
        "wave_valu_insts_per_launch": insts,
        "valu_insts_per_output_px": insts * 64 / px,
        "issue_bound_us": issue_us,
        "frac_of_issue_bound": issue_us / (lvl0_ms * 1e3),
        "useful_flop_per_px": USEFUL_FLOP_PER_LEVEL_PX,
        "useful_tflops": useful_tflops,
        "peak_tflops": VALU_PEAK_TFLOPS,
        "frac": useful_tflops / VALU_PEAK_TFLOPS,
        "source": source,
    }


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    if args.backend == "gloo" and not args.dry_run:
        sys.exit("[bench] --backend gloo is for --dry-run only: the hot path needs a GPU (no CPU fallback)")

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("MICV_BENCH_FORCE_DIST"):  # the env knob exercises the RCCL path at N=1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_run:
            dist.init_process_group(args.backend if args.backend == "gloo" or torch.cuda.is_available() else "gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(args.backend, device_id=torch.device("cuda", local_rank))
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}; running {n_gpus}", file=sys.stderr)
    B = args.pairs

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_and_all(dt, device):
        """MAX over ranks of a wall time + every rank's own value."""
        if dist is None:
            return dt, [dt]
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(every, t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), [float(x.item()) for x in every]

    if args.dry_run:
        # launcher / rendezvous / reduction path only: no kernels, nothing measured
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pass
        barrier()
        dt, every = max_and_all(time.perf_counter() - t0, torch.device("cpu"))
        if rank == 0:
            print(json.dumps({
                "metric": "Mpix/s (LK 5-level pyramid, 1080p pairs)", "value": None, "unit": "Mpix/s",
                "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                "higher_is_better": True, "scaling": "weak" if args.mode == "pairs" else "strong",
                "vs_baseline": None, "dtype": "f32", "data": "none (dry run: no kernels launched)",
                "config": {"workload": "dry run", "ranks": dist.get_world_size() if dist is not None else 1,
                           "backend": args.backend, "per_rank_s": every},
            }))
        if dist is not None:
            dist.destroy_process_group()
        return

    from introtocomputervision_amd import lk, synth
    from introtocomputervision_amd._capi import Context
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    # Synthetic C2/C4 frame pairs, distinct per rank, resident in HBM.
    prev_h = np.empty((B, ROWS, COLS), np.float32)
    next_h = np.empty((B, ROWS, COLS), np.float32)
    for i in range(B):
        prev_h[i], next_h[i] = synth.lk_pair(0x5EED0005 + rank * B + i, ROWS, COLS, 3, -2)
    prev = torch.from_numpy(prev_h).to(dev)
    nxt = torch.from_numpy(next_h).to(dev)
    u = torch.empty_like(prev)
    v = torch.empty_like(prev)
    ctx = Context(local_rank)
    ctx.set_lk_groups(args.lk_groups)
    from introtocomputervision_amd import _capi as _c
    ctx.set_option(_c.OPT_LK_CHAIN, args.lk_chain)

    def apply_opts(c):
        for kv in args.opt:
            k, v = kv.split("=")
            c.set_option(getattr(_c, k), int(v))
    apply_opts(ctx)
    stream = torch.cuda.current_stream(dev).cuda_stream

    if args.mode == "rowshard":
        # All ranks hold the same pairs (replicated inputs); each computes its row band of every pair.
        from introtocomputervision_amd import shard
        if dist is not None:
            for t in (prev, nxt):
                dist.broadcast(t, src=0)
        if args.next_margin is None:
            # the C ABI's own driver (micv_lk_flow_pyr_rowshard_dev: band launches + ncclSend / ncclRecv halo exchange
            # on the launch stream) -- the path a C++ caller links; torch.distributed only carries the RCCL unique id
            runner = shard.RowShardNative(ctx, ROWS, COLS, LEVELS, WIN, B, shard.MicvComm(ctx, rank, n_gpus, dist=dist))
        else:  # declared bound on |dv|: the Python driver with its device-side check and whole-frame fallback
            runner = shard.RowShardBatch(ctx, ROWS, COLS, LEVELS, WIN, B, rank, n_gpus,
                                         comm=shard.DistComm(rank, n_gpus) if dist is not None else None,
                                         next_margin=args.next_margin)
        a0, b0 = runner.band0
        margin_stats = [0, 0]  # steps, steps on which the declared margin sufficed

        def step():
            if args.next_margin is None:
                runner.run(prev, nxt, u, v, stream)
            else:
                margin_stats[0] += 1
                margin_stats[1] += 1 if runner.run_checked(prev, nxt, u, v, stream) else 0
    else:
        # `--inflight F` batches in flight: step i runs on context / HIP stream / output buffers
        # i mod F (a context owns its scratch arena and aux streams, so passes on different contexts
        # are independent).  With F = 2 the latency-bound coarse pyramid levels of one pass overlap
        # the throughput-bound fine levels of the previous one -- double buffering, as a video
        # pipeline would run it.  Every pass still does all of its work and writes its own outputs.
        F = max(1, args.inflight)
        lanes = [(ctx, torch.cuda.current_stream(dev), (u, v))]
        for _ in range(1, F):
            lanes.append((Context(local_rank), torch.cuda.Stream(dev), (torch.empty_like(prev), torch.empty_like(prev))))
            lanes[-1][0].set_lk_groups(args.lk_groups)
            lanes[-1][0].set_option(_c.OPT_LK_CHAIN, args.lk_chain)
            apply_opts(lanes[-1][0])
        counter = [0]

        def step():
            c, st, out = lanes[counter[0] % F]
            counter[0] += 1
            lk.calcOpticalFlowPyrBatch(prev, nxt, WIN, LEVELS, ctx=c, out=out, stream=st.cuda_stream)

    def timed(steps):
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        return max_and_all(time.perf_counter() - t0, dev)

    # N > 1, before anything is timed: (1) micv_comm_selftest -- the C ABI's communicator over RCCL, a ring of grouped
    # ncclSend / ncclRecv of a rank-stamped slab + an int32 all-reduce, verified on the device -- so that a fabric or
    # ordering failure is reported as such and not as a parity mismatch (rowshard mode: its own communicator, a failure
    # is fatal; pairs mode: an extra communicator used for nothing else, a failure is recorded, the data path has no
    # collective to break); (2) in pairs mode every rank's first pair is checked against rank 0's own recomputation of
    # that rank's seed after the timed runs (`per_rank_parity` below).
    comm_selftest = None
    if dist is not None:  # (also at world 1 when the RCCL path is forced: tests/test_rccl_gpu.py)
        try:
            if args.mode == "rowshard" and args.next_margin is None:
                runner.comm.selftest(stream)
                comm_selftest = "ok"
            else:
                # pairs mode: the communicator is diagnostic only, so it must not be able to take the measurement down
                # with it silently -- it runs on a helper thread with a deadline (ctypes releases the GIL); a failure is
                # reported in the line and the bench goes on, a rank stuck inside ncclCommInitRank ends the run (below)
                import threading
                from introtocomputervision_amd import shard as _shard
                box = {}

                def _run():
                    try:
                        torch.cuda.set_device(dev)
                        _tc = _shard.MicvComm(ctx, rank, n_gpus, dist=dist)
                        _tc.selftest(stream)
                        _tc.close()
                        box["r"] = "ok"
                    except Exception as e:  # noqa: BLE001
                        box["r"] = f"FAILED on rank {rank}: {e}"
                th = threading.Thread(target=_run, daemon=True)
                th.start()
                th.join(timeout=float(os.environ.get("MICV_BENCH_SELFTEST_TIMEOUT_S", "90")))
                if "r" not in box:
                    # the helper may still be inside a collective on this process group and stream; going on to
                    # all_gather_object and the timed steps beside it could interleave collectives out of order across the
                    # ranks (a hang) or put RCCL work into the measured region.  A rank that cannot build a communicator in
                    # 90 s has no measurement to give: say so and leave (the launcher ends the other ranks).
                    sys.stderr.write(json.dumps({"error": f"comm selftest TIMEOUT on rank {rank}: no answer from micv_comm_create / "
                                                          "micv_comm_selftest; nothing was timed"}) + "\n")
                    sys.stderr.flush()
                    os._exit(3)
                comm_selftest = box["r"]
        except Exception as e:  # noqa: BLE001 -- reported in the JSON line
            comm_selftest = f"FAILED on rank {rank}: {e}"
            if args.mode == "rowshard":
                raise
        gathered = [None] * world
        dist.all_gather_object(gathered, comm_selftest)
        comm_selftest = "ok" if all(g == "ok" for g in gathered) else "; ".join(str(g) for g in gathered if g != "ok")

    ctx.warmup(stream)
    # Pre-roll (untimed, before the W warm-up steps): the first ~30 ms of work after the device has
    # idled run ~15 % slower (power state ramp; measured 43 vs 50 Gpix/s at W=5, K=20), and W steps are
    # only W x 0.35 ms.  A quarter second of the same steps puts the timed window at sustained clocks,
    # which is what `sustained` (two seconds of steps) reports independently.
    if args.preroll_s > 0:
        t_end = time.perf_counter() + args.preroll_s
        while time.perf_counter() < t_end:
            for _ in range(8):
                step()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    dt, per_rank = timed(args.steps)

    # the same code path for >= --sustained-s seconds back to back: clocks under sustained load
    sustained = None
    if args.sustained_s > 0:
        n_s = max(args.steps, int(math.ceil(args.sustained_s / (dt / args.steps))))
        dts, _ = timed(n_s)
        sustained = {"seconds": dts, "steps": n_s, "ms_per_step": dts / n_s * 1e3,
                     "ratio_to_short_run": (dts / n_s) / (dt / args.steps)}

    # the same K steps one at a time on one context (no overlap between passes): reported beside
    # `value` so the effect of keeping two passes in flight is visible
    serial_ms = None
    single_pair_ms = None
    if args.mode == "pairs":
        if max(1, args.inflight) > 1:
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(args.steps):
                lk.calcOpticalFlowPyrBatch(prev, nxt, WIN, LEVELS, ctx=ctx, out=(u, v), stream=stream)
            torch.cuda.synchronize()
            serial_ms = (time.perf_counter() - ts) / args.steps * 1e3
        # BASELINE configs[1] read literally: ONE 1080p pair per call, calls back to back
        p1, n1, o1 = prev[:1], nxt[:1], (torch.empty_like(prev[:1]), torch.empty_like(prev[:1]))
        for _ in range(5):
            lk.calcOpticalFlowPyrBatch(p1, n1, WIN, LEVELS, ctx=ctx, out=o1, stream=stream)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        n1p = max(50, args.steps * 4)
        for _ in range(n1p):
            lk.calcOpticalFlowPyrBatch(p1, n1, WIN, LEVELS, ctx=ctx, out=o1, stream=stream)
        torch.cuda.synchronize()
        single_pair_ms = (time.perf_counter() - ts) / n1p * 1e3
        lk.calcOpticalFlowPyrBatch(prev, nxt, WIN, LEVELS, ctx=ctx, out=(u, v), stream=stream)
        torch.cuda.synchronize()

    # The drop-in path a cv::Mat caller links (micv_lk_flow_pyr_host: upload, kernels, download, sync in
    # every call, like OpticalFlow.cpp:12-39 / Pyramids.cu:34-73): one pair per call, pageable inputs,
    # preallocated outputs.  PCIe-inclusive, reported beside `value`, never as `value`.
    host_pair_ms = host_seq_ms = host_seq8_ms = host_seq_same = None
    if args.mode == "pairs" and rank == 0:
        from introtocomputervision_amd._capi import check as _check, lib as _lib
        hu = np.zeros((ROWS, COLS), np.float32)
        hv = np.zeros((ROWS, COLS), np.float32)
        hctx = Context(local_rank)

        def host_call():
            _check(_lib.micv_lk_flow_pyr_host(hctx.handle, prev_h[0].ctypes.data, next_h[0].ctypes.data, ROWS, COLS,
                                              COLS * 4, WIN, LEVELS, hu.ctypes.data, hv.ctypes.data, COLS * 4))
        for _ in range(3):
            host_call()
        ts = []
        for _ in range(15):
            t0 = time.perf_counter()
            host_call()
            ts.append(time.perf_counter() - t0)
        host_pair_ms = sorted(ts)[len(ts) // 2] * 1e3
        # the same boundary fed the way the ps5 driver feeds it -- consecutive frames of a sequence (Solution.cpp:255-285):
        # micv_lk_flow_seq_host, every frame uploaded once, upload / chain / download of consecutive pairs overlapped
        nseq = 16

        def page_aligned(n_img):  # images on pages of their own, as separately allocated cv::Mats are (one registration each)
            raw = np.zeros(n_img * ROWS * COLS + 1024, np.float32)
            off = (-raw.ctypes.data % 4096) // 4
            return raw[off:off + n_img * ROWS * COLS].reshape(n_img, ROWS, COLS)
        sf = page_aligned(nseq)
        for i in range(nseq):
            sf[i] = prev_h[i % len(prev_h)] if i % 2 == 0 else next_h[i % len(next_h)]
        seq = [sf[i] for i in range(nseq)]
        su, sv = page_aligned(nseq - 1), page_aligned(nseq - 1)   # preallocated and touched, like hu / hv above
        lk.calcOpticalFlowPyrSequence(seq, WIN, LEVELS, ctx=hctx, out=(su, sv))
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            lk.calcOpticalFlowPyrSequence(seq, WIN, LEVELS, ctx=hctx, out=(su, sv))
            ts.append((time.perf_counter() - t0) / (nseq - 1))
        host_seq_ms = sorted(ts)[len(ts) // 2] * 1e3
        pu, pv = lk.calcOpticalFlowPyrFrames(seq[4], seq[5], WIN, LEVELS, ctx=hctx)
        host_seq_same = bool(np.array_equal(su[4], pu) and np.array_equal(sv[4], pv))
        # ... and on 8-bit frames, what the driver's cv::imread hands over (the synthetic frames ARE 8-bit-valued): the
        # upload shrinks to 2 MB per frame, the device converts (micv_to_gray_f32_dev)
        seq8 = [f.astype(np.uint8) for f in seq]
        lk.calcOpticalFlowPyrSequence(seq8, WIN, LEVELS, ctx=hctx, out=(su, sv))
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            lk.calcOpticalFlowPyrSequence(seq8, WIN, LEVELS, ctx=hctx, out=(su, sv))
            ts.append((time.perf_counter() - t0) / (nseq - 1))
        host_seq8_ms = sorted(ts)[len(ts) // 2] * 1e3
        host_seq_same = host_seq_same and bool(np.array_equal(su[4], pu) and np.array_equal(sv[4], pv))
        hctx.close()

    # sanity of what was measured: known translation comes back (not part of the timing)
    if args.mode == "rowshard":
        chk_u, chk_v = u[0, a0 + 8:b0 - 8, 64:-64], v[0, a0 + 8:b0 - 8, 64:-64]
    else:
        chk_u, chk_v = u[0, 64:-64, 64:-64], v[0, 64:-64, 64:-64]
    um = float(chk_u.median())
    vm = float(chk_v.median())
    ok = abs(um - 3.0) < 0.25 and abs(vm + 2.0) < 0.25

    # Per-rank parity (N > 1, pairs mode): an exact checksum of every rank's first pair (sums of the flow's bit patterns,
    # plain and position-weighted, as int64) is gathered; rank 0 recomputes each rank's pair from its seed on its own
    # GPU and compares -- the SCALE line then carries every rank's result, not rank 0's only.
    def flow_checksum(fu, fv):
        out = []
        for f in (fu, fv):
            b = f.contiguous().view(torch.int32).to(torch.int64).flatten()
            w = (torch.arange(b.numel(), device=b.device, dtype=torch.int64) % 65521) + 1
            out += [int(b.sum().item()), int((b * w).sum().item())]
        return out
    per_rank_parity = None
    if dist is not None and args.mode == "pairs":
        lk.calcOpticalFlowPyrBatch(prev, nxt, WIN, LEVELS, ctx=ctx, out=(u, v), stream=stream)
        torch.cuda.synchronize()
        mine = flow_checksum(u[0], v[0])
        sums = [None] * world
        dist.all_gather_object(sums, mine)
        if rank == 0:
            bad = []
            for r in range(1, world):
                pr, nr = synth.lk_pair(0x5EED0005 + r * B, ROWS, COLS, 3, -2)
                ru, rv = lk.calcOpticalFlowPyrBatch(torch.from_numpy(pr[None]).to(dev), torch.from_numpy(nr[None]).to(dev), WIN, LEVELS, ctx=ctx)
                torch.cuda.synchronize()
                if flow_checksum(ru[0], rv[0]) != sums[r]:
                    bad.append(r)
            per_rank_parity = {"ranks_checked": world, "mismatching_ranks": bad, "ok": not bad,
                               "method": "int64 checksums of pair 0's (u, v) bit patterns per rank vs rank 0's recomputation of that rank's seed"}

    # Dominant kernel (fused level-0 LK) timed with HIP events on the launch stream, in a
    # second pass of the same K steps (events around every level launch; library hook).
    roofline = None
    if not args.no_profile_pass and args.mode == "pairs":
        # The throughput pass above runs the library default (the batch split into two stream
        # groups whose launches overlap).  The kernel roofline is taken with ONE group, so the
        # level-0 launch covers the whole batch and has the GPU to itself while it is timed.
        ctx.set_lk_groups(1)
        if args.preroll_s > 0:  # the device idled during the checks above: same pre-roll as the timed region
            t_end = time.perf_counter() + args.preroll_s
            while time.perf_counter() < t_end:
                for _ in range(8):
                    lk.calcOpticalFlowPyrBatch(prev, nxt, WIN, LEVELS, ctx=ctx, out=(u, v), stream=stream)
                torch.cuda.synchronize()
        ctx.profile(True)
        ctx.profile_reset()
        torch.cuda.synchronize()
        for _ in range(args.steps):  # one pass at a time on one context: nothing runs beside the timed launch
            lk.calcOpticalFlowPyrBatch(prev, nxt, WIN, LEVELS, ctx=ctx, out=(u, v), stream=stream)
        torch.cuda.synchronize()
        ctx.set_lk_groups(args.lk_groups)
        lvl_ms = []
        for l in range(LEVELS):
            ms, n = ctx.profile_lk_level(l)
            lvl_ms.append(ms / max(n, 1))
        ctx.profile(False)
        ctx.profile_reset()
        pairs_per_launch = ctx.profile_lk_pairs() or B  # the library splits the batch into stream groups
        k_bytes = level0_kernel_bytes_pair(ROWS, COLS, LEVELS) * pairs_per_launch
        achieved = k_bytes / (lvl_ms[0] * 1e-3) / 1e9
        traffic, traffic_source, kernel_name = None, None, ctx.lk_level_kernel_name(WIN, ROWS, COLS, pairs_per_launch)
        live = None
        if rank == 0 and n_gpus == 1 and not args.no_pmc:
            live = measure_traffic_live(pairs_per_launch, args.opt)
        tpath = os.path.join(ROOT, "profiles", "traffic.json")  # from rocprofv3 --pmc runs
        if live is not None:
            traffic = live["level0_hbm_bytes_per_launch"]
            kernel_name = live["kernel"].replace("micv::", "")
            traffic_source = live["method"]
        elif os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("pairs_per_launch") == pairs_per_launch:
                    traffic = tj.get("level0_hbm_bytes_per_launch")
                    kernel_name = tj.get("kernel", kernel_name).replace("micv::", "")
                    traffic_source = (f"profiles/traffic.json ({tj.get('profile', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE')}"
                                      ", separate passes, gfx950 x2 FETCH correction; not measured in this run)")
            except Exception:
                traffic = None
        roofline = {
            "bound": "hbm", "kernel": kernel_name + " (pyramid level 0)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "frac_of_measured_copy_bw": achieved / HBM_COPY_GBS, "measured_copy_bw": HBM_COPY_GBS,
            "traffic": traffic, "traffic_source": traffic_source,
            "bytes_per_launch": k_bytes, "pairs_per_launch": pairs_per_launch, "avg_launch_ms": lvl_ms[0],
            "level_ms": lvl_ms,
            "valu": valu_roofline(lvl_ms[0], pairs_per_launch, live, live["method"] if live else None),
            "traffic_detail": live,
            "binding_axis": "valu",
            "note": "HBM figures as the contract asks; the kernel itself is f32-VALU-issue bound "
                    "(5 x 15-tap separable window sums) -- see `valu`.  Timed in a second pass of the "
                    "same steps with the batch in one stream group (whole batch per launch, no "
                    "concurrent launches)",
        }

    # every lane (context / stream / output buffers) ran the same inputs: their last outputs must be the same bits
    lanes_identical = None
    if args.mode == "pairs" and max(1, args.inflight) > 1:
        for c_, st_, out_ in lanes:
            lk.calcOpticalFlowPyrBatch(prev, nxt, WIN, LEVELS, ctx=c_, out=out_, stream=st_.cuda_stream)
        torch.cuda.synchronize()
        lanes_identical = all(bool(torch.equal(o[0], u)) and bool(torch.equal(o[1], v)) for _, _, o in lanes[1:])

    secondary = None
    if rank == 0 and n_gpus == 1 and args.mode == "pairs" and not args.no_secondary:
        try:
            secondary = secondary_block(ctx, stream, torch, np)
        except Exception as e:  # the headline line must survive a failure here; the failure is reported, not hidden
            secondary = {"error": f"{type(e).__name__}: {e}"}

    cpu, parity = None, None
    if rank == 0 and n_gpus == 1 and args.cpu_pairs > 0:
        def hip_uv(i):
            if args.mode != "pairs" or i >= B:
                return None
            return u[i].cpu().numpy(), v[i].cpu().numpy()
        cpu, parity = cpu_baseline(args.cpu_pairs, hip_uv)

    if rank == 0:
        total_px = (n_gpus if args.mode == "pairs" else 1) * B * args.steps * ROWS * COLS
        value = total_px / dt / 1e6
        out = {
            "metric": "Mpix/s (LK 5-level pyramid, 1080p pairs)",
            "value": value,
            "unit": "Mpix/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if args.mode == "pairs" else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"C2 x{B}: {B} x 1920x1080 synthetic translated pairs per GPU per step "
                            f"(C4 per-GPU share), {LEVELS}-level pyramid, win {WIN}, device-resident",
                "pairs_per_gpu_per_step": B, "levels": LEVELS, "win": WIN,
                "parallelism": f"pair-dp{n_gpus}" if args.mode == "pairs" else
                               (f"row-shard{n_gpus} (coarse-flow halo, ncclSend/ncclRecv inside micv_lk_flow_pyr_rowshard_dev)"
                                if args.next_margin is None else f"row-shard{n_gpus} (coarse-flow halo, torch p2p, declared next margin)"),
                "mode": args.mode,
                "rccl_ranks": dist.get_world_size() if dist is not None else 1,
                "comm_selftest": comm_selftest,
                "per_rank_parity": per_rank_parity,
                "per_rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank],
                "flow_check": {"median_u": um, "median_v": vm, "ok": ok},
                "parity_1080p": None if parity is None else parity["bit_exact"],
                "parity_1080p_detail": parity,
                "lanes_identical": lanes_identical,
                "passes_in_flight": max(1, args.inflight) if args.mode == "pairs" else 1,
                "preroll_s": args.preroll_s,
                "one_pass_at_a_time_ms_per_step": serial_ms,
                "single_pair_ms": single_pair_ms,
                "single_pair_Mpix_s": None if not single_pair_ms else ROWS * COLS / single_pair_ms / 1e3,
                "next_margin": args.next_margin if args.mode == "rowshard" else None,
                "next_margin_sufficed_frac": (margin_stats[1] / max(1, margin_stats[0])) if args.mode == "rowshard" and args.next_margin is not None else None,
                "host_pair_ms": host_pair_ms,
                "host_pair_note": "micv_lk_flow_pyr_host, one 1080p pair per call: 33.2 MB over PCIe (pageable "
                                  "= pinned = 53 GB/s here: 0.62 ms) + the device call; PCIe-inclusive, not `value`",
                "host_sequence_ms_per_pair": host_seq_ms,
                "host_sequence_u8_ms_per_pair": host_seq8_ms,
                "host_sequence_note": "micv_lk_flow_seq_host, 16 x 1080p frames = 15 pairs per call: one upload per frame "
                                      "(f32: 8.3 MB, 8-bit: 2.1 MB + device conversion) and one download per pair (16.6 MB) beside "
                                      "the chains; copies from / to pageable memory do not overlap each other on this platform "
                                      "(0.157 + 0.31 ms per f32 pair is the floor; profiles/r06/host_sequence.md); "
                                      "pair 4 byte-identical to the per-pair call, f32 and 8-bit: " + str(host_seq_same),
            },
            "sustained": sustained,
            "algorithmic_GBps_pipeline": value * 1e6 * algorithmic_bytes_pair(ROWS, COLS, LEVELS)
                                         / (ROWS * COLS) / 1e9,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "secondary": secondary,
        }
        if args.mode == "rowshard":
            out["config"]["note"] = ("row-shard mode: every pair split by rows over the ranks (north_star's halo exchange); NOT the "
                                     "SCALE axis -- it costs ~2.5x the GPU time of the unsharded batch and replicates the `next` "
                                     "pyramid on every rank (DESIGN.md section 7), so strong-scaling efficiency is <= 0.4 by construction")
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
