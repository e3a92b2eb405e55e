#!/usr/bin/env python3
"""bench.py -- Mpix/s through 5-level pyramidal Lucas-Kanade on 1080p frame pairs.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (lk::calcOpticalFlowPyr, 5 levels, win 15) over one
batch of --pairs synthetic 1080p frame pairs that already sit in HBM.  Frame pairs are
independent, so N GPUs = N ranks each with its own batch (weak scaling, no data-path
collective).  Rank 0 prints ONE JSON line.  See DESIGN.md "Measurement" for the byte model.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROWS, COLS, WIN, LEVELS = 1080, 1920, 15, 5
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


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


def cpu_baseline(sample_pairs):
    """The CPU oracle (a plain-C port of the reference algorithm, single thread -- the
    reference's own loops are single-threaded, OpticalFlow.cpp:85-103) timed on this host."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as orc
    from introtocomputervision_amd import synth
    pairs = [synth.lk_pair(0x5EED0005 + i, ROWS, COLS, 3, -2) for i in range(sample_pairs)]
    t0 = time.perf_counter()
    for p, n in pairs:
        orc.lk_flow_pyr(p, n, WIN, LEVELS)
    dt = time.perf_counter() - t0
    out = {
        "value": sample_pairs * ROWS * COLS / dt / 1e6,
        "unit": "Mpix/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{sample_pairs} of the same 1080p pairs, 5 levels, win 15, oracle/liboracle.so, {dt:.1f} s",
    }
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
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pairs", type=int, default=8, help="frame pairs per GPU per step")
    ap.add_argument("--cpu-pairs", type=int, default=6, help="pairs timed on the CPU baseline (0 = skip)")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--inflight", type=int, default=2, help="passes in flight (contexts/streams), pairs mode")
    ap.add_argument("--mode", choices=["pairs", "rowshard"], default="pairs",
                    help="pairs: every rank owns whole frame pairs (default, weak scaling, no data-path "
                         "collective); rowshard: every pair is split by rows over all ranks with a "
                         "coarse-flow halo exchange per level (strong scaling, RCCL point-to-point)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from introtocomputervision_amd import lk, synth
    from introtocomputervision_amd._capi import Context

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
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}; using {n_gpus}", file=sys.stderr)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    # Synthetic C2/C4 frame pairs, distinct per rank, resident in HBM.
    B = args.pairs
    prev_h = np.empty((B, ROWS, COLS), np.float32)
    next_h = np.empty((B, ROWS, COLS), np.float32)
    for i in range(B):
        prev_h[i], next_h[i] = synth.lk_pair(0x5EED0005 + rank * B + i, ROWS, COLS, 3, -2)
    prev = torch.from_numpy(prev_h).to(dev)
    nxt = torch.from_numpy(next_h).to(dev)
    u = torch.empty_like(prev)
    v = torch.empty_like(prev)
    ctx = Context(local_rank)
    stream = torch.cuda.current_stream(dev).cuda_stream

    if args.mode == "rowshard":
        # All ranks hold the same pairs (replicated inputs); each computes its row band of every pair.
        from introtocomputervision_amd import pyr, shard
        if dist is not None:
            for t in (prev, nxt):
                dist.broadcast(t, src=0)
        plan = shard.RowShardPlan(ROWS, COLS, LEVELS, n_gpus)
        level_fn = shard.gpu_level_fn(ctx, WIN)

        class _NoComm:
            def exchange(self, *a):
                pass
        comm = shard.DistComm(rank, n_gpus) if dist is not None else _NoComm()
        a0, b0 = plan.band(0, rank)

        def step():
            for i in range(B):
                pp = pyr.makeGaussianPyramid(prev[i], LEVELS, ctx=ctx)
                npyr = pyr.makeGaussianPyramid(nxt[i], LEVELS, ctx=ctx)
                bu, bv = shard.lk_pyr_band(pp, npyr, plan, rank, WIN, level_fn, comm)
                u[i, a0:b0] = bu[a0:b0]
                v[i, a0:b0] = bv[a0:b0]
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
        counter = [0]

        def step():
            c, st, out = lanes[counter[0] % F]
            counter[0] += 1
            lk.calcOpticalFlowPyrBatch(prev, nxt, WIN, LEVELS, ctx=c, out=out, stream=st.cuda_stream)

    def barrier():
        if dist is not None:
            dist.barrier()

    ctx.warmup(stream)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the same K steps one at a time on one context (no overlap between passes): reported beside
    # `value` so the effect of keeping two passes in flight is visible
    serial_ms = None
    if args.mode == "pairs" and max(1, args.inflight) > 1:
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(args.steps):
            lk.calcOpticalFlowPyrBatch(prev, nxt, WIN, LEVELS, ctx=ctx, out=(u, v), stream=stream)
        torch.cuda.synchronize()
        serial_ms = (time.perf_counter() - ts) / args.steps * 1e3

    # sanity of what was measured: known translation comes back (not part of the timing)
    if args.mode == "rowshard":
        chk_u, chk_v = u[0, a0 + 8:b0 - 8, 64:-64], v[0, a0 + 8:b0 - 8, 64:-64]
    else:
        chk_u, chk_v = u[0, 64:-64, 64:-64], v[0, 64:-64, 64:-64]
    um = float(chk_u.median())
    vm = float(chk_v.median())
    ok = abs(um - 3.0) < 0.25 and abs(vm + 2.0) < 0.25

    # Dominant kernel (fused level-0 LK) timed with HIP events on the launch stream, in a
    # second pass of the same K steps (events around every level launch; library hook).
    roofline = None
    if not args.no_profile_pass and args.mode == "pairs":
        # The throughput pass above runs the library default (the batch split into two stream
        # groups whose launches overlap).  The kernel roofline is taken with ONE group, so the
        # level-0 launch covers the whole batch and has the GPU to itself while it is timed.
        saved_groups = os.environ.get("MICV_LK_GROUPS")
        os.environ["MICV_LK_GROUPS"] = "1"
        ctx.profile(True)
        ctx.profile_reset()
        torch.cuda.synchronize()
        for _ in range(args.steps):  # one pass at a time on one context: nothing runs beside the timed launch
            lk.calcOpticalFlowPyrBatch(prev, nxt, WIN, LEVELS, ctx=ctx, out=(u, v), stream=stream)
        torch.cuda.synchronize()
        if saved_groups is None:
            del os.environ["MICV_LK_GROUPS"]
        else:
            os.environ["MICV_LK_GROUPS"] = saved_groups
        lvl_ms = []
        for l in range(LEVELS):
            ms, n = ctx.profile_lk_level(l)
            lvl_ms.append(ms / max(n, 1))
        ctx.profile(False)
        ctx.profile_reset()
        pairs_per_launch = ctx.profile_lk_pairs() or B  # the library splits the batch into stream groups
        k_bytes = level0_kernel_bytes_pair(ROWS, COLS, LEVELS) * pairs_per_launch
        achieved = k_bytes / (lvl_ms[0] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")  # from rocprofv3 --pmc runs
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("pairs_per_launch") == pairs_per_launch:
                    traffic = tj.get("level0_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {
            "bound": "hbm", "kernel": "lk_level_kernel<7,COARSE> (pyramid level 0)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "bytes_per_launch": k_bytes, "pairs_per_launch": pairs_per_launch, "avg_launch_ms": lvl_ms[0],
            "level_ms": lvl_ms,
            "note": "kernel is f32-VALU/LDS-bound (5 x 15-tap separable window sums), not HBM-bound; "
                    "timed in a second pass of the same steps with the batch in one stream group "
                    "(whole batch per launch, no concurrent launches)",
        }

    cpu = None
    if rank == 0 and n_gpus == 1 and args.cpu_pairs > 0:
        cpu = cpu_baseline(args.cpu_pairs)

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
                "parallelism": f"pair-dp{n_gpus}" if args.mode == "pairs" else f"row-shard{n_gpus} (coarse-flow halo, p2p)",
                "flow_check": {"median_u": um, "median_v": vm, "ok": ok},
                "passes_in_flight": max(1, args.inflight) if args.mode == "pairs" else 1,
                "one_pass_at_a_time_ms_per_step": serial_ms,
            },
            "algorithmic_GBps_pipeline": value * 1e6 * algorithmic_bytes_pair(ROWS, COLS, LEVELS)
                                         / (ROWS * COLS) / 1e9,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
