#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory.

 * per-kernel stats from `rocprofv3 --kernel-trace --stats` (kernel_stats.csv)
 * per (kernel, grid) mean duration from kernel_trace.csv -- the same lk_level_kernel runs
   once per pyramid level, the largest grid is level 0 (the roofline kernel)
 * PMC means per (kernel, grid); HBM traffic of the level-0 launch with the gfx950
   correction of MI355X_MICROARCH.md §HBM (FETCH_SIZE counts 64 B per 128-B request for
   wide coalesced reads -> reported both raw and x2; FETCH_SIZE/WRITE_SIZE are in KiB).
Writes summary.json (+ traffic.json when both FETCH_SIZE and WRITE_SIZE are present).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
res = {}


def short(name):
    return name.split("(")[0].replace("void ", "")[:48]


for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:10]:
        print("  {:48s} calls={:>6} total_ms={:>10.3f} avg_us={:>10.2f} pct={}".format(
            short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
            float(r["AverageNs"]) / 1e3, r["Percentage"]))
    res["kernel_stats"] = rows[:16]

# "trace": bench.py as the driver runs it (two passes in flight: traced launches overlap each other);
# "trace_alone": the same steps one pass at a time, one stream group -- the level-0 launch alone on the
# GPU, the duration bench.py's roofline line quotes.
for sub, key, title in (("trace", "durations_us", "two passes in flight"),
                        ("trace_alone", "durations_alone_us", "one pass at a time: launches alone on the GPU")):
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(out, sub, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "Grid_Size_X" in r:  # total work-items = X * Y * Z (batch index is grid.y)
                g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            else:
                g = int(r.get("Grid_Size", 0) or 0)
            dur[(short(r["Kernel_Name"]), g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if dur:
        print(f"== mean duration per (kernel, grid threads), {title}")
        res[key] = {}
        for (k, g), v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            if "micv" not in k:
                continue
            v2 = v[len(v) // 4:]  # drop warm-up quarter
            m = sum(v2) / len(v2) / 1e3
            res[key][f"{k}|{g}"] = {"mean_us": m, "n": len(v2)}
            print(f"  {k:48s} grid={g:>9d} n={len(v2):4d} mean_us={m:10.2f}")

pmc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
        c = pmc[k][r["Counter_Name"]]
        c[0] += float(r["Counter_Value"])
        c[1] += 1
print("== PMC (mean per dispatch) for lk_level kernels")
summ = {}
for (k, g), d in sorted(pmc.items(), key=lambda kv: -kv[0][1]):
    summ[f"{k}|{g}"] = {c: v[0] / max(v[1], 1) for c, v in d.items()}
    if "lk_level" not in k:
        continue
    print(f"  {k} grid={g}")
    for c, v in sorted(d.items()):
        print(f"      {c:26s} {v[0] / max(v[1], 1):18.1f}  (n={v[1]})")
res["pmc_mean_per_dispatch"] = summ

# the level-0 launch = the lk_level* dispatch that writes the most (the plain kernel or the chain kernel)
lvl0 = [kg for kg in pmc if "lk_level" in kg[0] and "WRITE_SIZE" in pmc[kg]]
if lvl0:
    kg = max(lvl0, key=lambda t: pmc[t]["WRITE_SIZE"][0] / pmc[t]["WRITE_SIZE"][1])
    d = pmc[kg]
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        fetch = d["FETCH_SIZE"][0] / d["FETCH_SIZE"][1] * 1024
        write = d["WRITE_SIZE"][0] / d["WRITE_SIZE"][1] * 1024
        t = {"kernel": kg[0], "grid_threads": kg[1], "pairs_per_launch": pairs,
             "fetch_bytes_raw": fetch, "fetch_bytes_x2_gfx950": 2 * fetch, "write_bytes": write,
             "level0_hbm_bytes_per_launch": 2 * fetch + write,
             "valu_insts_per_launch": (d["SQ_INSTS_VALU"][0] / d["SQ_INSTS_VALU"][1]) if "SQ_INSTS_VALU" in d else None,
             "profile": os.path.basename(os.path.normpath(out)),
             "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KiB -> bytes; "
                       "FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (gfx950 counts 64 B of each "
                       "128-B request); mean over the level-0 dispatches"}
        json.dump(t, open(os.path.join(out, "traffic.json"), "w"), indent=1)
        print("== level-0 HBM traffic per launch:", json.dumps(t))
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
