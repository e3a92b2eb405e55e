#!/usr/bin/env python3
"""Median device time per (kernel, grid size) of a rocprofv3 kernel trace CSV -- separates the launches of one kernel
on different pyramid levels, which `--stats` averages together.  python tools/probes/trace_levels.py <dir-or-csv> [substr]"""
import csv, collections, glob, os, sys
p = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
files = [p] if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True)
d = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            d[(r["Kernel_Name"].split("(")[0][-44:], int(r.get("Grid_Size_X", r.get("Grid_Size", 0))))].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    v.sort()
    print(f"{k[0]:44s} grid={k[1]:8d} n={len(v):4d} median_us={v[len(v)//2]/1e3:8.2f} min_us={v[0]/1e3:8.2f}")
