"""Difference the cumulative MICV_LK_STOP=k runs of tools/phase_pmc.sh into per-phase counters of the
level-0 launch of lk_level_kernel<7,1,*> (largest grid)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
order = ["0", "2", "3", "41", "42", "43", "4", "-1"]
names = {"0": "stage", "2": "pyrUp+warp", "3": "gradients", "41": "sweep A rows", "42": "sweep A cols",
         "43": "sweep B rows", "4": "sweep B cols", "-1": "solve+store"}
cum = {}
for k in order:
    acc = defaultdict(lambda: [0.0, 0])
    grid = 0
    rows = []
    for f in glob.glob(os.path.join(out, f"pmc_{k}", "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows = [r for r in rows if "lk_level_kernel<7, 1" in r["Kernel_Name"]]
    if not rows:
        continue
    grid = max(int(r["Grid_Size"]) for r in rows)
    for r in rows:
        if int(r["Grid_Size"]) == grid:
            c = acc[r["Counter_Name"]]
            c[0] += float(r["Counter_Value"])
            c[1] += 1
    cum[k] = {c: v[0] / v[1] for c, v in acc.items()}
    us = []
    for f in glob.glob(os.path.join(out, f"trace_{k}", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "lk_level_kernel<7, 1" in r["Kernel_Name"]:
                g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
                us.append((g, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    if us:
        gm = max(g for g, _ in us)
        d = [t for g, t in us if g == gm]
        cum[k]["duration_us"] = sum(d) / len(d)
cols = ["duration_us", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT",
        "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"]
print(f"{'phase':16s}" + "".join(f"{c[3:] if c.startswith('SQ_') else c:>20s}" for c in cols))
prev = defaultdict(float)
for k in order:
    if k not in cum:
        continue
    print(f"{names[k]:16s}" + "".join(f"{cum[k].get(c, 0) - prev[c]:20.1f}" for c in cols))
    for c in cols:
        prev[c] = cum[k].get(c, 0)
print(f"{'total':16s}" + "".join(f"{prev[c]:20.1f}" for c in cols))
