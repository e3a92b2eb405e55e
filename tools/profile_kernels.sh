#!/usr/bin/env bash
# rocprofv3 kernel-trace stats of tools/bench_kernels.py (the non-LK kernels on their BASELINE
# configs).  Usage (GPU box): bash tools/profile_kernels.sh <tag> -> gpurun_out/kprof_<tag>/
set -uo pipefail
tag="${1:-r01}"
repo="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
out="$repo/gpurun_out/kprof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- \
    python3 "$repo/tools/bench_kernels.py" > "$out/bench_kernels.log" 2>&1
echo "rc=$?"
f=$(ls "$out"/trace/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY' | tee "$out/kernel_stats.txt"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print(f'{r["Name"][:90]:90s} calls={int(r["Calls"]):5d} avg_us={float(r["AverageNs"])/1e3:9.2f} total_ms={float(r["TotalDurationNs"])/1e6:8.3f}')
PY
find "$out" -name '*kernel_trace.csv' -size +2M -delete 2>/dev/null

# HBM traffic of the same kernels: FETCH_SIZE / WRITE_SIZE in their own passes (no tracing flags)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/pmc_$c" -- \
      python3 "$repo/tools/bench_kernels.py" > "$out/bench_pmc_$c.log" 2>&1
  echo "pmc $c rc=$?"
done
python3 - "$out" <<'PY' | tee "$out/traffic.txt"
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
def short(n): return n.split("(")[0].replace("void ", "")[:60]
dur = defaultdict(list)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        dur[(short(r["Kernel_Name"]), g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
pmc = defaultdict(lambda: defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            pmc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][c].append(float(r["Counter_Value"]))
# FETCH_SIZE is raw: the gfx950 x2 correction (MI355X_MICROARCH.md) applies to 16-B-per-lane loads only
# (the LK level kernel's LDS-DMA); dword-load kernels read back their algorithmic bytes uncorrected
# (fused Sobel 4K: 35.0 MB raw for a 33.2 MB image + halo).
print(f'{"kernel (largest grid)":62s} {"us":>8s} {"fetch MB":>10s} {"write MB":>9s} {"GB/s raw":>9s} {"% of 8 TB/s":>11s}')
best = {}
for (k, g) in dur:
    if k not in best or g > best[k]: best[k] = g
for k, g in sorted(best.items(), key=lambda kv: -sum(dur[(kv[0], kv[1])])):
    if not k.startswith("micv::"): continue
    us = sum(dur[(k, g)]) / len(dur[(k, g)])
    f = pmc[(k, g)]["FETCH_SIZE"]; w = pmc[(k, g)]["WRITE_SIZE"]
    if not f or not w: continue
    fm = sum(f) / len(f) * 1024 / 1e6; wm = sum(w) / len(w) * 1024 / 1e6
    gbs = (fm + wm) / us * 1e3
    print(f"{k:62s} {us:8.1f} {fm:10.2f} {wm:9.2f} {gbs:9.0f} {gbs / 80:11.1f}")
PY
