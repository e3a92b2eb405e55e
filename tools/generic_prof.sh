#!/usr/bin/env bash
# Per-kernel times of the generic (multi-launch) LK path at window 43, 1080p.
repo="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
out="$repo/gpurun_out/genprof"; rm -rf "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$repo/tools/generic_prof.py" > "$out.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2))
PY
