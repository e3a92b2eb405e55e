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
