#!/usr/bin/env bash
# Device time of every kernel a python script launches: rocprofv3 kernel trace + stats, printed as a table.
#   bash tools/trace_script.sh tools/probes/a456_trace.py [script args ...]      (GPU box)
repo="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
script="$(realpath "$1")"; shift
mkdir -p "$repo/gpurun_out"
out="$repo/gpurun_out/trace_$(basename "$script" .py)"; rm -rf "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$script" "$@" > "$out.log" 2>&1
echo "rocprofv3 rc=$? (log: $out.log)"
f=$(ls "$out"/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -z "$f" ] && { echo "no kernel_stats.csv under $out"; tail -5 "$out.log"; exit 1; }
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print(f'{r["Name"][:100]:100s} calls={int(r["Calls"]):5d} avg_us={float(r["AverageNs"])/1e3:9.2f}')
PY
[ -n "$KEEP_TRACE" ] || find "$out" -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
