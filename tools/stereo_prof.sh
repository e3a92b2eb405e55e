#!/usr/bin/env bash
# Runs on the GPU box (via gpurun): per-kernel times and counters of the C3 stereo call (tools/stereo_bench.py).
# Usage: bash tools/stereo_prof.sh <tag> -> gpurun_out/stereo_<tag>/{stats.txt,pmc.txt}
set -uo pipefail
tag="${1:-r06}"
repo="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
out="$repo/gpurun_out/stereo_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$repo/tools/stereo_bench.py" > "$out/bench.log" 2>&1
echo "trace rc=$?"
f=$(find "$out/trace" -name '*kernel_stats.csv' | head -1)
python3 - "$f" > "$out/stats.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "stereo" in n:
        print(f'{float(r["AverageNs"])/1000:9.2f} us avg  x{r["Calls"]:>5}  {n[:150]}')
PY
cat "$out/stats.txt"
if [[ "${2:-}" == "pmc" ]]; then
  i=0
  for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
             "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d "$out/pmc$i" -- python3 "$repo/tools/stereo_bench.py" > "$out/pmc$i.log" 2>&1
    echo "pmc$i rc=$?"
  done
  python3 - "$out" > "$out/pmc.txt" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "stereo" in n:
            acc[n[:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
  cat "$out/pmc.txt"
fi
