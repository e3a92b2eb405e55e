#!/usr/bin/env bash
# Issue / stall counters (one rocprofv3 --pmc pass, 8 SQ counters) of every kernel a python script launches:
#   bash tools/pmc_script.sh tools/probes/win43_trace.py [kernel-name-substring [script args ...]]
repo="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
script="$(realpath "$1")"; filt="${2:-}"; shift; [ $# -gt 0 ] && shift
mkdir -p "$repo/gpurun_out"
out="$repo/gpurun_out/pmcs"; rm -rf "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES \
    --output-format csv -d "$out" -- python3 "$script" "$@" > "$out.log" 2>&1
rc=$?
echo "rocprofv3 rc=$rc (log: $out.log)"
if ! find "$out" -name '*counter_collection.csv' 2>/dev/null | grep -q .; then
  echo "no counter_collection.csv under $out: the pass failed or launched no kernels"; tail -5 "$out.log"; exit 1
fi
python3 - "$out" "$filt" <<'PY'
import csv, glob, sys, collections, json
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            d[(r["Kernel_Name"].split("(")[0][-48:], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(d):
    print(json.dumps({"kernel": k[0], "grid": k[1], "n": len(next(iter(d[k].values()))),
                      "counters_M": {c: round(sum(v) / len(v) / 1e6, 3) for c, v in sorted(d[k].items())}}))
PY
