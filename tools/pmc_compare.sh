#!/usr/bin/env bash
# Stall / issue counters of the level-0 launch for several option sets, one rocprofv3 --pmc run each
# (8 SQ counters = one pass):  bash tools/pmc_compare.sh <outfile> "<bench flags>" ["<bench flags>" ...]
repo="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
outfile="$(realpath -m "$1")"; shift
cd /tmp && export TMPDIR=/tmp
i=0
: > "$outfile"
for flags in "$@"; do
  i=$((i+1)); out="$repo/gpurun_out/pmcc_$i"; rm -rf "$out"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d "$out" -- \
      python3 "$repo/bench.py" --cpu-pairs 0 --no-pmc --no-secondary --steps 3 --warmup 1 --no-profile-pass --inflight 1 --lk-groups 1 --sustained-s 0 --preroll-s 0 $flags > /dev/null 2>&1
  python3 - "$out" "$flags" <<'PY' | tee -a "$outfile"
import csv, glob, sys, collections, json
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lk_level" in r["Kernel_Name"]:
            d[(r["Kernel_Name"].split("(")[0][-44:], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
if not d:
    print(sys.argv[2], "no lk_level dispatches found"); sys.exit(0)
k = max(d, key=lambda t: sum(d[t]["SQ_INSTS_VALU"]) / max(1, len(d[t]["SQ_INSTS_VALU"])))
print(json.dumps({"flags": sys.argv[2], "kernel": k[0], "grid": k[1],
                  "counters_M": {c: round(sum(v) / len(v) / 1e6, 2) for c, v in sorted(d[k].items())}}))
PY
done
