#!/usr/bin/env bash
# SQ_INSTS_VALU / SQ_BUSY of the level-0 launch for a set of bench flags.  usage: bash tools/pmc_quick.sh "<flags>" ...
repo="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
i=0
for flags in "$@"; do
  i=$((i+1)); out="$repo/gpurun_out/pmcq_$i"; rm -rf "$out"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d "$out" -- \
      python3 "$repo/bench.py" --cpu-pairs 0 --steps 3 --warmup 1 --no-profile-pass --inflight 1 --lk-groups 1 --sustained-s 0 $flags > /dev/null 2>&1
  python3 - "$out" "$flags" <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lk_level" in r["Kernel_Name"]:
            d[(r["Kernel_Name"].split("(")[0][-40:], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
k = max(d, key=lambda t: sum(d[t]["SQ_INSTS_VALU"]) / len(d[t]["SQ_INSTS_VALU"]))
print(sys.argv[2], k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in d[k].items()})
PY
done
