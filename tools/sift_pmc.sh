#!/usr/bin/env bash
# Counters of sift_descriptor_kernel, one rocprofv3 --pmc pass per group.  usage: bash tools/sift_pmc.sh
repo="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp REPS=3
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1)); out="$repo/gpurun_out/siftpmc_$i"; rm -rf "$out"
  rocprofv3 --pmc $grp --output-format csv -d "$out" -- python3 "$repo/tools/sift_bench.py" > /dev/null 2>&1
  python3 - "$out" <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sift_descriptor" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({c: round(sum(v) / len(v) / 1e6, 3) for c, v in d.items()})
PY
done
