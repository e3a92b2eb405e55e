#!/usr/bin/env bash
# HBM counters (FETCH_SIZE, WRITE_SIZE: separate passes, KiB) + VALU instructions of every LK kernel of the split forms
# (MICV_OPT_LK_SPLIT = 1, 2, 3) and of the fused launch (-1), 8 x 1080p, per (kernel, grid).  GPU box.
repo="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
for m in 0 2 1 3; do
  for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    out="$repo/gpurun_out/splitpmc_${m}_$c"; rm -rf "$out"
    rocprofv3 --pmc $c --output-format csv -d "$out" -- python3 "$repo/tools/probes/split_trace.py" $m 8 4 > "$out.log" 2>&1
  done
  python3 - "$repo/gpurun_out" "$m" <<'PY'
import csv, glob, sys, collections, json
root, m = sys.argv[1], sys.argv[2]
d = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{root}/splitpmc_{m}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "lk_" in r["Kernel_Name"] and r["Counter_Name"] == c:
                acc[(r["Kernel_Name"].split("(")[0][-40:], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        d[k][c] = sum(v) / len(v)
for k in sorted(d):
    e = d[k]
    f, w = e.get("FETCH_SIZE", 0) * 1024, e.get("WRITE_SIZE", 0) * 1024
    print(json.dumps({"split": int(m), "kernel": k[0], "grid": k[1], "fetch_MB_raw": round(f / 1e6, 1), "fetch_MB_x2": round(2 * f / 1e6, 1),
                      "write_MB": round(w / 1e6, 1), "valu_M": round(e.get("SQ_INSTS_VALU", 0) / 1e6, 2)}))
PY
done
