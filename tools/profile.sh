#!/usr/bin/env bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py.
# Usage: bash tools/profile.sh <tag>   -> writes gpurun_out/prof_<tag>/...
# --pmc passes are separate runs with no tracing flags (pool rule), one counter group each.
set -uo pipefail
tag="${1:-r01}"
repo="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
out="$repo/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
# --no-pmc --no-secondary: no nested rocprofv3 children and no extra kernels inside a profiled run
BENCH=(python3 "$repo/bench.py" --cpu-pairs 0 --no-pmc --no-secondary)

rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- \
    "${BENCH[@]}" --steps 20 --warmup 5 --sustained-s 0 > "$out/bench_trace.log" 2>&1
echo "trace rc=$?"
# The same steps one pass at a time in one stream group: the level-0 launch alone on the GPU, the
# duration bench.py's roofline line is about (with two passes in flight the traced launches overlap).
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_alone" -- \
    "${BENCH[@]}" --steps 20 --warmup 5 --sustained-s 0 --inflight 1 --lk-groups 1 --no-profile-pass > "$out/bench_trace_alone.log" 2>&1
echo "trace_alone rc=$?"

# PMC passes: one stream group (--lk-groups 1), so each level-0 dispatch covers the whole batch (the
# launch the roofline line of bench.py is about); counters serialise dispatches anyway.
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$out/pmc$i" -- \
      "${BENCH[@]}" --steps 3 --warmup 1 --no-profile-pass --lk-groups 1 --inflight 1 --sustained-s 0 > "$out/bench_pmc$i.log" 2>&1
  echo "pmc$i ($grp) rc=$?"
done
python3 "$repo/tools/summarize_profile.py" "$out" > "$out/summary.txt" 2>&1
cat "$out/summary.txt" | head -80
# keep the merge small: drop raw per-dispatch traces, keep stats + pmc csv
find "$out" -name '*kernel_trace.csv' -size +4M -delete 2>/dev/null
du -sh "$out"
