#!/usr/bin/env bash
# Attribute PMC counters of the fused LK level kernel to its phases: the same bench is run with
# MICV_LK_STOP=k (leave the kernel after phase k; results are garbage, only counters matter) and
# the cumulative counters are differenced by tools/phase_pmc_summary.py.
# Usage (GPU box): bash tools/phase_pmc.sh <tag>  -> gpurun_out/phase_pmc_<tag>/
set -uo pipefail
tag="${1:-r01}"
repo="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
out="$repo/gpurun_out/phase_pmc_$tag"
mkdir -p "$out"
# the stop-after-phase path exists only in the diagnostic flavour of the library
MICV_OUT=libmicv_diag.so EXTRA_HIPCC_FLAGS=-DMICV_DIAG bash "$repo/introtocomputervision_amd/csrc/build.sh" || exit 1
export MICV_LIB="$repo/introtocomputervision_amd/libmicv_diag.so"
cd /tmp && export TMPDIR=/tmp
for stop in 0 2 3 41 42 43 4 -1; do
  export MICV_LK_STOP=$stop
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS \
      --output-format csv -d "$out/pmc_$stop" -- \
      python3 "$repo/bench.py" --cpu-pairs 0 --no-pmc --no-secondary --steps 2 --warmup 1 --no-profile-pass --inflight 1 --lk-groups 1 --sustained-s 0 > "$out/bench_pmc_$stop.log" 2>&1
  echo "stop=$stop pmc rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$stop" -- \
      python3 "$repo/bench.py" --cpu-pairs 0 --no-pmc --no-secondary --steps 10 --warmup 3 --no-profile-pass --inflight 1 --lk-groups 1 --sustained-s 0 > "$out/bench_trace_$stop.log" 2>&1
  echo "stop=$stop trace rc=$?"
done
python3 "$repo/tools/phase_pmc_summary.py" "$out" | tee "$out/summary.txt"
find "$out" -name '*kernel_trace.csv' -size +2M -delete 2>/dev/null
