#!/usr/bin/env bash
# A/B on ONE box: the same bench against two builds of the library (MICV_LIB selects the .so).
# usage: bash tools/ab_bench.sh <base.so> <new.so> [rounds]
repo="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
base="$1"; new="$2"; rounds="${3:-3}"
for r in $(seq 1 "$rounds"); do
  for lib in "$base" "$new"; do
    MICV_LIB="$repo/introtocomputervision_amd/$lib" python "$repo/tools/chain_bench.py" 2>&1 | grep '"groups": 1, "max_chain": 1\|"groups": 2, "max_chain": 0' | sed "s/^/$lib r$r /"
  done
done
