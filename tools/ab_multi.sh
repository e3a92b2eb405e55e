#!/usr/bin/env bash
# A/B/C... on ONE box: tools/chain_bench.py against several builds of the library.
# usage: bash tools/ab_multi.sh <rounds> <lib.so> [<lib.so> ...]   (names inside introtocomputervision_amd/)
repo="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
rounds="$1"; shift
for r in $(seq 1 "$rounds"); do
  for lib in "$@"; do
    MICV_LIB="$repo/introtocomputervision_amd/$lib" python "$repo/tools/chain_bench.py" 2>&1 | grep '"groups": 1, "max_chain": 1' | sed "s/^/$lib r$r /"
  done
done
