#!/usr/bin/env bash
# 64x64 / 1024-thread tiles (MICV_OPT_LK_TALL_TILES) against the 64x32 default, plain and streamed,
# interleaved on one box:  bash tools/tall_bench.sh [rounds]
repo="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
for r in $(seq 1 "${1:-3}"); do
  python "$repo/tools/level_bench.py" 15 8 | sed "s/^/r$r default      /"
  python "$repo/tools/level_bench.py" 15 8 OPT_LK_TALL_TILES=1 | sed "s/^/r$r tall         /"
  python "$repo/tools/level_bench.py" 15 8 OPT_LK_TALL_TILES=1 OPT_LK_STREAM=1 | sed "s/^/r$r tall+stream  /"
  python "$repo/tools/level_bench.py" 15 8 OPT_LK_STREAM=1 | sed "s/^/r$r stream       /"
done
