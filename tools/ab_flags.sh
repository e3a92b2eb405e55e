#!/usr/bin/env bash
# A/B of bench.py flag sets on ONE box.  usage: bash tools/ab_flags.sh "<flags A>" "<flags B>" ...
repo="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
for r in 1 2 3; do
  for flags in "$@"; do
    python "$repo/bench.py" --cpu-pairs 0 --no-profile-pass --sustained-s 1.0 $flags 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$flags] r$r', 'value', round(d['value']), 'ms', round(d['ms_per_step'], 4), 'sustained', round(d['sustained']['ms_per_step'], 4), 'serial', round(d['config']['one_pass_at_a_time_ms_per_step'] or 0, 4))"
  done
done
