#!/usr/bin/env bash
# A/B of the headline bench on ONE box: bench.py against two builds of the library.
repo="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
for r in 1 2; do
  for lib in "$@"; do
    MICV_LIB="$repo/introtocomputervision_amd/$lib" python "$repo/bench.py" --cpu-pairs 0 --no-profile-pass --sustained-s 1.0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib r$r', 'value', round(d['value']), 'ms', round(d['ms_per_step'], 4), 'sustained', round(d['sustained']['ms_per_step'], 4), 'serial', round(d['config']['one_pass_at_a_time_ms_per_step'], 4), 'single', round(d['config']['single_pair_ms'], 4))"
  done
done
