#!/usr/bin/env python3
"""A/B of context OPTIONS on one box, interleaved rounds (the pool's boxes differ by several per cent, and a
box drifts within a run): `python tools/opt_ab.py <rounds> <pairs> "OPT_A=1 OPT_B=2" "OPT_A=0" ...` runs
tools/level_bench.py once per option set and round and prints every line plus the per-set medians.
An option set of "-" is the library default."""
import json, os, statistics, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
rounds, pairs, sets = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
res = {s: [] for s in sets}
for r in range(rounds):
    for s in sets:
        args = [] if s == "-" else s.split()
        out = subprocess.run([sys.executable, os.path.join(here, "level_bench.py"), "15", pairs] + args, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(json.dumps({"set": s, "error": out.stderr[-300:]})); continue
        d = json.loads(line[-1]); d["set"] = s; d["round"] = r
        res[s].append(d); print(json.dumps(d), flush=True)
for s, v in res.items():
    if v:
        print(json.dumps({"set": s, "n": len(v), "median_ms_per_step": round(statistics.median(x["ms_per_step"] for x in v), 4),
                          "median_level_ms": [round(statistics.median(x["level_ms"][l] for x in v), 4) for l in range(5)]}))
