#!/usr/bin/env python3
"""A/B/C... on ONE box, interleaved rounds (the pool's boxes differ by up to 9 %, and clocks drift):
  python tools/ab.py <rounds> <lib.so> [<lib.so> ...] [-- level_bench args]
Each round runs tools/level_bench.py once per library (MICV_LIB selects the build; names are relative to
introtocomputervision_amd/).  Prints every run and, at the end, the median step / level-0 time per library."""
import json, os, statistics, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
rounds, libs = int(args[0]), args[1:]
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, MICV_LIB=os.path.join(root, "introtocomputervision_amd", l))
        p = subprocess.run([sys.executable, os.path.join(root, "tools", "level_bench.py"), *extra], env=env,
                           capture_output=True, text=True)
        line = [x for x in p.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(l, "FAILED", p.stderr[-500:])
            continue
        d = json.loads(line[-1])
        res[l].append(d)
        print(f"r{r} {l}: {json.dumps(d)}", flush=True)
for l in libs:
    if res[l]:
        print(json.dumps({"lib": l, "runs": len(res[l]),
                          "median_ms_per_step": round(statistics.median(d["ms_per_step"] for d in res[l]), 4),
                          "median_level0_ms": round(statistics.median(d["level_ms"][0] for d in res[l]), 4),
                          "median_level_ms": [round(statistics.median(d["level_ms"][k] for d in res[l]), 4) for k in range(5)]}))
