#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table of one csrc/*.hip file (no GPU needed):
compiles it with -Rpass-analysis=kernel-resource-usage and prints one line per kernel.
  python tools/kernel_resources.py lk_fused [-DMICV_DIAG ...] [--filter lk_level_kernel]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "introtocomputervision_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fno-gpu-flush-denormals-to-zero", "-Rpass-analysis=kernel-resource-usage"]


def main():
    args = sys.argv[1:]
    flt = None
    if "--filter" in args:
        i = args.index("--filter")
        flt = args[i + 1]
        del args[i:i + 2]
    name = args[0]
    extra = args[1:]
    src = os.path.join(CSRC, name if name.endswith(".hip") else name + ".hip")
    out = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-c", src, "-o", "/dev/null"],
                         capture_output=True, text=True)
    if out.returncode:
        sys.stderr.write(out.stderr)
        sys.exit(out.returncode)
    cur = None
    rows = []
    for line in out.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs Spill|SGPRs Spill|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|TotalSGPRs|"
                      r"Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    for r in rows:
        n = re.sub(r"\(.*", "", r["name"]).replace("void micv::", "")
        if flt and flt not in n:
            continue
        print(f"{n:58s} vgpr {r.get('VGPRs','?'):>4} spill {r.get('VGPRs Spill','?'):>3} sgpr {r.get('TotalSGPRs','?'):>4} "
              f"sspill {r.get('SGPRs Spill','?'):>3} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} "
              f"occ {r.get('Occupancy [waves/SIMD]','?')} lds {r.get('LDS Size [bytes/block]','?')}")


if __name__ == "__main__":
    main()
