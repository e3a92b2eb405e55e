#!/usr/bin/env python3
"""Splits one kernel of a --save-temps .s file at its barriers and prints the instruction mix of every segment
(static counts), optionally dumping the kernel body with line numbers relative to its start.
  python tools/isa_segments.py file.s 'lk_level_kernelILi7ELi1ELi512ELi32E' [--dump out.s]"""
import re
import sys
from collections import Counter


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + pat + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    if "--dump" in sys.argv:
        open(sys.argv[sys.argv.index("--dump") + 1], "w").write("\n".join(body))
    bars = [i for i, l in enumerate(body) if "s_barrier" in l]
    print(len(body), "lines, barriers at", bars)
    for a, b in zip([0] + bars, bars + [len(body)]):
        seg = [l.split()[0] for l in body[a:b] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        valu = sum(1 for x in seg if x.startswith("v_"))
        print(f"{a:6d}..{b:6d} {len(seg):5d} instr, {valu:5d} v_*: {Counter(seg).most_common(12)}")


if __name__ == "__main__":
    main()
