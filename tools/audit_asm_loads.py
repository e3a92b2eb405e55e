#!/usr/bin/env python3
"""ISA audit of the hand-placed LDS loads of csrc/lk_fused.hip (the column pass's ds_read2st64_b32, which
hipcc does not count -- MI355X HIP guide, section 5.7): compiles the file with -save-temps and checks, in
every kernel, that between an asm-block load and the asm-block `s_waitcnt lgkmcnt(0)` that retires it NO
instruction touches the load's destination registers (a register copy, a spill or a reuse placed there by
the compiler would read data that has not landed: wrong results that depend on timing).  Exit status 0 and
a one-line summary when clean; prints every offending instruction otherwise.  No GPU needed.
  python tools/audit_asm_loads.py [-DMICV_DIAG ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "introtocomputervision_amd", "csrc", "lk_fused.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fno-gpu-flush-denormals-to-zero"]


def regs_of(operand_text):
    """VGPR numbers named in an operand string: v12, v[12:13]."""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", operand_text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", operand_text):
        out.add(int(m.group(1)))
    return out


def audit(asm_text):
    problems, loads_seen, kernels = [], 0, 0
    kernel = None
    in_asm = False
    pending = {}  # vgpr -> line number of the load that writes it
    for ln, raw in enumerate(asm_text.split("\n"), 1):
        line = raw.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, pending = m.group(1), {}
            kernels += 1
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line.startswith((";", ".")) or line.endswith(":"):
            continue
        mnem, _, ops = line.partition(" ")
        if in_asm and mnem.startswith("ds_read"):
            dst = regs_of(ops.split(",")[0])
            clash = dst & set(pending)
            if clash:
                problems.append((kernel, ln, line, f"overwrites v{sorted(clash)} of a load still in flight (line {pending[min(clash)]})"))
            for r in dst:
                pending[r] = ln
            loads_seen += 1
            continue
        if in_asm and mnem == "s_waitcnt" and "lgkmcnt(0)" in ops:
            pending = {}
            continue
        if pending:
            if mnem in ("s_endpgm", "s_branch", "s_cbranch_execz", "s_cbranch_execnz", "s_cbranch_vccz", "s_cbranch_vccnz",
                        "s_cbranch_scc0", "s_cbranch_scc1", "s_barrier"):
                problems.append((kernel, ln, line, "control flow or a barrier between a hand-placed load and its wait"))
                pending = {}
                continue
            touched = regs_of(ops) & set(pending)
            if touched:
                problems.append((kernel, ln, line, f"touches v{sorted(touched)} before the wait (loaded at line {pending[min(touched)]})"))
    return problems, loads_seen, kernels


def main():
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *sys.argv[1:], "-save-temps", "-c", SRC, "-o", os.path.join(d, "x.o")],
                           cwd=d, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr[-3000:])
            return 2
        s = [f for f in os.listdir(d) if f.endswith("gfx950.s")]
        text = open(os.path.join(d, s[0])).read()
    problems, loads, kernels = audit(text)
    for k, ln, line, why in problems:
        print(f"{k}: line {ln}: `{line}` {why}")
    print(f"audit: {loads} hand-placed LDS loads in {kernels} functions, {len(problems)} problems")
    return 1 if problems or loads == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
