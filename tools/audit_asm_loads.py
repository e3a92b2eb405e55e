#!/usr/bin/env python3
"""ISA audit of the hand-placed LDS loads of csrc/lk_fused.hip and csrc/lk_split.hip (the column pass's ds_read2st64_b32, which
hipcc does not count -- MI355X HIP guide, section 5.7): compiles the file with -save-temps and checks, in
every kernel, that between an asm-block load and the asm-block `s_waitcnt lgkmcnt(N)` that retires it (N = the
younger operations that may stay outstanding: the column pass waits in three steps, r04) NO instruction touches the
load's destination registers, and no LDS / scalar-memory instruction of the compiler's own sits inside the counted window (a register copy, a spill or a reuse placed there by
the compiler would read data that has not landed: wrong results that depend on timing).  Exit status 0 and
a one-line summary when clean; prints every offending instruction otherwise.  No GPU needed.
  python tools/audit_asm_loads.py [-DMICV_DIAG ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = [os.path.join(ROOT, "introtocomputervision_amd", "csrc", f) for f in ("lk_fused.hip", "lk_split.hip")]
# kernels whose column pass waits in COUNTED steps (lk_window.hpp, col_pass_partial): the audit fails when one of them
# is not found in the listing or holds no partial wait -- a renamed kernel must not skip the check
REQUIRED_PARTIAL = ("lk_level_kernelILi7ELi1ELi512ELi32ELb0ELi64E", "lk_sums_stream_kernelILi7E")
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
    """pending: the hand-placed LDS operations not yet retired, in issue order (LDS operations return in order, so
    `s_waitcnt lgkmcnt(N)` retires all but the N youngest).  While any is pending: nothing may touch its destination,
    no control flow, and -- because the waits count operations -- no LDS or scalar-memory instruction the compiler
    placed on its own (it would shift a partial count; a scalar load also returns out of order)."""
    problems, loads_seen, kernels = [], 0, 0
    kernel = None
    in_asm = False
    pending = []  # (line number, destination vgprs) in issue order
    foreign = []  # compiler-placed LDS / scalar-memory instructions seen while something is pending

    def pending_regs():
        out = {}
        for ln_, regs in pending:
            for r in regs:
                out[r] = ln_
        return out

    for ln, raw in enumerate(asm_text.split("\n"), 1):
        line = raw.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, pending, foreign = m.group(1), [], []
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
            clash = dst & set(pending_regs())
            if clash:
                problems.append((kernel, ln, line, f"overwrites v{sorted(clash)} of a load still in flight (line {pending_regs()[min(clash)]})"))
            pending.append((ln, dst))
            loads_seen += 1
            continue
        if in_asm and mnem.startswith("ds_write") and pending:
            pending.append((ln, set()))  # counts in lgkmcnt, writes no register
            continue
        if in_asm and mnem == "s_waitcnt":
            mm = re.search(r"lgkmcnt\((\d+)\)", ops)
            if mm:
                n = int(mm.group(1))
                if n > 0:  # a PARTIAL wait counts operations: none of the compiler's own may be among them
                    for fl, ftxt in foreign:
                        problems.append((kernel, fl, ftxt, f"a compiler-placed LDS / scalar-memory operation inside a window "
                                                           f"retired by a partial wait (lgkmcnt({n}) at line {ln})"))
                pending = pending[len(pending) - n:] if n and n < len(pending) else ([] if n == 0 else pending)
                if not pending:
                    foreign = []
            continue
        if pending:
            if mnem in ("s_endpgm", "s_branch", "s_cbranch_execz", "s_cbranch_execnz", "s_cbranch_vccz", "s_cbranch_vccnz",
                        "s_cbranch_scc0", "s_cbranch_scc1", "s_barrier"):
                problems.append((kernel, ln, line, "control flow or a barrier between a hand-placed load and its wait"))
                pending, foreign = [], []
                continue
            if not in_asm and (mnem.startswith(("ds_", "s_load", "s_buffer_load", "s_memtime", "s_memrealtime"))):
                foreign.append((ln, line))  # harmless before a full wait, a miscount before a partial one
            touched = regs_of(ops) & set(pending_regs())
            if touched:
                problems.append((kernel, ln, line, f"touches v{sorted(touched)} before the wait (loaded at line {pending_regs()[min(touched)]})"))
    return problems, loads_seen, kernels


def partial_waits_by_kernel(asm_text):
    """{function symbol: asm-block `s_waitcnt lgkmcnt(N > 0)` statements in it}"""
    out, kernel, in_asm = {}, None, False
    for raw in asm_text.split("\n"):
        line = raw.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel = m.group(1)
            out.setdefault(kernel, 0)
        elif line.startswith(";;#ASMSTART"):
            in_asm = True
        elif line.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm and kernel and line.startswith("s_waitcnt"):
            mm = re.search(r"lgkmcnt\((\d+)\)", line)
            if mm and int(mm.group(1)) > 0:
                out[kernel] += 1
    return out


def main():
    problems, loads, kernels, partial = [], 0, 0, {}
    for src in SRCS:
        with tempfile.TemporaryDirectory() as d:
            r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *sys.argv[1:], "-save-temps", "-c", src, "-o", os.path.join(d, "x.o")],
                               cwd=d, capture_output=True, text=True)
            if r.returncode:
                sys.stderr.write(r.stderr[-3000:])
                return 2
            s = [f for f in os.listdir(d) if f.endswith("gfx950.s")]
            text = open(os.path.join(d, s[0])).read()
        p, l, k = audit(text)
        problems += p
        loads += l
        kernels += k
        partial.update(partial_waits_by_kernel(text))
    for k, ln, line, why in problems:
        print(f"{k}: line {ln}: `{line}` {why}")
    fullwait = any("MICV_LK_COL_FULLWAIT" in a for a in sys.argv[1:])  # the single-wait build has no counted windows to find
    missing = [] if fullwait else [req for req in REQUIRED_PARTIAL if not any(req in k and n > 0 for k, n in partial.items())]
    for req in missing:
        print(f"audit: no function matching `{req}` with a counted (partial) wait was found -- renamed kernel or a "
              f"-DMICV_LK_COL_FULLWAIT build: the counted windows were NOT checked")
    print(f"audit: {loads} hand-placed LDS loads in {kernels} functions, {len(problems)} problems, "
          f"{sum(1 for n in partial.values() if n)} functions with counted waits")
    return 1 if problems or loads == 0 or missing else 0


if __name__ == "__main__":
    sys.exit(main())
