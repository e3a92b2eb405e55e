#!/usr/bin/env bash
# Builds introtocomputervision_amd/libmicv.so (gfx950 only) from csrc/*.hip.
#   -ffp-contract=off   fused multiply-adds only where the source says fmaf() -- the
#                       arithmetic contract that makes HIP == CPU oracle bit for bit.
#   objects are compiled in parallel, then linked; only amdhip64 is linked (no torch).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
# MICV_OUT=libmicv_diag.so EXTRA_HIPCC_FLAGS=-DMICV_DIAG builds the diagnostic flavour (in-kernel
# phase stamps, MICV_LK_STOP) beside the product library, with its own object directory.
name="${MICV_OUT:-libmicv.so}"
out="$here/../$name"
obj="$here/.obj"
[[ "$name" != "libmicv.so" ]] && obj="$here/.obj_${name%.so}"
# objects are keyed on the extra flags as well (the audit's -DMICV_LK_COL_FULLWAIT rebuild used to leave its
# objects where a later plain build would find them fresh and link them)
[[ -n "${EXTRA_HIPCC_FLAGS:-}" ]] && obj="${obj}_$(printf '%s' "${EXTRA_HIPCC_FLAGS}" | cksum | cut -d' ' -f1)"
mkdir -p "$obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
#   -fno-gpu-flush-denormals-to-zero  keep f32 subnormals, as the host oracle does.
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math
       -fno-gpu-flush-denormals-to-zero -Wall -Wno-unused-function)
pids=()
objs=()
for src in "$here"/*.hip; do
  o="$obj/$(basename "${src%.hip}").o"
  objs+=("$o")
  if [[ ! -f "$o" || "$src" -nt "$o" || -n "$(find "$here" -maxdepth 1 \( -name '*.hpp' -o -name 'build.sh' \) -newer "$o" -print -quit)" || "$here/../../include/mi_cv.h" -nt "$o" ]]; then
    "$HIPCC" "${FLAGS[@]}" ${EXTRA_HIPCC_FLAGS:-} -c "$src" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}"
echo "built $out"
# MICV_AUDIT=1: after building, run the ISA audit of the hand-placed LDS loads (tools/audit_asm_loads.py: every load has
# its wait before its first use; no compiler-placed LDS / scalar-memory operation inside a COUNTED wait's window).  When
# it fails -- a compiler upgrade that reloads taps inside a counted window would give wrong window sums, not a build
# error -- the two files that use counted waits are rebuilt with the single-wait column pass (-DMICV_LK_COL_FULLWAIT,
# ~1 % slower) and audited again.  (The CPU test suite runs the same audit; this is for builds that ship without it.)
if [[ -n "${MICV_AUDIT:-}" && -z "${MICV_AUDIT_DONE:-}" ]]; then
  audit="$here/../../tools/audit_asm_loads.py"
  if ! python3 "$audit" ${EXTRA_HIPCC_FLAGS:-}; then
    echo "audit failed: rebuilding with -DMICV_LK_COL_FULLWAIT (its own object directory)"
    MICV_AUDIT_DONE=1 EXTRA_HIPCC_FLAGS="${EXTRA_HIPCC_FLAGS:-} -DMICV_LK_COL_FULLWAIT" bash "$here/build.sh"
    python3 "$audit" ${EXTRA_HIPCC_FLAGS:-} -DMICV_LK_COL_FULLWAIT
  fi
fi
