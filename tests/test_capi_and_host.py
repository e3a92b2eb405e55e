"""CPU-side checks: libmicv.so loads and exports every symbol include/mi_cv.h declares (no
compute calls here), host-only entry points, the synthetic-input generator and the byte model."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "mi_cv.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(micv_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from introtocomputervision_amd import _capi
    names = header_functions()
    assert len(names) >= 50
    missing = [n for n in names if not hasattr(_capi.lib, n)]
    assert not missing, f"declared in include/mi_cv.h but not exported: {missing}"
    assert not _capi.MISSING
    undeclared = [n for n in names if n not in _capi.SIGNATURES]
    assert not undeclared, f"no ctypes signature for: {undeclared}"
    assert _capi.lib.micv_version().decode().startswith("micv")


def test_product_does_not_link_or_reference_the_oracle():
    import subprocess
    so = os.path.join(ROOT, "introtocomputervision_amd", "libmicv.so")
    deps = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
    assert "oracle" not in deps and "torch" not in deps
    for dirpath, _, files in os.walk(os.path.join(ROOT, "introtocomputervision_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".sh")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "_oracle" not in text and "orc_" not in text, f


def test_host_only_entry_points():
    from introtocomputervision_amd import _capi, hough
    # common::divRoundUp (Utils.h:12-15): max(1, ceil(float(n)/float(d)))
    assert _capi.lib.micv_div_round_up(10, 3) == 4
    assert _capi.lib.micv_div_round_up(0, 16) == 1
    assert _capi.lib.micv_div_round_up(1920, 64) == 30
    assert hough.linesAccumulatorShape(1080, 1920, 1, 1) == (4406, 180)
    assert hough.linesAccumulatorShape(1080, 1920, 2, 7) == (2203, 26)
    with pytest.raises(_capi.MicvError):
        hough.linesAccumulatorShape(0, 10)


def test_errors_do_not_need_a_gpu():
    from introtocomputervision_amd import _capi
    import ctypes as C
    h = _capi.vp()
    rc = _capi.lib.micv_ctx_create(0, C.byref(h))
    if rc == 0:  # a GPU is present: nothing to check here
        _capi.lib.micv_ctx_destroy(h)
        return
    assert rc in (_capi.EHIP, _capi.EINVAL) and _capi.last_error()


def test_splitmix64_reference_values():
    from introtocomputervision_amd import synth
    # splitmix64(seed=0) first outputs: e220a8397b1dcdaf 6e789e6aa1b965f4 06c45d188009454f
    assert synth.splitmix64_u8(0, 3).tolist() == [0xE2, 0x6E, 0x06]
    a = synth.smooth_noise(0x5EED0005, 32, 48)
    assert a.dtype == np.float32 and np.array_equal(a, np.round(a)) and 0 <= a.min() and a.max() <= 255
    p, n = synth.lk_pair(0x5EED0005, 32, 48, 3, -2)
    assert np.array_equal(n[0, 3:], p[2, :-3])  # next(y, x) = prev(y + 2, x - 3)
    l, r, nd = synth.stereo_pair(1, 20, 64)
    assert np.array_equal(r[5, :30], l[5, -nd[5]:-nd[5] + 30])


def test_byte_model_matches_baseline_md():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.algorithmic_bytes_pair(1080, 1920, 5) == 55207680
    assert b.algorithmic_bytes_pair(1080, 1920, 4) == 54950400
    assert b.algorithmic_bytes_pair(2160, 3840, 5) == 220838400
    assert b.level0_kernel_bytes_pair(1080, 1920, 5) == 16 * 2073600 + 8 * 518400


def test_bench_gpus_n_launches_n_ranks():
    """`python bench.py --gpus 2` run directly (no torchrun environment) must start 2 ranks itself:
    the launcher spawns torch.distributed.run as a child and relays rank 0's JSON line.  --dry-run
    over gloo: no kernels (there is no CPU fallback), only launcher / rendezvous / MAX / JSON."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--dry-run", "--steps", "3", "--warmup", "1"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2 and len(d["config"]["per_rank_s"]) == 2
    assert d["steps"] == 3 and d["value"] is None and "dry run" in d["data"]
    # a failing child is reported, not swallowed (gloo without --dry-run is refused by every rank)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--steps", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_no_environment_knobs_in_the_default_build():
    """The shipped library reads no environment variables (options live in the context); the only
    getenv left is behind -DMICV_DIAG."""
    import subprocess
    so = os.path.join(ROOT, "introtocomputervision_amd", "libmicv.so")
    syms = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True).stdout
    assert "getenv" not in syms
    blob = open(so, "rb").read()
    for knob in (b"MICV_LK_STOP", b"MICV_LK_GROUPS", b"MICV_FORCE_GENERIC", b"MICV_LK_NT"):
        assert knob not in blob, knob


def _schedule(rows, cols, batch, win, max_chain):
    import ctypes as C
    from introtocomputervision_amd import _capi
    n, tw, th = _capi.i64(), _capi.i32(), _capi.i32()
    _capi.check(_capi.lib.micv_lk_schedule_host(rows, cols, batch, win, max_chain, None, 0, C.byref(n),
                                                C.byref(tw), C.byref(th)))
    buf = np.zeros((n.value, 4), np.int32)
    _capi.check(_capi.lib.micv_lk_schedule_host(rows, cols, batch, win, max_chain, buf.ctypes.data, n.value,
                                                C.byref(n), C.byref(tw), C.byref(th)))
    return buf, tw.value, th.value


@pytest.mark.parametrize("win", [15, 11, 7])
def test_chain_schedule_covers_every_tile_exactly_once(win):
    """the schedule of the chain / streamed launches probed column tiles_x/2 for the
    interior rows; for cols 194..206 that column is not x-interior and the interior tiles of column 1
    were never scheduled.  Host-only check over the widths around every tile-count change."""
    shapes = [(200, 400), (400, 200), (200, 194), (200, 206), (270, 480), (135, 240), (540, 960), (1080, 1920),
              (33, 64), (64, 65), (300, 129), (97, 333)]
    shapes += [(200, c) for c in range(120, 330, 7)] + [(r, 200) for r in range(40, 300, 13)]
    for rows, cols in shapes:
        for batch in (1, 3):
            for max_chain in (1, 2, 4, 32):
                sched, tw, th = _schedule(rows, cols, batch, win, max_chain)
                assert len(sched) % 8 == 0
                tiles_x, tiles_y = -(-cols // tw), -(-rows // th)
                seen = np.zeros((batch, tiles_y, tiles_x), np.int32)
                for x, y, cnt, p in sched:
                    assert 0 <= cnt <= max_chain
                    if cnt == 0:
                        continue
                    assert 0 <= p < batch and 0 <= x < tiles_x and 0 <= y and y + cnt <= tiles_y, (rows, cols, x, y, cnt)
                    seen[p, y:y + cnt, x] += 1
                assert (seen == 1).all(), (win, rows, cols, batch, max_chain, np.argwhere(seen != 1)[:5])


def test_cmake_build_exports_the_same_abi(tmp_path):
    """The top-level CMakeLists.txt (HIP language, gfx950) is the integration entry point SURVEY.md section 7
    describes: configure + build it from scratch and check that its libmicv.so exports exactly the symbols
    include/mi_cv.h declares, and links neither torch nor the oracle."""
    import shutil
    import subprocess
    if not (shutil.which("cmake") and shutil.which("ninja") and os.path.exists("/opt/rocm/lib/llvm/bin/clang++")):
        pytest.skip("cmake / ninja / ROCm clang not available")
    b = str(tmp_path / "build")
    r = subprocess.run(["cmake", "-S", ROOT, "-B", b, "-G", "Ninja"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run(["cmake", "--build", b], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    so = os.path.join(b, "libmicv.so")
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (micv_[a-z0-9_]+)$", syms, flags=re.M)))
    assert exported == header_functions()
    deps = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
    assert "amdhip64" in deps and "torch" not in deps and "oracle" not in deps
    assert os.path.exists(os.path.join(b, "liboracle.so"))


def test_hand_placed_lds_loads_are_not_touched_before_their_wait():
    """The column pass of the LK level kernel reads LDS with inline-asm ds_read2st64_b32 (the compiler's own
    load merging costs a v_mov per value there).  hipcc does not count asm loads: a copy, spill or reuse of a
    destination register between the load and the asm `s_waitcnt lgkmcnt(0)` would read stale data, depending
    on timing.  tools/audit_asm_loads.py compiles lk_fused.hip and checks the ISA of every kernel; its
    detector is checked first on a doctored listing."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("audit", os.path.join(ROOT, "tools", "audit_asm_loads.py"))
    audit = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(audit)
    good = """_Zk:
\t;;#ASMSTART
\tds_read2st64_b32 v[4:5], v9 offset0:1 offset1:5
\t;;#ASMEND
\tv_add_u32_e32 v1, v2, v3
\t;;#ASMSTART
\ts_waitcnt lgkmcnt(0)
\t;;#ASMEND
\tv_pk_fma_f32 v[6:7], v[4:5], s[0:1], v[6:7]
"""
    assert audit.audit(good) == ([], 1, 1)
    for bad_line in ("\tv_mov_b32_e32 v8, v5", "\tv_pk_fma_f32 v[6:7], v[4:5], s[0:1], v[6:7]", "\ts_barrier"):
        bad = good.replace("\tv_add_u32_e32 v1, v2, v3", bad_line)
        problems, loads, kernels = audit.audit(bad)
        assert len(problems) == 1 and loads == 1, bad_line
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_asm_loads.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ", 0 problems" in r.stdout and " 0 hand-placed" not in r.stdout


def test_shim_functions_have_the_reference_headers_types():
    """tests/cpp/shim_signatures.cpp: one static_assert per shim function against the type the reference header declares
    (ps5 OpticalFlow.h / Pyramids.h, ps4 Harris.h / Descriptors.h, ps2 Disparity*.h, ps1 Hough.h, ps7 MotionHistory.h),
    plus calls that rely on the reference's default arguments.  Compile-only (VERDICT r4 Missing #3; the first run of it
    found lk::calcOpticalFlowPyr carrying a sixth parameter)."""
    import subprocess
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-Wno-unused-function", "-I" + ROOT,
                        "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "shim_signatures.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_comm_builds_without_the_rccl_header(tmp_path):
    """RCCL is dlopen'ed; comm.hip must not need <rccl/rccl.h> to BUILD: -DMICV_NO_RCCL_HEADER compiles the
    branch with the local declarations of the NCCL 2.x ABI subset (the header branch static_asserts that they agree)."""
    import subprocess
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "introtocomputervision_amd", "csrc", "comm.hip")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-DMICV_NO_RCCL_HEADER",
                        "-c", src, "-o", str(tmp_path / "comm.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_level_kernel_name_needs_no_gpu_and_follows_the_options():
    """micv_lk_level_kernel_name is answered by the launch dispatch itself (nothing is launched).  No GPU here, so the
    context cannot be created: only the symbol and its argument checks are exercised on the CPU; the GPU suite
    compares names with what rocprofv3 reports (tests/test_lk_gpu.py)."""
    from introtocomputervision_amd import _capi
    assert hasattr(_capi.lib, "micv_lk_level_kernel_name")
    import ctypes as C
    buf = C.create_string_buffer(160)
    assert _capi.lib.micv_lk_level_kernel_name(None, 15, 1080, 1920, 8, buf, 160) == _capi.EINVAL
