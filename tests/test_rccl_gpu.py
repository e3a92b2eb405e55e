"""RCCL under test on the one-GPU box: every path that touches torch.distributed's
nccl backend (= RCCL on ROCm) -- bench.py's barrier / MAX / all-gather, the row-shard batch's
batch_isend_irecv halo exchange, the Hough int32 all-reduce, the Harris corner-list all-gather -- runs
with init_process_group("nccl", world_size=1) in a FRESH child process (a process group cannot be
re-initialised inside the pytest process, and the child never execs after touching the GPU).
With one rank the p2p transfer list is empty and the collectives are identities: what this proves is
that RCCL initialises on this image, the code path is the distributed one, and results equal the
non-distributed ones.  The 2-rank logic is covered by the gloo tests (tests/test_shard*.py)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                "MICV_BENCH_FORCE_DIST": "1"})
    return env


def _bench(*extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                        "--preroll-s", "0", "--sustained-s", "0", *extra],
                       capture_output=True, text=True, env=_env(), timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_pairs_mode_over_rccl_world_1():
    d = _bench("--cpu-pairs", "1")
    assert d["config"]["rccl_ranks"] == 1 and d["n_gpus"] == 1 and d["config"]["mode"] == "pairs"
    assert d["config"]["flow_check"]["ok"]
    assert d["config"]["parity_1080p"] is True  # same bits as the oracle, with the process group up
    # the N > 1 hardening runs here too: micv_comm_selftest over RCCL, per-rank checksum gather (one rank: nothing to compare)
    assert d["config"]["comm_selftest"] == "ok" and d["config"]["per_rank_parity"]["ok"]
    assert d["value"] > 1000 and d["roofline"]["frac"] > 0.01 and d["cpu_baseline"]["kind"] == "port"


def test_bench_rowshard_mode_over_rccl_world_1():
    d = _bench("--cpu-pairs", "0", "--mode", "rowshard", "--no-profile-pass")
    assert d["config"]["rccl_ranks"] == 1 and d["config"]["mode"] == "rowshard" and d["scaling"] == "strong"
    assert d["config"]["flow_check"]["ok"] and d["value"] > 1000 and d["config"]["comm_selftest"] == "ok"


_CHILD = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
out = sys.argv[2]
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
from introtocomputervision_amd import shard, shard_ops as so, synth, lk
from introtocomputervision_amd._capi import Context
ctx = Context(0)
fns = so.gpu_fns(ctx)
comm = so.TorchDist(0, 1)
# (c) Hough votes summed by the int32 all-reduce, Harris corner list by the all-gather
m = synth.hough_mask(270, 480)[0]
dm = torch.from_numpy(m).cuda()
lines = so.hough_lines_sharded(dm, (0, 270), 270, 1, 1, fns.hough_lines_band, comm)
circ = so.hough_circles_sharded(dm, (0, 270), 270, 20, fns.hough_circles_band, comm)
img = synth.checkerboard(240, 320, 40, seed=1)
di = torch.from_numpy(img).cuda()
r, c, locs = so.harris_sharded(di, (0, 240), (0, 240), 3, 5, 1.5, 0.04, 5e8, 5, fns.grad, fns.response, fns.refine, comm)
# (b) the row-shard batch with its batched exchange, kernels on a NON-current stream
B, rows, cols, levels, win = 2, 270, 480, 4, 15
pn = [synth.lk_pair(77 + i, rows, cols, 2, -1) for i in range(B)]
prev = torch.from_numpy(np.stack([p for p, _ in pn])).cuda()
nxt = torch.from_numpy(np.stack([n for _, n in pn])).cuda()
u = torch.zeros_like(prev); v = torch.zeros_like(prev)
runner = shard.RowShardBatch(ctx, rows, cols, levels, win, B, 0, 1, comm=shard.DistComm(0, 1))
side = torch.cuda.Stream()
torch.cuda.synchronize()
runner.run(prev, nxt, u, v, side.cuda_stream)
side.synchronize()
ref_u, ref_v = lk.calcOpticalFlowPyrBatch(prev, nxt, win, levels, ctx=Context(0))
torch.cuda.synchronize()
# a plain p2p round trip with ourselves is not defined for nccl; a barrier and an all_gather are
dist.barrier()
t = torch.arange(4, device="cuda", dtype=torch.float64)
g = [torch.zeros_like(t)]
dist.all_gather(g, t)
np.savez(out, lines=lines.cpu().numpy(), circ=circ.cpu().numpy(), locs=locs.cpu().numpy(),
         u=u.cpu().numpy(), v=v.cpu().numpy(), ref_u=ref_u.cpu().numpy(), ref_v=ref_v.cpu().numpy(),
         gathered=g[0].cpu().numpy())
dist.destroy_process_group()
'''


def test_sharded_ops_over_rccl_world_1(tmp_path):
    import _oracle as orc
    from introtocomputervision_amd import synth
    out = str(tmp_path / "out.npz")
    env = _env()
    p = subprocess.run([sys.executable, "-c", _CHILD, ROOT, out], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = np.load(out)
    m = synth.hough_mask(270, 480)[0]
    assert np.array_equal(d["lines"], orc.hough_lines(m, 1, 1))
    assert np.array_equal(d["circ"], orc.hough_circles(m, 20))
    img = synth.checkerboard(240, 320, 40, seed=1)
    gx, gy = orc.sobel(img, 3, 1.0)
    _, locs = orc.harris_refine(orc.harris_response(gx, gy, 5, 1.5, 0.04), 5e8, 5)
    assert np.array_equal(d["locs"], locs) and len(locs) > 10
    assert np.array_equal(d["u"], d["ref_u"]) and np.array_equal(d["v"], d["ref_v"])
    for i in range(2):
        pn = synth.lk_pair(77 + i, 270, 480, 2, -1)
        eu, ev = orc.lk_flow_pyr(pn[0], pn[1], 15, 4)
        assert np.array_equal(d["u"][i], eu) and np.array_equal(d["v"][i], ev)
    assert np.array_equal(d["gathered"], np.arange(4.0))


# ---- the C ABI's own communicator (csrc/comm.hip) ---------------------------------------------

_CHILD_NATIVE = r'''
import os, sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
out = sys.argv[2]
torch.cuda.set_device(0)
from introtocomputervision_amd import shard, synth, lk, hough
from introtocomputervision_amd._capi import Context, check, lib
ctx = Context(0)
# a communicator of one rank from an RCCL unique id: ncclGetUniqueId + ncclCommInitRank inside libmicv (dlopen'd RCCL)
comm = shard.MicvComm(ctx, 0, 1)
comm.selftest()   # micv_comm_selftest: ring slab (a copy at world 1) + int32 all-reduce, verified on the device
B, rows, cols, levels, win = 2, 270, 480, 4, 15
pn = [synth.lk_pair(77 + i, rows, cols, 2, -1) for i in range(B)]
prev = torch.from_numpy(np.stack([p for p, _ in pn])).cuda()
nxt = torch.from_numpy(np.stack([n for _, n in pn])).cuda()
u = torch.full_like(prev, float("nan")); v = torch.full_like(prev, float("nan"))
runner = shard.RowShardNative(ctx, rows, cols, levels, win, B, comm)
side = torch.cuda.Stream()
torch.cuda.synchronize()
for _ in range(2):
    runner.run(prev, nxt, u, v, side.cuda_stream)   # a non-current stream: launches and RCCL calls all follow it
side.synchronize()
ref_u, ref_v = lk.calcOpticalFlowPyrBatch(prev, nxt, win, levels, ctx=Context(0))
# the cv::Mat caller's form
hu = np.zeros((rows, cols), np.float32); hv = np.zeros_like(hu)
for _ in range(3):   # repeated calls take the context's cached device blocks (no hipMalloc per call)
    check(lib.micv_lk_flow_pyr_rowshard_host(ctx.handle, comm.handle, pn[0][0].ctypes.data, pn[0][1].ctypes.data, rows, cols, cols * 4,
                                             win, levels, hu.ctypes.data, hv.ctypes.data, cols * 4))
# Hough: band launch + the int32 all-reduce through the library's communicator
m = synth.hough_mask(270, 480)[0]
dm = torch.from_numpy(m).cuda()
ref_acc = hough.houghLinesAccumulate(dm, 1, 1, ctx=ctx)
acc = torch.empty_like(ref_acc)
s = torch.cuda.current_stream().cuda_stream
check(lib.micv_hough_lines_rowshard_dev(ctx.handle, comm.handle, dm.data_ptr(), 270, 480, 480, 0, 270, 1, 1, acc.data_ptr(), s))
t = torch.arange(1000, device="cuda", dtype=torch.int32)
comm.allreduce_sum_i32(t)
torch.cuda.synchronize()
np.savez(out, u=u.cpu().numpy(), v=v.cpu().numpy(), ref_u=ref_u.cpu().numpy(), ref_v=ref_v.cpu().numpy(), hu=hu, hv=hv,
         acc=acc.cpu().numpy(), ref_acc=ref_acc.cpu().numpy(), t=t.cpu().numpy(), band=np.array(runner.band0))
comm.close()
'''


def test_native_communicator_world_1(tmp_path):
    """micv_comm_unique_id / micv_comm_create (RCCL dlopen'd by libmicv.so), micv_lk_flow_pyr_rowshard_dev on a side
    stream, the host form, the Hough all-reduce: one rank, fresh process -- results equal the unsharded calls and the
    oracle.  (World sizes > 1 of the same driver: test_native_virtual_ranks below.)"""
    import _oracle as orc
    from introtocomputervision_amd import synth
    out = str(tmp_path / "out.npz")
    p = subprocess.run([sys.executable, "-c", _CHILD_NATIVE, ROOT, out], capture_output=True, text=True, env=_env(), timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = np.load(out)
    assert tuple(d["band"]) == (0, 270)
    assert np.array_equal(d["u"], d["ref_u"]) and np.array_equal(d["v"], d["ref_v"])
    pn = synth.lk_pair(77, 270, 480, 2, -1)
    eu, ev = orc.lk_flow_pyr(pn[0], pn[1], 15, 4)
    assert np.array_equal(d["u"][0], eu) and np.array_equal(d["hu"], eu) and np.array_equal(d["hv"], ev)
    assert np.array_equal(d["acc"], d["ref_acc"]) and np.array_equal(d["t"], np.arange(1000))


@pytest.mark.parametrize("rows,cols,levels,world,win,batch", [(270, 480, 4, 2, 15, 2), (540, 960, 5, 8, 15, 1), (333, 517, 3, 3, 15, 2),
                                                              (270, 480, 3, 4, 21, 1), (200, 320, 3, 5, 7, 1), (1080, 1920, 5, 8, 15, 1)])
def test_native_virtual_ranks(rows, cols, levels, world, win, batch):
    """The C ABI's row-shard driver at world sizes > 1 on one device (micv_lk_flow_pyr_rowshard_virtual_dev): the plan,
    the slab packing and the band launches of micv_lk_flow_pyr_rowshard_dev with copies in place of ncclSend / ncclRecv,
    every rank's private memory NaN-poisoned first.  Same bits as the unsharded call."""
    torch = pytest.importorskip("torch")
    from introtocomputervision_amd import lk, shard, synth
    from introtocomputervision_amd._capi import Context
    ctx = Context(0)
    pn = [synth.lk_pair(501 + i + rows, rows, cols, 3, -2) for i in range(batch)]
    prev = torch.from_numpy(np.stack([p for p, _ in pn])).cuda()
    nxt = torch.from_numpy(np.stack([n for _, n in pn])).cuda()
    ref_u, ref_v = lk.calcOpticalFlowPyrBatch(prev, nxt, win, levels, ctx=ctx)
    u, v = shard.run_virtual_native(ctx, world, prev, nxt, win, levels, poison=True)
    assert torch.equal(u, ref_u) and torch.equal(v, ref_v)


def test_native_halo_is_needed_and_world_too_large_is_refused():
    torch = pytest.importorskip("torch")
    from introtocomputervision_amd import lk, shard, synth
    from introtocomputervision_amd._capi import Context, MicvError
    ctx = Context(0)
    p, n = synth.lk_pair(9, 270, 480, 3, -2)
    prev, nxt = torch.from_numpy(p[None]).cuda(), torch.from_numpy(n[None]).cuda()
    with pytest.raises(MicvError):  # 16-row coarsest level (270 >> 4) cannot be split 17 ways
        shard.run_virtual_native(ctx, 17, prev, nxt, 15, 5)


_CPP_CALLER = r'''
// A C++ process sharding lk::calcOpticalFlowPyr without Python: the shim with a communicator (world 1 here).
#include <cstdio>
#include <vector>
#include "introtocomputervision_amd/shim/micv_shim.hpp"
int main(int argc, char **argv) {
    const int rows = 135, cols = 240;
    micv_shim::Mat prev(rows, cols, micv_shim::F32), next(rows, cols, micv_shim::F32), u, v, u1, v1;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f || std::fread(prev.data, 4, (size_t)rows * cols, f) != (size_t)rows * cols || std::fread(next.data, 4, (size_t)rows * cols, f) != (size_t)rows * cols) return 3;
    std::fclose(f);
    lk::calcOpticalFlowPyr(prev, next, u1, v1, 15);             // unsharded
    unsigned char id[MICV_COMM_ID_BYTES];
    micv_shim::check(micv_comm_unique_id(id));                 // rank 0 makes it; MPI / a file would carry it to the others
    micv_shim::init_comm(id, 0, 1);                            // every rank: its rank and the world size
    lk::calcOpticalFlowPyr(prev, next, u, v, 15);              // row-sharded over the communicator, whole fields back
    micv_shim::close_comm();
    f = std::fopen(argv[2], "wb");
    std::fwrite(u.data, 4, (size_t)rows * cols, f); std::fwrite(v.data, 4, (size_t)rows * cols, f);
    std::fwrite(u1.data, 4, (size_t)rows * cols, f); std::fwrite(v1.data, 4, (size_t)rows * cols, f);
    std::fclose(f);
    return 0;
}
'''


def test_cpp_caller_shards_through_the_shim(tmp_path):
    """The reference's caller path (ps5_cpp/src/Solution.cpp:60-64 -> lk::calcOpticalFlowPyr) with a communicator set
    on the shim: C++ -> micv_lk_flow_pyr_rowshard_host -> RCCL, no Python in the process."""
    import _oracle as orc
    from introtocomputervision_amd import synth
    src = tmp_path / "caller.cpp"
    src.write_text(_CPP_CALLER)
    exe = str(tmp_path / "caller")
    lib = os.path.join(ROOT, "introtocomputervision_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + ROOT, "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe,
                    "-L" + lib, "-lmicv", "-Wl,-rpath," + lib], check=True)
    p, n = synth.lk_pair(31, 135, 240, 3, -2)
    with open(tmp_path / "in.f32", "wb") as f:
        f.write(p.tobytes()); f.write(n.tobytes())
    env = _env()
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe, str(tmp_path / "in.f32"), str(tmp_path / "out.f32")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    o = np.fromfile(tmp_path / "out.f32", np.float32).reshape(4, 135, 240)
    eu, ev = orc.lk_flow_pyr(p, n, 15, 4)  # the shim keeps the reference's depth of 4 (OpticalFlow.cpp:127)
    assert np.array_equal(o[0], eu) and np.array_equal(o[1], ev) and np.array_equal(o[2], eu) and np.array_equal(o[3], ev)


# ---- the real transport at world > 1: needs as many GPUs as ranks, so it is skipped on the one-GPU box ----

_CHILD_MULTI = r"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
rank, world, work = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
torch.cuda.set_device(rank)
from introtocomputervision_amd import shard, synth, lk
from introtocomputervision_amd._capi import Context, check, lib, MICV_COMM_ID_BYTES
import ctypes as C
ctx = Context(rank)
idf = os.path.join(work, "id.bin")
if rank == 0:
    buf = (C.c_char * MICV_COMM_ID_BYTES)()
    check(lib.micv_comm_unique_id(buf))
    open(idf + ".tmp", "wb").write(bytes(buf.raw)); os.rename(idf + ".tmp", idf)
t0 = time.time()
while not os.path.exists(idf):
    assert time.time() - t0 < 120, "no unique id from rank 0"
    time.sleep(0.05)
comm = shard.MicvComm(ctx, rank, world, unique_id=open(idf, "rb").read())
comm.selftest()
B, rows, cols, levels, win = 2, 540, 960, 5, 15
pn = [synth.lk_pair(77 + i, rows, cols, 2, -1) for i in range(B)]
prev = torch.from_numpy(np.stack([p for p, _ in pn])).cuda(); nxt = torch.from_numpy(np.stack([n for _, n in pn])).cuda()
u = torch.full_like(prev, float("nan")); v = torch.full_like(prev, float("nan"))
runner = shard.RowShardNative(ctx, rows, cols, levels, win, B, comm)
side = torch.cuda.Stream()
torch.cuda.synchronize()
for _ in range(3):
    runner.run(prev, nxt, u, v, side.cuda_stream)
side.synchronize()
ref_u, ref_v = lk.calcOpticalFlowPyrBatch(prev, nxt, win, levels, ctx=Context(rank))
a, b = runner.band0
band_ok = bool(torch.equal(u[:, a:b], ref_u[:, a:b]) and torch.equal(v[:, a:b], ref_v[:, a:b]))
hu = np.zeros((rows, cols), np.float32); hv = np.zeros_like(hu)
for _ in range(2):
    check(lib.micv_lk_flow_pyr_rowshard_host(ctx.handle, comm.handle, pn[0][0].ctypes.data, pn[0][1].ctypes.data, rows, cols, cols * 4,
                                             win, levels, hu.ctypes.data, hv.ctypes.data, cols * 4))
host_ok = bool(np.array_equal(hu, ref_u[0].cpu().numpy()) and np.array_equal(hv, ref_v[0].cpu().numpy()))
t = torch.full((1000,), rank + 1, device="cuda", dtype=torch.int32)
comm.allreduce_sum_i32(t)
torch.cuda.synchronize()
red_ok = bool((t == world * (world + 1) // 2).all())
json.dump({"rank": rank, "band": [a, b], "band_ok": band_ok, "host_ok": host_ok, "red_ok": red_ok}, open(os.path.join(work, f"r{rank}.json"), "w"))
comm.close()
"""


@pytest.mark.parametrize("world", [2, 4, 8])
def test_native_communicator_real_ranks(tmp_path, world):
    """The grouped ncclSend / ncclRecv halo exchange, the multi-root broadcast gather of the host form and the all-reduce
    over REAL ranks: one fresh process per GPU, a micv_comm from a shared unique id, micv_comm_selftest first.  Every
    rank's band and the host form's whole fields equal the unsharded bits.  Skipped without `world` GPUs (the pool's
    boxes have one: until this has run on a multi-GPU node the transport is unverified at world > 1)."""
    torch = pytest.importorskip("torch")
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {torch.cuda.device_count()}")
    procs = [subprocess.Popen([sys.executable, "-c", _CHILD_MULTI, ROOT, str(r), str(world), str(tmp_path)], env=_env(),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    for r in range(world):
        d = json.load(open(tmp_path / f"r{r}.json"))
        assert d["band_ok"] and d["host_ok"] and d["red_ok"], d
