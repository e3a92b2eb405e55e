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
    assert d["value"] > 1000 and d["roofline"]["frac"] > 0.01 and d["cpu_baseline"]["kind"] == "port"


def test_bench_rowshard_mode_over_rccl_world_1():
    d = _bench("--cpu-pairs", "0", "--mode", "rowshard", "--no-profile-pass")
    assert d["config"]["rccl_ranks"] == 1 and d["config"]["mode"] == "rowshard" and d["scaling"] == "strong"
    assert d["config"]["flow_check"]["ok"] and d["value"] > 1000


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
