"""SIFT-style descriptor kernel alone: 4K checkerboard, Harris corners, size-10 keypoints; prints ms per call.
Run under `rocprofv3 --pmc ...` (tools/sift_pmc.sh) for the kernel's counters."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from introtocomputervision_amd import harris, lk, synth

rows, cols = (2160, 3840) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
reps = int(os.environ.get("REPS", "30"))
ctx = lk.default_context()
img = torch.from_numpy(synth.checkerboard(rows, cols, 40, seed=0x5EED0004)).cuda()
gx, gy = harris.getGradients(img, 3, ctx=ctx)
R = harris.getCornerResponse(gx, gy, 5, 1.5, 0.04, ctx=ctx)
kp = harris.getKeypoints(gx, gy, harris.refineCorners(R, 5e8, 5, capacity=1 << 16, ctx=ctx)[1], 10, ctx=ctx)
kpd = torch.from_numpy(kp).cuda() if not torch.is_tensor(kp) else kp
for _ in range(3):
    harris.computeDescriptors(gx, gy, kpd, ctx=ctx)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    harris.computeDescriptors(gx, gy, kpd, ctx=ctx)
torch.cuda.synchronize()
print({"keypoints": len(kp), "ms": round((time.perf_counter() - t0) * 1e3 / reps, 4)})
