import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from introtocomputervision_amd import harris, _capi
rows, cols, density = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
rng = np.random.default_rng(rows + cols)
R = rng.random((rows, cols)).astype(np.float32)
thr = 1.0 - density
ys, xs = np.nonzero(R.astype(np.float64) >= thr)
exp = np.stack([ys, xs], axis=1).astype(np.int32)
dR = torch.from_numpy(R).cuda()
c1 = _capi.Context(0)
c3 = _capi.Context(0)
c3.set_option(_capi.OPT_COMPACT_3PASS, 1)
def step(name, fn):
    print(name, "...", end="", flush=True)
    t = time.time(); r = fn(); torch.cuda.synchronize(); print(" ok %.3f s" % (time.time() - t), flush=True); return r
for i, ctx in enumerate((c1, c3, c1, c1)):
    _, locs = step("call %d" % i, lambda: harris.refineCorners(dR, thr, 0, ctx=ctx))
    print("  equal:", np.array_equal(locs.cpu().numpy(), exp), flush=True)
_, locs = step("cap 7", lambda: harris.refineCorners(dR, thr, 0, capacity=7, ctx=c1))
print("  equal:", np.array_equal(locs.cpu().numpy(), exp[:7]), flush=True)
if len(sys.argv) > 4 and sys.argv[4] == "streams":
    s2 = torch.cuda.Stream()
    def two():
        with torch.cuda.stream(s2):
            dR2 = dR.clone()
            _, l2 = harris.refineCorners(dR2, thr, 0, ctx=c1)
        _, l1 = harris.refineCorners(dR, thr, 0, ctx=c1)
        return l1, l2
    l1, l2 = step("two streams", two)
    print("  equal:", np.array_equal(l1.cpu().numpy(), exp), np.array_equal(l2.cpu().numpy(), exp), flush=True)
if len(sys.argv) > 5:
    s2 = torch.cuda.Stream()
    mode = sys.argv[5]
    ca, cb = (c3, c3) if mode == "3pass" else (c1, _capi.Context(0))
    def two2():
        with torch.cuda.stream(s2):
            dR2 = dR.clone()
            _, l2 = harris.refineCorners(dR2, thr, 0, ctx=cb)
        _, l1 = harris.refineCorners(dR, thr, 0, ctx=ca)
        return l1, l2
    l1, l2 = step("two streams " + mode, two2)
    print("  equal:", np.array_equal(l1.cpu().numpy(), exp), np.array_equal(l2.cpu().numpy(), exp), flush=True)
if len(sys.argv) > 6:
    s2 = torch.cuda.Stream()
    def seq():
        with torch.cuda.stream(s2):
            dR2 = dR.clone()
            _, l2 = harris.refineCorners(dR2, thr, 0, ctx=c1)
        torch.cuda.synchronize()
        print(" [s2 done]", end="", flush=True)
        _, l1 = harris.refineCorners(dR, thr, 0, ctx=c1)
        torch.cuda.synchronize()
        print(" [default done]", end="", flush=True)
        with torch.cuda.stream(s2):
            _, l2 = harris.refineCorners(dR2, thr, 0, ctx=c1)
        return l1, l2
    l1, l2 = step("same ctx, two streams, one after the other", seq)
    print("  equal:", np.array_equal(l1.cpu().numpy(), exp), np.array_equal(l2.cpu().numpy(), exp), flush=True)
