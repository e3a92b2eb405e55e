import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import _oracle as orc
from introtocomputervision_amd import hough
rows, cols, seed, gs, sigma, lo, hi = 70, 107, 147, 31, 2.125, 0, 102
rng = np.random.default_rng(seed)
img = rng.integers(0, 256, (rows, cols)).astype(np.uint8)
fn = orc._sig("orc_generate_edge", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_size_t])
exp = np.empty((rows, cols), np.uint8)
assert fn(img.ctypes.data, rows, cols, cols, gs, float(sigma), float(lo), float(hi), exp.ctypes.data, cols) == 0
bad = 0
for rep in range(200):
    got = hough.generateEdge(torch.from_numpy(img).cuda(), gs, float(sigma), lo, hi).cpu().numpy()
    d = got != exp
    if d.any():
        bad += 1
        if bad <= 3: print("rep", rep, int(d.sum()), np.argwhere(d)[:6].tolist())
print("bad runs", bad, "of 200; edges in oracle", int((exp > 0).sum()))
