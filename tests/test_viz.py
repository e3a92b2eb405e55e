"""ps5's driver-level plumbing without OpenCV: shim/micv_viz.hpp (C++) against
its numpy mirror introtocomputervision_amd/viz.py, byte for byte, and examples/ps5_demo end to end."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "introtocomputervision_amd")

HARNESS = r'''
#include "../../introtocomputervision_amd/shim/micv_viz.hpp"
#include <cstdlib>
static micv_viz::Mat loadf(const std::string &p, int r, int c) {
    micv_viz::Mat m(r, c, micv_shim::F32);
    std::ifstream f(p, std::ios::binary);
    f.read(reinterpret_cast<char *>(m.data), (std::streamsize)r * c * 4);
    return m;
}
int main(int argc, char **argv) {   // dir rows cols : draw / colour maps / file round trips, no GPU call
    const std::string d = argv[1];
    const int rows = std::atoi(argv[2]), cols = std::atoi(argv[3]);
    micv_viz::Mat img = micv_viz::imread(d + "/in.ppm"), grey = micv_viz::imread(d + "/in.pgm");
    micv_viz::Mat bmp = micv_viz::imread(d + "/check.bmp");
    micv_viz::imwrite(d + "/bmp_as.pgm", bmp);
    micv_viz::imwrite(d + "/copy.bmp", img);
    micv_viz::imwrite(d + "/copy_of_copy.ppm", micv_viz::imread(d + "/copy.bmp"));
    micv_viz::Mat u = loadf(d + "/u.f32", rows, cols), v = loadf(d + "/v.f32", rows, cols);
    micv_viz::Mat a = img.clone(), b = grey.clone();
    micv_viz::drawVelocityVectors(a, u, v, micv_viz::Scalar(0, 255, 0, 255));
    micv_viz::drawVelocityVectors(b, u, v, micv_viz::Scalar(0, 255, 0, 255));
    micv_viz::imwrite(d + "/arrows_rgb.ppm", a);
    micv_viz::imwrite(d + "/arrows_grey.ppm", b);
    micv_viz::imwrite(d + "/u_norm.pgm", micv_viz::normalize_minmax_u8(u));
    micv_viz::imwrite(d + "/u_jet.ppm", micv_viz::apply_colormap_jet(micv_viz::normalize_minmax_u8(u)));
    micv_viz::Mat flat = micv_viz::Mat::zeros(rows, cols, micv_shim::F32);
    micv_viz::imwrite(d + "/flat_norm.pgm", micv_viz::normalize_minmax_u8(flat));
    return 0;
}
'''


def _frames(rows, cols, seed):
    from introtocomputervision_amd import synth
    prev, nxt = synth.lk_pair(seed, rows, cols, 3, -2)
    rng = np.random.default_rng(seed)

    def colour(g):  # B, G, R planes that really differ
        c = np.stack([g * 0.8 + 20, 255 - g * 0.7, g * 0.5 + 60], axis=-1)
        return np.clip(c + rng.integers(-2, 3, c.shape), 0, 255).astype(np.uint8)
    return prev, nxt, colour(prev), colour(nxt)


def test_viz_header_matches_numpy_mirror(tmp_path):
    from introtocomputervision_amd import viz
    d = str(tmp_path)
    rows, cols = 97, 150
    prev, nxt, pc, nc = _frames(rows, cols, 5)
    viz.imwrite(os.path.join(d, "in.ppm"), pc)
    viz.imwrite(os.path.join(d, "in.pgm"), prev.astype(np.uint8))
    import shutil
    shutil.copy(os.path.join(ROOT, "tests", "golden", "check.bmp"), os.path.join(d, "check.bmp"))
    rng = np.random.default_rng(1)
    u = (rng.standard_normal((rows, cols)) * 6).astype(np.float32)
    v = (rng.standard_normal((rows, cols)) * 6).astype(np.float32)
    u[0, 0], v[0, 5], u[3, 10], v[6, 0] = np.nan, np.inf, 400.0, -300.0  # skipped / far outside the image
    u.tofile(os.path.join(d, "u.f32")); v.tofile(os.path.join(d, "v.f32"))
    src = os.path.join(d, "harness.cpp")
    open(src, "w").write(HARNESS.replace("../../introtocomputervision_amd", LIB))
    exe = os.path.join(d, "harness")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", src, "-o", exe, "-L" + LIB, "-lmicv",
                    "-Wl,-rpath," + LIB], check=True)
    subprocess.run([exe, d, str(rows), str(cols)], check=True)
    rd = lambda n: viz.imread(os.path.join(d, n))  # noqa: E731
    assert np.array_equal(rd("in.ppm"), pc) and np.array_equal(rd("in.pgm"), prev.astype(np.uint8))
    from PIL import Image
    assert np.array_equal(rd("bmp_as.pgm"), np.asarray(Image.open(os.path.join(d, "check.bmp")).convert("L")))
    assert np.array_equal(rd("copy_of_copy.ppm"), pc) and np.array_equal(rd("copy.bmp"), pc)
    assert np.array_equal(rd("arrows_rgb.ppm"), viz.drawVelocityVectors(pc, u, v))
    assert np.array_equal(rd("arrows_grey.ppm"), viz.drawVelocityVectors(prev.astype(np.uint8), u, v))
    assert (rd("arrows_rgb.ppm") != pc).any()
    un = viz.normalize_minmax_u8(u)
    assert np.array_equal(rd("u_norm.pgm"), un) and un.min() == 0 and un.max() == 255
    assert np.array_equal(rd("u_jet.ppm"), viz.apply_colormap_jet(un))
    assert not rd("flat_norm.pgm").any()
    lut = viz.jet_lut()  # blue -> cyan -> yellow -> red, dark ends
    assert tuple(lut[0]) == (128, 0, 0) and tuple(lut[255]) == (0, 0, 128) and tuple(lut[96])[1] > 200


def test_line_is_an_8_connected_walk_between_its_end_points():
    from introtocomputervision_amd import viz
    rng = np.random.default_rng(2)
    for _ in range(200):
        img = np.zeros((40, 50), np.uint8)
        p1, p2 = (int(rng.integers(0, 50)), int(rng.integers(0, 40))), (int(rng.integers(0, 50)), int(rng.integers(0, 40)))
        viz.line(img, p1, p2, 255)
        ys, xs = np.nonzero(img)
        assert len(xs) == max(abs(p1[0] - p2[0]), abs(p1[1] - p2[1])) + 1
        assert img[p1[1], p1[0]] and img[p2[1], p2[0]]
        dist = np.abs((p2[0] - p1[0]) * (ys - p1[1]) - (p2[1] - p1[1]) * (xs - p1[0])) / max(1.0, np.hypot(p2[0] - p1[0], p2[1] - p1[1]))
        assert dist.max() <= 0.75  # every pixel within half a pixel (plus slack) of the ideal segment


@pytest.mark.gpu
@pytest.mark.parametrize("mode,win", [("pyr", 15), ("naive", 21)])
def test_ps5_demo_end_to_end(tmp_path, mode, win):
    """examples/ps5_demo: two colour frames on disk -> flow arrows + colour maps on disk, through the shim
    and libmicv.so, compared with the numpy mirror applied to the oracle's flow."""
    import _oracle as orc
    from introtocomputervision_amd import viz
    d = str(tmp_path)
    rows, cols = 180, 240
    prev, nxt, pc, nc = _frames(rows, cols, 0x5EED0005)
    viz.imwrite(os.path.join(d, "prev.ppm"), pc)
    viz.imwrite(os.path.join(d, "next.ppm"), nc)
    exe = os.path.join(d, "ps5_demo")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "examples", "ps5_demo.cpp"), "-o", exe,
                    "-L" + LIB, "-lmicv", "-Wl,-rpath," + LIB], check=True)
    r = subprocess.run([exe, os.path.join(d, "prev.ppm"), os.path.join(d, "next.ppm"), d, str(win), mode],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    gp, gn = orc.to_gray(pc), orc.to_gray(nc)   # cvtColor(COLOR_RGB2GRAY) on the stored channel order + convertTo(CV_32F)
    eu, ev = orc.lk_flow(gp, gn, win) if mode == "naive" else orc.lk_flow_pyr(gp, gn, win, 4)
    u = np.fromfile(os.path.join(d, "u.f32"), np.float32).reshape(rows, cols)
    v = np.fromfile(os.path.join(d, "v.f32"), np.float32).reshape(rows, cols)
    assert np.array_equal(u, eu) and np.array_equal(v, ev)
    assert np.array_equal(viz.imread(os.path.join(d, "flow.ppm")), viz.drawVelocityVectors(pc, eu, ev))
    assert np.array_equal(viz.imread(os.path.join(d, "flow-uColorMap.ppm")), viz.apply_colormap_jet(viz.normalize_minmax_u8(eu)))
    assert np.array_equal(viz.imread(os.path.join(d, "flow-vColorMap.ppm")), viz.apply_colormap_jet(viz.normalize_minmax_u8(ev)))
    pyr = viz.imread(os.path.join(d, "pyramid.pgm"))
    assert pyr.shape == (2 * rows, 2 * cols)
    lv = orc.gaussian_pyramid(gp, 4)
    assert np.array_equal(pyr[:rows, :cols], viz.normalize_minmax_u8(lv[0]))
    l3 = viz.normalize_minmax_u8(lv[3])  # cv::resize(INTER_NEAREST) back to level-0 size: src index = floor(dst * src / dst_size)
    iy = np.minimum(np.floor(np.arange(rows) * (l3.shape[0] / rows)).astype(int), l3.shape[0] - 1)
    ix = np.minimum(np.floor(np.arange(cols) * (l3.shape[1] / cols)).astype(int), l3.shape[1] - 1)
    assert np.array_equal(pyr[rows:, cols:], l3[iy][:, ix])


@pytest.mark.gpu
def test_ps5_demo_sequence_mode(tmp_path):
    """examples/ps5_demo --sequence: four colour frames on disk -> the flows of the three consecutive pairs through ONE
    call of the frame-sequence entry (micv_viz::denseLKSequence), each pair's files equal to the oracle's."""
    import _oracle as orc
    from introtocomputervision_amd import viz
    d = str(tmp_path)
    rows, cols = 120, 200
    frames = []
    for t in range(4):
        _, _, pc, nc = _frames(rows, cols, 0x5EED0005 + t)
        frames.append(pc if t % 2 == 0 else nc)
        viz.imwrite(os.path.join(d, f"f{t}.ppm"), frames[-1])
    exe = os.path.join(d, "ps5_demo")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "examples", "ps5_demo.cpp"), "-o", exe,
                    "-L" + LIB, "-lmicv", "-Wl,-rpath," + LIB], check=True)
    r = subprocess.run([exe, "--sequence", d, "15"] + [os.path.join(d, f"f{t}.ppm") for t in range(4)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for p in range(3):
        eu, ev = orc.lk_flow_pyr(orc.to_gray(frames[p]), orc.to_gray(frames[p + 1]), 15, 4)
        u = np.fromfile(os.path.join(d, f"u{p}.f32"), np.float32).reshape(rows, cols)
        v = np.fromfile(os.path.join(d, f"v{p}.f32"), np.float32).reshape(rows, cols)
        assert np.array_equal(u, eu) and np.array_equal(v, ev), p
        assert np.array_equal(viz.imread(os.path.join(d, f"flow{p}.ppm")), viz.drawVelocityVectors(frames[p], eu, ev))
