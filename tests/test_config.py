"""The run-configuration reader (config/psN.yaml of the reference; SURVEY.md §2 marks the key reader
IN as a data contract) and BASELINE config C1 driven through it: "ps4 Harris corners on a single
640x480 greyscale frame via config/ps4.yaml"."""
import os
import subprocess

import numpy as np
import pytest

import _oracle as orc
from introtocomputervision_amd import config, synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CFG = os.path.join(HERE, "golden", "config")


def test_reader_agrees_with_pyyaml():
    yaml = pytest.importorskip("yaml")
    for name in ("ps4.yaml", "pipeline.yaml"):
        path = os.path.join(CFG, name)
        mine = config.load(path)
        ref = yaml.safe_load(open(path))
        assert set(mine) == set(ref)
        for k, v in ref.items():
            if isinstance(v, dict):
                assert set(mine.child(k)) == set(v)
                for kk, vv in v.items():
                    got = mine.child(k)[kk]
                    assert float(got) == float(vv) if isinstance(vv, (int, float)) and not isinstance(vv, bool) else got == str(vv)
            elif isinstance(v, bool):
                assert mine.as_bool(k) == v
            elif isinstance(v, (int, float)):
                assert mine.as_float(k) == float(v)
            else:
                assert mine.as_str(k) == v


def test_typed_sections_and_errors():
    cfg = config.load(os.path.join(CFG, "ps4.yaml"))
    assert config.harris_params(cfg, "harris_trans") == {
        "sobel_kernel_size": 3, "window_size": 5, "gaussian_sigma": 1.5, "alpha": 0.04,
        "response_threshold": 5e8, "min_distance": 5}
    assert config.harris_params(cfg, "harris_sim")["response_threshold"] == 2e9
    assert cfg.as_bool("use_gpu") is True and cfg.as_str("output_dir") == "./ps4_output"
    assert [int(w, 16) for w in cfg.as_str("mersenne_seed").split()][:3] == [0x16, 0x38, 0xC7]  # Config.cpp:86-93
    p = config.load(os.path.join(CFG, "pipeline.yaml"))
    assert config.edge_params(p, "edge_detector_p2")["gaussian_sigma"] == 1.4
    assert config.hough_params(p, "hough_transform_p2") == {"rho_bin_size": 1, "theta_bin_size": 1, "num_peaks": 6, "threshold": 40}
    assert config.disparity_params(p, "problem_2_ssd") == {"window_radius": 5, "disparity_range": 30}
    assert p.as_int("lk_window_size_4") == 15 and p.as_int("pyr_level_3-b") == 2 and p.as_str("output_dir") == "./out dir"
    assert p.as_bool("use_gpu_disparity") is False
    with pytest.raises(config.ConfigError):
        cfg.as_int("output_dir")
    with pytest.raises(config.ConfigError):
        cfg.child("use_gpu")
    with pytest.raises(config.ConfigError):
        cfg.as_float("missing")
    assert config.loads("a:\n  b:\n    c: 1\n").child("a").child("b").as_int("c") == 1
    assert config.loads("a:\n- 1\n- 2\nb: 3\n").seq("a") == ["1", "2"]  # a sequence at its key's own indentation
    for bad in ("  orphan: 1\n", "a:\n  b: 1\n c: 2\n", "a: 1\n  b: 2\n", "a:\n  - x: 1\n", "a:\n  - 1\n    - 2\n",
                "a:\n\tb: 1\n", "- 1\n", "a:\n  - 1\n  b: 2\n", "just text\n"):
        with pytest.raises(config.ConfigError):
            config.loads(bad)


REF = os.path.join(CFG, "ref")  # byte copies of /root/reference/config/ps0..7.yaml (data files of the reference)
REF_FILES = [f"ps{i}.yaml" for i in range(8)]


def _same(mine, ref, path=""):
    if isinstance(ref, dict):
        assert isinstance(mine, config.Node) and set(mine) == {str(k) for k in ref}, path
        for k, v in ref.items():
            _same(mine[str(k)], v, f"{path}/{k}")
    elif isinstance(ref, list):
        assert isinstance(mine, list) and len(mine) == len(ref), path
        for a, b in zip(mine, ref):
            _same(a, b, path + "[]")
    elif isinstance(ref, bool):
        assert mine.lower() == str(ref).lower(), path
    elif isinstance(ref, (int, float)):
        assert float(mine) == float(ref), path
    else:
        assert mine == str(ref).strip(), path


@pytest.mark.parametrize("name", REF_FILES)
def test_reference_config_files_load_and_agree_with_pyyaml(name):
    """Every run configuration the reference ships loads, and means what a full YAML parser says it means
    (ps7.yaml nests lists under a three-level map, config/ps7.yaml:7-40)."""
    yaml = pytest.importorskip("yaml")
    path = os.path.join(REF, name)
    _same(config.load(path), yaml.safe_load(open(path)))


def test_reference_config_typed_sections():
    """The typed views the reference's Config classes take of ps1 / ps2 / ps4 / ps5 / ps7, with the values
    the files hold ("Config / flags")."""
    c1 = config.load(os.path.join(REF, "ps1.yaml"))
    assert config.edge_params(c1, "edge_detector_p2") == {"gaussian_size": 1, "gaussian_sigma": 0.0001, "lower_threshold": 1,
                                                          "upper_threshold": 3, "sobel_aperture_size": 3}
    assert config.edge_params(c1, "edge_detector_p6")["gaussian_size"] == 7
    assert config.hough_params(c1, "hough_transform_p2") == {"rho_bin_size": 1, "theta_bin_size": 1, "num_peaks": 6, "threshold": 200}
    assert config.hough_params(c1, "hough_transform_p6") == {"rho_bin_size": 2, "theta_bin_size": 3, "num_peaks": 10, "threshold": 80}
    assert config.hough_circle_params(c1, "hough_circle_transform_p5") == {"min_radius": 20, "max_radius": 50, "num_peaks": 10, "threshold": 130}
    assert c1.child("images").as_str("input0_noise") == "../Resources/ProblemSet1/ps1-input0-noise.png"
    c2 = config.load(os.path.join(REF, "ps2.yaml"))
    assert c2.as_bool("use_gpu_disparity") is True
    assert config.disparity_params(c2, "problem_1_ssd") == {"window_radius": 6, "disparity_range": 3}
    assert config.disparity_params(c2, "problem_2_ssd") == {"window_radius": 7, "disparity_range": 95}
    assert config.disparity_params(c2, "problem_5_ncorr") == {"window_radius": 7, "disparity_range": 80}
    c4 = config.load(os.path.join(REF, "ps4.yaml"))
    for section in ("harris_trans", "harris_sim"):
        assert config.harris_params(c4, section) == {"sobel_kernel_size": 3, "window_size": 5, "gaussian_sigma": 1.5, "alpha": 0.04,
                                                     "response_threshold": 5e8, "min_distance": 5}
    assert c4.as_bool("use_gpu") is True and len(c4.as_str("mersenne_seed").split()) == 16
    assert c4.child("ransac_sim").as_float("consensus_ratio") == 0.6
    c5 = config.load(os.path.join(REF, "ps5.yaml"))
    assert config.lk_params(c5) == {"lk_window_size_1": 43, "lk_window_size_3": 7, "pyr_level_3-a": 1, "pyr_level_3-b": 2,
                                    "lk_window_size_4": 15}
    assert c5.child("image_sets").as_str("shift") == "../Resources/ProblemSet5/TestSeq"
    c7 = config.load(os.path.join(REF, "ps7.yaml"))
    frames = config.last_frames(c7)  # ps7_cpp/lib/Config.cpp:49-66
    assert len(frames) == 27 and frames["PS7A1P1T1"] == 50 and frames["PS7A2P2T3"] == 23 and frames["PS7A3P3T3"] == 19
    assert config.mhi_params(c7, "mhi_action1") == {"diff_threshold": 1.7, "pre_blur_size": 31, "pre_blur_sigma": 10.0, "tau": 25}
    assert config.mhi_params(c7, "mhi_action3")["tau"] == 34


def test_cpp_reader_agrees_on_the_reference_files(tmp_path):
    """shim/micv_config.hpp (what a psN-style main() includes) reads the same leaves from the same files."""
    exe = str(tmp_path / "config_dump")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(HERE, "cpp", "config_dump.cpp"), "-o", exe], check=True)
    for path in [os.path.join(REF, n) for n in REF_FILES] + [os.path.join(CFG, "ps4.yaml"), os.path.join(CFG, "pipeline.yaml")]:
        r = subprocess.run([exe, path], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert r.stdout == config.dumps(config.load(path)), path
    r = subprocess.run([exe, "--ps7", os.path.join(REF, "ps7.yaml")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    frames = config.last_frames(config.load(os.path.join(REF, "ps7.yaml")))
    assert lines[:-1] == [f"{k} {v}" for k, v in sorted(frames.items())] and lines[-1] == "mhi_action3 1.7 31 10 34"
    for bad in ("  orphan: 1\n", "a:\n  b: 1\n c: 2\n", "a: 1\n  b: 2\n", "a:\n  - x: 1\n", "a:\n\tb: 1\n", "- 1\n"):
        f = tmp_path / "bad.yaml"
        f.write_text(bad)
        r = subprocess.run([exe, str(f)], capture_output=True, text=True)
        assert r.returncode == 1 and "config: line" in r.stderr, bad


def build_ps4_demo(tmp):
    exe = os.path.join(tmp, "ps4_demo")
    lib = os.path.join(ROOT, "introtocomputervision_amd")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", os.path.join(HERE, "cpp", "ps4_demo.cpp"),
                    "-o", exe, "-L" + lib, "-lmicv", "-Wl,-rpath," + lib], check=True)
    return exe


def test_ps4_demo_compiles_and_reports_a_bad_config(tmp_path):
    exe = build_ps4_demo(str(tmp_path))
    bad = tmp_path / "bad.yaml"
    bad.write_text("harris_trans:\n  sobel_kernel_size: three\n")
    r = subprocess.run([exe, str(bad), "harris_trans", str(tmp_path), "4", "4"], capture_output=True, text=True)
    assert r.returncode == 1 and "not an integer" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("section", ["harris_trans", "harris_sim"])
def test_c1_harris_via_config(tmp_path, section):
    """BASELINE C1: 480x640 greyscale checkerboard (+-4 noise), parameters read from the ps4-format
    configuration, through the C++ shim (the reference's executable path) and through the Python
    mirror; response and corner list bit-exact against the oracle."""
    import torch
    from introtocomputervision_amd import harris
    rows, cols = 480, 640
    img = synth.checkerboard(rows, cols, square=40, seed=0x5EED0001)
    cfg_path = os.path.join(CFG, "ps4.yaml")
    hp = config.harris_params(config.load(cfg_path), section)
    gx, gy = orc.sobel(img, hp["sobel_kernel_size"], 1.0)
    eR = orc.harris_response(gx, gy, hp["window_size"], hp["gaussian_sigma"], hp["alpha"])
    _, el = orc.harris_refine(eR, hp["response_threshold"], hp["min_distance"])
    assert len(el) > 50
    # Python mirror
    d = torch.from_numpy(img).cuda()
    dgx, dgy = harris.getGradients(d, hp["sobel_kernel_size"])
    R = harris.getCornerResponse(dgx, dgy, hp["window_size"], hp["gaussian_sigma"], hp["alpha"])
    _, locs = harris.refineCorners(R, hp["response_threshold"], hp["min_distance"])
    assert np.array_equal(R.cpu().numpy(), eR) and np.array_equal(locs.cpu().numpy(), el)
    # the C++ executable path: config file -> shim -> C ABI
    exe = build_ps4_demo(str(tmp_path))
    img.tofile(tmp_path / "img.f32")
    r = subprocess.run([exe, cfg_path, section, str(tmp_path), str(rows), str(cols)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert f"window {hp['window_size']} " in r.stdout and "use_gpu 1" in r.stdout
    assert np.array_equal(np.fromfile(tmp_path / f"{section}_R.f32", np.float32).reshape(rows, cols), eR)
    assert np.array_equal(np.fromfile(tmp_path / f"{section}_locs.i32", np.int32).reshape(-1, 2), el)
