"""The C++ host mirror of the reference's DepthImage (include/stepth_depth_image.hpp) through
its CLI (stepth_amd/stepth_depth_cli, built by __graft_entry__.build()): the README flow of
the reference (Readme.md:11-16) with PNM files, output compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

CLI = os.path.join(ROOT, "stepth_amd", "stepth_depth_cli")
CHECK = os.path.join(ROOT, "stepth_amd", "stepth_cabi_check")


@pytest.fixture(scope="module", autouse=True)
def built_programs():
    """the two host programs are build products (git-ignored): make them if a checkout or a
    snapshot came without them -- the same make as __graft_entry__.build()"""
    if not (os.path.exists(CLI) and os.path.exists(CHECK)):
        from stepth_amd import _lib
        _lib.build()


def write_pnm(path, a):
    a = np.ascontiguousarray(a, np.uint8)
    magic = b"P6" if a.ndim == 3 else b"P5"
    with open(path, "wb") as f:
        f.write(magic + b"\n# written by the test\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
        f.write(a.tobytes())


def read_pgm(path):
    data = open(path, "rb").read()
    parts = data.split(b"\n", 3)
    assert parts[0] == b"P5" and parts[2] == b"255"
    w, h = map(int, parts[1].split())
    return np.frombuffer(parts[3], np.uint8).reshape(h, w)


def test_cli_is_built_and_fails_loudly_without_a_device():
    import torch
    assert os.path.exists(CLI), "make -C stepth_amd/csrc did not produce the CLI"
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([CLI, "--check-errors"], capture_output=True, text=True)
    assert r.returncode != 0 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_cpp_mirror_error_contract():
    r = subprocess.run([CLI, "--check-errors"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "OK", r.stderr


@pytest.mark.gpu
def test_cpp_mirror_readme_flow_matches_oracle(oracle, asset, tmp_path):
    write_pnm(tmp_path / "main.ppm", asset["main"])
    write_pnm(tmp_path / "additional.ppm", asset["additional"])
    ref = oracle.depth_from_additional(asset["main"], asset["additional"], [36] * 3)
    out = tmp_path / "depth.pgm"
    r = subprocess.run([CLI, str(tmp_path / "main.ppm"), str(tmp_path / "additional.ppm"),
                        str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    # default = like the reference: precision u8::MAX/7 and the Gaussian tail (:133-134)
    np.testing.assert_array_equal(read_pgm(out), oracle.gaussian_same_size(ref["depth"]))
    r = subprocess.run([CLI, str(tmp_path / "main.ppm"), str(tmp_path / "additional.ppm"),
                        str(out), "36", "--no-gaussian"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    np.testing.assert_array_equal(read_pgm(out), ref["depth"])
    ref5 = oracle.depth_from_additional(asset["main"], asset["additional"], [20] * 3, 64, (5, 5))
    r = subprocess.run([CLI, str(tmp_path / "main.ppm"), str(tmp_path / "additional.ppm"),
                        str(out), "20", "--no-gaussian", "--max-steps", "64", "--tile", "5", "5"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    np.testing.assert_array_equal(read_pgm(out), ref5["depth"])
    r = subprocess.run([CLI, str(tmp_path / "missing.ppm"), str(tmp_path / "additional.ppm"),
                        str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "Failed to open image" in r.stderr


def read_ppm(path):
    data = open(path, "rb").read()
    parts = data.split(b"\n", 3)
    assert parts[0] == b"P6" and parts[2] == b"255"
    w, h = map(int, parts[1].split())
    return np.frombuffer(parts[3], np.uint8).reshape(h, w, 3)


@pytest.mark.gpu
def test_cpp_mirror_readme_second_half_matches_oracle(oracle, asset, tmp_path):
    """Readme.md:18-24: open_depth, invert_depth, select_foreground, apply_mask, save."""
    write_pnm(tmp_path / "main.ppm", asset["main"])
    write_pnm(tmp_path / "depth.pgm", asset["depth"])
    out = tmp_path / "foreground.ppm"
    r = subprocess.run([CLI, "--foreground", str(tmp_path / "main.ppm"), str(tmp_path / "depth.pgm"),
                        str(out), "--zones", "5"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    inv = oracle.invert_depth(asset["depth"])
    want_zones = oracle.depth_split(inv, 5)["zones"]
    got_zones = [tuple(map(int, l.split()[1:])) for l in r.stdout.splitlines() if l.startswith("zone")]
    assert got_zones == want_zones
    mask = oracle.select_foreground(inv)["mask"]
    rgba = np.concatenate([asset["main"], np.full(asset["main"].shape[:2] + (1,), 255, np.uint8)], 2)
    np.testing.assert_array_equal(read_ppm(out), oracle.apply_mask(rgba, mask)[..., :3])
    # a flat depth map: the reference panics in depth_split (src/depth_image.rs:217)
    write_pnm(tmp_path / "flat.pgm", np.full(asset["depth"].shape, 9, np.uint8))
    r = subprocess.run([CLI, "--foreground", str(tmp_path / "main.ppm"), str(tmp_path / "flat.pgm"),
                        str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "step_by" in r.stderr


@pytest.mark.gpu
def test_cpp_mirror_adaptive_groups_flow(oracle, asset, tmp_path):
    """--adaptive: groups from stepth_adaptive_groups (stand-in for disage, own design) with the
    reference's (16, ceil(log2(W*H))) splits, then the usual path; equals the oracle on the
    oracle's statement of the same grouping rule."""
    write_pnm(tmp_path / "main.ppm", asset["main"])
    write_pnm(tmp_path / "additional.ppm", asset["additional"])
    out = tmp_path / "depth.pgm"
    r = subprocess.run([CLI, str(tmp_path / "main.ppm"), str(tmp_path / "additional.ppm"), str(out),
                        "36", "--no-gaussian", "--adaptive"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    groups = oracle.adaptive_groups(asset["main"], [36] * 3)
    ref = oracle.depth_from_additional(asset["main"], asset["additional"], [36] * 3, 255, groups=groups)
    np.testing.assert_array_equal(read_pgm(out), ref["depth"])


@pytest.mark.gpu
def test_cabi_check_program():
    """stepth_amd/csrc/cabi_check.cpp: every entry point of the C-ABI from plain C++ (host batch
    pipeline == single calls, device-resident adaptive groups -> list mode, consumers, SAD,
    tiles, Gaussian, page-locked memory)"""
    r = subprocess.run([CHECK], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and "CABI-CHECK-OK" in r.stdout, r.stdout + r.stderr
