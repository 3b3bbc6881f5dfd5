"""CPU-side checks of the drop-in boundary: libstepth_amd.so loads, exports every symbol
include/stepth_amd.h declares, struct layouts agree, and the product never reaches into
oracle/.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from stepth_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "stepth_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(stepth_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound(lib):
    from stepth_amd import _lib
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/stepth_amd.h but not exported"
    assert sorted(_lib.SYMBOLS) == names, "ctypes binding and header disagree"


def test_abi_version_and_status_strings(lib):
    from stepth_amd import _lib
    assert lib.stepth_abi_version() == _lib.ABI_VERSION == 4
    # the reference's fixed error text, src/depth_image.rs:45-48
    assert lib.stepth_status_string(_lib.ERR_SIZE_MISMATCH) == b"Sizes don't match"
    assert lib.stepth_status_string(0) == b"ok"
    for st in (_lib.ERR_SPLIT_STEP, _lib.ERR_SPLIT_EMPTY_ZONE, _lib.ERR_SPLIT_DIVERGED):
        assert lib.stepth_status_string(st) != b"unknown status"


def test_struct_layouts():
    from stepth_amd import _lib
    from stepth_amd.engine import GROUP_DTYPE
    from oracle import oracle as O
    assert C.sizeof(_lib.Group) == 20 == GROUP_DTYPE.itemsize == O.GROUP_DTYPE.itemsize
    assert C.sizeof(_lib.Image) == 32
    assert C.sizeof(_lib.Params) == 56  # static_assert'ed in csrc/engine.hip
    assert C.sizeof(_lib.Stats) == 56
    assert C.sizeof(_lib.Zone) == 4
    assert _lib.Params.flags.offset == 40 and _lib.Params.max_steps.offset == 12


def test_null_arguments_fail_cleanly_without_a_device(lib):
    from stepth_amd import _lib
    assert lib.stepth_engine_create(0, None) == _lib.ERR_BAD_ARG
    assert lib.stepth_engine_sync(None) == _lib.ERR_BAD_ARG
    assert lib.stepth_last_error(None) == b"null engine"
    lib.stepth_engine_destroy(None)  # no-op


def test_no_exception_crosses_the_c_abi(lib):
    """include/stepth_amd.h: "calls never throw or abort".  Every status-returning entry runs
    inside the exception guard (source scan), and the guard turns a failed host allocation into
    STEPTH_ERR_OUT_OF_MEMORY and anything else into STEPTH_ERR_HIP (self-test entry, no device)"""
    from stepth_amd import _lib
    assert lib.stepth_selftest_exception_guard(None, 0) == _lib.ERR_OUT_OF_MEMORY
    assert lib.stepth_selftest_exception_guard(None, 1) == _lib.ERR_HIP
    assert lib.stepth_selftest_exception_guard(None, 2) == _lib.ERR_HIP
    assert lib.stepth_selftest_exception_guard(None, 3) == 0
    csrc = os.path.join(ROOT, "stepth_amd", "csrc")
    seen = set()
    for f in os.listdir(csrc):
        if not f.endswith((".hip", ".h")):
            continue
        text = open(os.path.join(csrc, f)).read()
        for m in re.finditer(r"^int (stepth_\w+)\([^;{]*\)\n\{\n(.*?)^\}", text, flags=re.S | re.M):
            seen.add(m.group(1))
            body = m.group(2)
            assert body.lstrip().startswith("STEPTH_GUARD_BEGIN") and "STEPTH_GUARD_END(" in body.rstrip().splitlines()[-1], m.group(1)
    status_entries = {n for n in declared_functions()
                      if n not in ("stepth_abi_version", "stepth_status_string", "stepth_last_error",
                                   "stepth_engine_destroy", "stepth_host_free")}
    assert seen == status_entries, (sorted(status_entries - seen), sorted(seen - status_entries))


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from stepth_amd import Engine, StepthError
    with pytest.raises(StepthError) as ei:
        Engine(0)
    assert ei.value.status in (-5, -3)


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "stepth_amd")
    for base, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="replace").read()
                for line in text.splitlines():
                    code = line.split("#")[0] if f.endswith(".py") else line
                    assert not re.search(r"(import|from|include|link|dlopen|CDLL).*oracle", code), \
                        f"{f}: {line.strip()}"


def test_depth_image_host_logic(tmp_path):
    from PIL import Image
    from stepth_amd import DepthImage, InvalidInput
    rgb = np.random.default_rng(0).integers(0, 255, (6, 9, 3), dtype=np.uint8)
    p = tmp_path / "m.png"
    Image.fromarray(rgb).save(p)
    img = DepthImage.open(str(p))
    assert img.image.shape == (6, 9, 4) and (img.image[..., 3] == 255).all()
    assert (img.image[..., :3] == rgb).all()
    assert img.depth.shape == (6, 9) and not img.depth.any()  # from_pixel(.., Luma([0]))
    assert (img.width(), img.height()) == (9, 6) == img.dimensions()
    img.load_depth(np.full((6, 9), 7, np.uint8))
    assert (img.depth == 7).all()
    with pytest.raises(InvalidInput, match="Sizes don't match"):
        img.load_depth(np.zeros((9, 6), np.uint8))
    with pytest.raises(InvalidInput, match="Failed to open image"):
        DepthImage.open(str(tmp_path / "missing.jpg"))
    with pytest.raises(InvalidInput, match="Failed to open image"):
        img.open_depth_from_additional(str(tmp_path / "missing.jpg"), [36, 36, 36])
    with pytest.raises(InvalidInput, match="Failed to open image"):
        img.open_depth(str(tmp_path / "missing.jpg"))
    d = DepthImage()
    assert d.image.shape == (0, 0, 4) and d.depth.shape == (0, 0)
    c = img.clone()
    c.depth[:] = 1
    assert (img.depth == 7).all()


def test_strip_bounds():
    from stepth_amd.sharding import strip_bounds
    assert strip_bounds(1080, 8) == [(135 * i, 135 * (i + 1)) for i in range(8)]
    b = strip_bounds(10, 4)
    assert b == [(0, 3), (3, 6), (6, 8), (8, 10)]
    b = strip_bounds(1080, 8, tile_h=5)
    assert b[0][0] == 0 and b[-1][1] == 1080
    assert all(a % 5 == 0 for a, _ in b) and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    assert strip_bounds(3, 8)[3:] == [(3, 3)] * 5
    assert strip_bounds(7, 2, tile_h=5) == [(0, 5), (5, 7)]


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """the boundary is a C ABI: include/stepth_amd.h must compile as C99 and a C program must
    link against libstepth_amd.so (what a cgo / Rust `extern "C"` binding relies on)"""
    import subprocess
    from stepth_amd import _lib
    hdr = os.path.join(ROOT, "include", "stepth_amd.h")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                        "-x", "c", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    src = tmp_path / "use.c"
    src.write_text('#include "stepth_amd.h"\n#include <stdio.h>\n'
                   'int main(void) { stepth_engine *e = 0; int rc = stepth_engine_create(-1, &e);\n'
                   '  printf("%u %d %s\\n", stepth_abi_version(), rc, stepth_status_string(rc));\n'
                   '  stepth_engine_destroy(e); return 0; }\n')
    exe = tmp_path / "use_c"
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-L", libdir,
                        "-lstepth_amd", f"-Wl,-rpath,{libdir}", "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.split()[0] == str(_lib.ABI_VERSION), out.stdout + out.stderr
