"""Committed golden vectors (tests/golden/vectors.npz, made by tests/golden/make_golden_vectors.py
from the pinned oracle): the CPU suite re-derives them with the oracle (guards the oracle
against drift), the GPU suite reproduces them through the C-ABI without trusting the oracle
library of the day."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

RING = ["ring_1x1_200x120", "ring_1x1_short_reach_160x96", "ring_tiles5_203x117",
        "ring_tight_precision_96x64"]


@pytest.fixture(scope="module")
def vec():
    with np.load(os.path.join(GOLDEN, "vectors.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def _params(vec, name):
    p = [int(v) for v in vec[f"{name}/params"]]
    return p[:3], p[3], (p[4], p[5])


def _groups(vec):
    from oracle.oracle import GROUP_DTYPE
    a = vec["adaptive_groups_200x120/groups"]
    g = np.zeros(len(a), GROUP_DTYPE)
    g["x"], g["y"], g["w"], g["h"] = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    g["value"] = a[:, 4:7].astype(np.uint8)
    return g


@pytest.mark.parametrize("name", RING)
def test_oracle_reproduces_golden(oracle, vec, name):
    prec, ms, tile = _params(vec, name)
    ref = oracle.depth_from_additional(vec[f"{name}/main"], vec[f"{name}/additional"], prec, ms, tile)
    assert ref["max"] == int(vec[f"{name}/max"][0])
    assert np.array_equal(ref["raw"], vec[f"{name}/raw"])
    assert np.array_equal(ref["depth"], vec[f"{name}/depth"])
    assert np.array_equal(oracle.gaussian_same_size(ref["depth"]), vec[f"{name}/gaussian"])


def test_oracle_reproduces_golden_groups_and_consumers(oracle, vec):
    main, add = vec["ring_1x1_200x120/main"], vec["ring_1x1_200x120/additional"]
    g = oracle.adaptive_groups(main, [36] * 3, 8, 0)
    g = g[np.lexsort((g["x"], g["y"]))]
    want = _groups(vec)
    assert len(g) == len(want) and (g == want).all()
    depth = oracle.depth_from_additional(main, add, [36] * 3, 255, groups=g)["depth"]
    assert np.array_equal(depth, vec["adaptive_groups_200x120/depth"])
    inv = oracle.invert_depth(depth)
    fg = oracle.select_foreground(inv)
    assert fg["zone"] == tuple(vec["foreground_200x120/zone"])
    assert np.array_equal(fg["mask"], vec["foreground_200x120/mask"])
    assert oracle.depth_split(inv, 4)["zones"] == [tuple(z) for z in vec["foreground_200x120/zones4"]]
    assert np.array_equal(oracle.sad_rows(main, add, 32, 5, [4, 4, 4]), vec["sad_5x5_d32_200x120/raw"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", RING)
def test_gpu_reproduces_golden(engine, vec, name):
    prec, ms, tile = _params(vec, name)
    depth, mx = engine.depth_from_additional(vec[f"{name}/main"], vec[f"{name}/additional"], prec,
                                             max_steps=ms, tile=tile)
    assert mx == int(vec[f"{name}/max"][0]) and np.array_equal(depth, vec[f"{name}/depth"])
    raw, _ = engine.depth_from_additional(vec[f"{name}/main"], vec[f"{name}/additional"], prec,
                                          max_steps=ms, tile=tile, raw_only=True)
    assert np.array_equal(raw, vec[f"{name}/raw"])
    tail, _ = engine.depth_from_additional(vec[f"{name}/main"], vec[f"{name}/additional"], prec,
                                           max_steps=ms, tile=tile, gaussian_tail=True)
    assert np.array_equal(tail, vec[f"{name}/gaussian"])


@pytest.mark.gpu
def test_gpu_reproduces_golden_groups_and_consumers(engine, vec):
    from stepth_amd import MODE_SAD_WINDOW
    main, add = vec["ring_1x1_200x120/main"], vec["ring_1x1_200x120/additional"]
    g = engine.adaptive_groups(main, [36] * 3, 8, 0)
    g = g[np.lexsort((g["x"], g["y"]))]
    want = _groups(vec)
    assert len(g) == len(want) and all((g[f] == want[f]).all() for f in ("x", "y", "w", "h", "value"))
    depth, _ = engine.depth_from_additional(main, add, [36] * 3, groups=g)
    assert np.array_equal(depth, vec["adaptive_groups_200x120/depth"])
    inv = depth.copy()
    engine.invert_depth(inv)
    mask, zone = engine.select_foreground(inv)
    assert zone == tuple(vec["foreground_200x120/zone"]) and np.array_equal(mask, vec["foreground_200x120/mask"])
    assert engine.depth_split(inv, 4) == [tuple(z) for z in vec["foreground_200x120/zones4"]]
    raw, _ = engine.depth_from_additional(main, add, [4, 4, 4], mode=MODE_SAD_WINDOW, d_max=32, win=5,
                                          raw_only=True)
    assert np.array_equal(raw, vec["sad_5x5_d32_200x120/raw"])
