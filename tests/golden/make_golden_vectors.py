"""Generates tests/golden/vectors.npz: small seeded inputs and the outputs the CPU oracle
(oracle/, the restatement of the reference pinned by tests/test_oracle_asset.py and
tests/test_oracle_zones.py) gives for them.  Data only -- inputs and expected outputs; the
tests load it with numpy.load(allow_pickle=False).

The file guards two things: the GPU path against the oracle's answers even if the oracle
library is unavailable or changes, and the oracle itself against silent drift (the CPU suite
re-derives every vector).  Regenerate only after a deliberate, re-pinned change of the oracle:
    python tests/golden/make_golden_vectors.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from stepth_amd import synth  # noqa: E402

CASES = [  # name, width, height, d_max, seed, precision, max_steps, tile
    ("ring_1x1_200x120", 200, 120, 24, 11, [36, 36, 36], 255, (1, 1)),
    ("ring_1x1_short_reach_160x96", 160, 96, 16, 12, [30, 30, 30], 12, (1, 1)),
    ("ring_tiles5_203x117", 203, 117, 24, 13, [20, 25, 30], 64, (5, 5)),
    ("ring_tight_precision_96x64", 96, 64, 8, 14, [6, 6, 6], 255, (1, 1)),
]


def main():
    out = {}
    for name, w, h, d, seed, prec, ms, tile in CASES:
        l, r, _ = synth.stereo_pair(w, h, d, seed=seed)
        ref = O.depth_from_additional(l, r, prec, ms, tile)
        assert ref["status"] == 0
        out[f"{name}/main"], out[f"{name}/additional"] = l, r
        out[f"{name}/params"] = np.array(prec + [ms, tile[0], tile[1]], np.int64)
        out[f"{name}/depth"], out[f"{name}/raw"] = ref["depth"], ref["raw"]
        out[f"{name}/max"] = np.array([ref["max"]], np.int64)
        out[f"{name}/gaussian"] = O.gaussian_same_size(ref["depth"])
    # adaptive groups (own rule) + the path on them, and the consumers of the depth map
    name = "adaptive_groups_200x120"
    l, r, _ = synth.stereo_pair(200, 120, 24, seed=11)
    g = O.adaptive_groups(l, [36] * 3, 8, 0)
    g = g[np.lexsort((g["x"], g["y"]))]
    ref = O.depth_from_additional(l, r, [36] * 3, 255, groups=g)
    out[f"{name}/groups"] = np.stack([g["x"], g["y"], g["w"], g["h"], g["value"][:, 0].astype(np.uint32),
                                      g["value"][:, 1].astype(np.uint32), g["value"][:, 2].astype(np.uint32)], 1)
    out[f"{name}/depth"] = ref["depth"]
    inv = O.invert_depth(ref["depth"])
    fg = O.select_foreground(inv)
    assert fg["status"] == 0
    out["foreground_200x120/mask"] = fg["mask"]
    out["foreground_200x120/zone"] = np.array(fg["zone"], np.int64)
    z = O.depth_split(inv, 4)
    out["foreground_200x120/zones4"] = np.array(z["zones"], np.int64)
    # windowed-SAD mode (self-oracle)
    out["sad_5x5_d32_200x120/raw"] = O.sad_rows(l, r, 32, 5, [4, 4, 4])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
