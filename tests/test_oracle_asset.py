"""Pins the oracle against the reference's only shipped result: assets/main.jpg +
assets/additional.jpg -> assets/depth.jpg at precision [36;3] (Readme.md:13-16,28-34).

depth.jpg is lossy, post-Gaussian and of unknown commit, and the inputs were decoded with
Pillow/libjpeg-turbo rather than jpeg-decoder 0.1.22, so the comparison with depth.jpg is a
TOLERANCE test; the histogram / max / mean figures are exact known answers for these decoded
bytes.  No GPU."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def result(oracle, asset):
    return oracle.depth_from_additional(asset["main"], asset["additional"], [36, 36, 36],
                                        max_steps=255, tile=(1, 1))


def test_asset_tolerance_vs_depth_jpg(result, asset):
    d = result["depth"].astype(np.float64)
    ref = asset["depth"].astype(np.float64)
    r = np.corrcoef(d.ravel(), ref.ravel())[0, 1]
    mae = np.abs(d - ref).mean()
    assert result["status"] == 0
    assert r >= 0.975, r          # SURVEY.md 8c acceptance; measured 0.9811
    assert mae <= 1.5, mae        # measured 1.222
    assert (np.abs(d - ref) <= 4).mean() >= 0.94  # measured 0.949


def test_asset_known_answers(result, oracle, asset):
    assert result["max"] == 205  # raw distance max, no u8 wrap on this pair
    assert abs(result["depth"].mean() - 12.838) < 5e-3
    g = oracle.uniform_groups(asset["main"], 1, 1)
    m = oracle.match_groups(g, asset["additional"], 255, [36, 36, 36])
    assert m["found"].all()
    cx, cy = (g["x"] + g["w"]) // 2, (g["y"] + g["h"]) // 2
    ring = np.maximum(np.abs(m["pos"][:, 0].astype(int) - cx),
                      np.abs(m["pos"][:, 1].astype(int) - cy))
    hist = [int(((ring >= lo) & (ring <= hi)).sum())
            for lo, hi in [(0, 0), (1, 4), (5, 16), (17, 64), (65, 128), (129, 10 ** 6)]]
    assert hist == [175117, 12472, 18654, 23554, 8769, 1434]
    assert np.percentile(ring, [50, 75, 90, 95, 99, 100]).tolist() == [0, 2, 34, 62, 89, 171]


def test_rejected_hypotheses_stay_rejected(asset):
    """a true-centre `middle` does not reproduce depth.jpg (r = 0.37); guards
    against 'fixing' the reference's half-coordinate middle."""
    from oracle import oracle as O
    main, add = asset["main"], asset["additional"]
    g = O.uniform_groups(main, 1, 1)
    g2 = g.copy()
    g2["x"] = g["x"] * 2  # (2x + 1)/2 = x: search centred on the pixel itself
    g2["y"] = g["y"] * 2
    m = O.match_groups(g2, add, 255, [36, 36, 36])
    rc, out, _ = O.normalise(m["raw"])
    r = np.corrcoef(out.astype(float), asset["depth"].astype(float).ravel())[0, 1]
    assert r < 0.6


@pytest.fixture(scope="module")
def variants(tmp_path_factory, asset):
    """normalised depth maps of the asset pair under alternative readings of disage's
    PixelOpps::{substract,lt} (source absent; call site src/helpers.rs:35), from
    tests/tools/predicate_variants.c -- a literal ring scan with the predicate swapped"""
    import ctypes as C
    import os
    import subprocess
    from conftest import ROOT
    out = str(tmp_path_factory.mktemp("pv") / "predicate_variants.so")
    subprocess.run(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC",
                    os.path.join(ROOT, "tests", "tools", "predicate_variants.c"), "-o", out, "-lm"],
                   check=True)
    lib = C.CDLL(out)
    lib.variant_raw.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    main = np.ascontiguousarray(asset["main"][..., :3])
    add = np.ascontiguousarray(asset["additional"][..., :3])
    ref = asset["depth"].astype(np.float64)
    res = {}
    for v in range(5):
        raw = np.zeros(main.shape[:2], np.uint8)
        p = (C.c_uint8 * 3)(36, 36, 36)
        lib.variant_raw(main.ctypes.data, main.shape[1], main.shape[0], add.ctypes.data, add.shape[1],
                        add.shape[0], C.cast(p, C.c_void_p), 255, v, raw.ctypes.data)
        d = (raw.astype(np.uint32) * 255 // max(int(raw.max()), 1)).astype(np.float64)
        res[v] = {"raw": raw, "r": np.corrcoef(d.ravel(), ref.ravel())[0, 1],
                  "mae": np.abs(d - ref).mean()}
    return res


def test_pinned_predicate_is_the_literal_scan(result, variants):
    """variant 0 of the helper (|v-w| < p on all channels) is the oracle, byte for byte"""
    assert np.array_equal(variants[0]["raw"], result["raw"])
    assert abs(variants[0]["r"] - 0.9811) < 5e-4 and abs(variants[0]["mae"] - 1.222) < 5e-3


@pytest.mark.parametrize("variant,name,r_measured", [
    (1, "saturating subtraction instead of |v-w|", 0.9505),
    (2, "any channel below precision instead of all", 0.6577),
    (3, "channel sum below the summed precision", 0.9106),
])
def test_rejected_predicates_fit_depth_jpg_worse(variants, variant, name, r_measured):
    """the three other readings of substract/lt reproduce assets/depth.jpg
    clearly worse than the pinned one (r 0.9811, MAE 1.222)"""
    v, ref = variants[variant], variants[0]
    assert abs(v["r"] - r_measured) < 2e-3, (name, v["r"])
    assert v["r"] < ref["r"] - 0.025 and v["mae"] > ref["mae"] + 1.0, name


def test_strict_less_than_is_weakly_preferred_over_less_or_equal(variants):
    """`lt` read as <= : r 0.9796 / MAE 1.314 against r 0.9811 / MAE 1.222 for strict <.  The
    asset prefers the strict reading, but only by a JPEG-noise-sized margin: `<` is pinned by
    the method's name (`lt`, src/helpers.rs:35) and merely not contradicted by the asset."""
    lt, le = variants[0], variants[4]
    assert lt["r"] > le["r"] and lt["mae"] < le["mae"]
    assert lt["r"] - le["r"] < 0.005  # ... weakly: do not read more into it


def test_gaussian_tail_improves_the_fit(result, oracle, asset):
    """src/depth_image.rs:133-134: the same-size Gaussian resample (image 0.23.14's sampler
    restated; PARITY UNPINNED).  With it the oracle matches depth.jpg more closely, which is
    consistent with -- though no proof of -- the restatement being right."""
    ref = asset["depth"].astype(np.float64)
    bare = result["depth"].astype(np.float64)
    tail = oracle.gaussian_same_size(result["depth"]).astype(np.float64)
    r0 = np.corrcoef(bare.ravel(), ref.ravel())[0, 1]
    r1 = np.corrcoef(tail.ravel(), ref.ravel())[0, 1]
    assert r1 > r0 and r1 >= 0.985                      # measured 0.9811 -> 0.9863
    assert np.abs(tail - ref).mean() < np.abs(bare - ref).mean() <= 1.5   # 1.222 -> 1.071


def test_adaptive_grouping_stand_in_keeps_the_fit(oracle, asset):
    """The reference groups pixels with disage (source absent).  The engine's own adaptive
    stand-in with the reference's (16, ceil(log2(W*H))) splits reproduces depth.jpg at least as
    well as the pinned 1x1 configuration (r 0.9835 / MAE 1.129 bare, r 0.9880 / MAE 0.974 with
    the Gaussian tail, against 0.9811 / 1.222 and 0.9863 / 1.071) -- consistent with, not proof
    of, the stand-in resembling disage; parity stays unpinned."""
    g = oracle.adaptive_groups(asset["main"], [36, 36, 36])
    assert len(g) == 67115 and int((g["w"].astype(int) * g["h"]).sum()) == 240000
    r = oracle.depth_from_additional(asset["main"], asset["additional"], [36, 36, 36], 255, groups=g)
    ref = asset["depth"].astype(np.float64)
    d = r["depth"].astype(np.float64)
    assert np.corrcoef(d.ravel(), ref.ravel())[0, 1] >= 0.98
    assert np.abs(d - ref).mean() <= 1.2
    dg = oracle.gaussian_same_size(r["depth"]).astype(np.float64)
    assert np.corrcoef(dg.ravel(), ref.ravel())[0, 1] >= 0.985
    assert np.abs(dg - ref).mean() <= 1.05
