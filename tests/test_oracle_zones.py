"""Oracle for the consumers of a finished depth map: depth_split,
select_foreground, slice, invert_depth (src/depth_image.rs:162-247) and apply_mask
(src/mask_image.rs:205-213).

Pinned against the reference's second shipped result: Readme.md:18-24 turns assets/depth.jpg
into assets/foreground.jpg by open_depth -> invert_depth -> select_foreground -> apply_mask ->
save.  foreground.jpg is a JPEG of the masked main image, so that comparison is a TOLERANCE
test; the zone boundaries are exact known answers for the decoded depth.jpg bytes.  The
hand-derived cases pin the quirks of the reference's k-means.  No GPU."""
import numpy as np
import pytest


def _rgba(rgb):
    return np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=2)


def test_readme_second_half_reproduces_foreground_jpg(oracle, asset):
    d = oracle.invert_depth(asset["depth"])                     # Readme.md:21
    r = oracle.select_foreground(d)                             # Readme.md:22
    assert r["status"] == 0 and r["zone"] == (0, 201)
    out = oracle.apply_mask(_rgba(asset["main"]), r["mask"])    # Readme.md:23
    got = out[..., :3].astype(np.float64)
    want = asset["foreground"].astype(np.float64)
    assert np.corrcoef(got.ravel(), want.ravel())[0, 1] >= 0.99   # measured 0.9968
    assert np.abs(got - want).mean() <= 1.0                        # measured 0.528
    black = want.sum(-1) < 40                                      # JPEG-tolerant "pixel was cleared"
    assert ((r["mask"] == 0) == black).mean() >= 0.99              # measured 0.9963
    assert abs((r["mask"] == 255).mean() - 0.0982) < 1e-3


def test_asset_zone_known_answers(oracle, asset):
    d = oracle.invert_depth(asset["depth"])
    assert oracle.depth_split(d, 2) == {"status": 0, "zones": [(0, 201), (202, 255)], "iters": 7}
    assert oracle.depth_split(d, 3)["zones"] == [(0, 139), (140, 212), (213, 255)]
    assert oracle.depth_split(d, 5)["zones"] == [(0, 88), (89, 155), (156, 201), (202, 240),
                                                 (241, 255)]


def test_fewer_than_two_zones_is_none_none(oracle):
    d = np.arange(256, dtype=np.uint8)
    for z in (0, 1):
        assert oracle.depth_split(d, z) == {"status": 0, "zones": [(None, None)], "iters": 0}


def test_two_clusters_by_hand(oracle):
    # values 10 x4, 20 x2, 200 x3.  init: step = (200-10)/1 - 1 = 189 -> centres [10, 199]
    # pass 1: {10,20} -> 10 (|20-10| < |20-199|), {200} -> 199; means 80/6 = 13, 200
    # pass 2: centres [13, 200]: same clusters, means [13, 200] -> converged on pass 2
    d = np.array([10] * 4 + [20] * 2 + [200] * 3, np.uint8)
    r = oracle.depth_split(d, 2)
    assert r == {"status": 0, "zones": [(10, 20), (200, 200)], "iters": 2}
    s = oracle.select_foreground(d)
    assert s["zone"] == (10, 20) and s["mask"].tolist() == [255] * 6 + [0] * 3


def test_tie_goes_to_the_first_centre(oracle):
    # centres [0, 9] (step = 10/1 - 1 = 9; 10 itself is excluded, the range is half open).
    # pass 1: 4 -> 0 (|4-0| = 4 < 5), 10 -> 9; means [2, 10]
    # pass 2: centres [2, 10]; 0 -> 2, 4 -> 2, 10 -> 10; means [2, 10] -> converged
    assert oracle.depth_split(np.array([0, 4, 10], np.uint8), 2)["zones"] == [(0, 4), (10, 10)]
    # equidistant item: centres [0, 9], later [.., ..]; item 3 between centres 1 and 5
    # init for {1, 3, 5, 11}: step 9 -> [1, 10]; 3,5 -> 1 ; 11 -> 10; means [3, 11]
    # pass 2: 1,3,5 -> 3 (|5-3| = 2 < 6); converged [3, 11]
    assert oracle.depth_split(np.array([1, 3, 5, 11], np.uint8), 2)["zones"] == [(1, 5), (11, 11)]
    # a true tie: centres [0, 4] after init for {0, 2, 5}: step = 5 - 1 = 4 -> [0, 4];
    # 2 is 2 away from both -> first centre (0).  means: {0,2} -> 1, {5} -> 5; next pass
    # centres [1, 5]: 0,2 -> 1 ; 5 -> 5 -> converged
    assert oracle.depth_split(np.array([0, 2, 5], np.uint8), 2)["zones"] == [(0, 2), (5, 5)]


def test_reference_panics_are_statuses(oracle):
    O = oracle
    # (max - min) / (zones - 1) < 2: step_by(0) or usize underflow (src/depth_image.rs:217)
    assert O.depth_split(np.full(100, 7, np.uint8), 2)["status"] == O.SPLIT_ERR_STEP
    assert O.depth_split(np.array([3, 4], np.uint8), 2)["status"] == O.SPLIT_ERR_STEP
    assert O.depth_split(np.array([0, 255], np.uint8), 200)["status"] == O.SPLIT_ERR_STEP
    assert O.depth_split(np.array([3, 5], np.uint8), 2)["status"] == 0
    assert O.depth_split(np.zeros(0, np.uint8), 2)["status"] == O.SPLIT_ERR_INPUT
    # an empty zone at convergence (:203-206): {0, 255} with 3 zones -> centres [0, 126, 252];
    # 126 attracts nothing -> mean 0 -> sorted [0, 0, 255]; the duplicate 0 shares one cluster,
    # next means are [0, 255] == the first two of [0, 0, 255]?  no: (0,0), (0,255) differ ->
    # next pass centres [0, 255] -> converged with two zones
    assert O.depth_split(np.array([0, 255], np.uint8), 3)["zones"] == [(0, 0), (255, 255)]
    # centre 0 kept alive by an empty cluster while no pixel is near it: values {100, 250},
    # 3 zones: step = 150/2 - 1 = 74 -> [100, 174, 248]; 174 empty -> means [0, 100, 250];
    # pass 2: 0 attracts nothing again -> means [0, 100, 250] == centres -> converged with an
    # empty zone -> the reference unwraps None
    assert O.depth_split(np.array([100, 250], np.uint8), 3)["status"] == O.SPLIT_ERR_EMPTY


def test_centre_count_follows_the_step_not_zones(oracle):
    # (img_min..img_max).step_by(step) can yield more or fewer centres than `zones`
    d = np.arange(0, 256, dtype=np.uint8)
    for zones in (2, 3, 4, 7, 9, 16, 40, 100):
        step = 255 // (zones - 1) - 1
        centres = len(range(0, 255, step))
        r = oracle.depth_split(d, zones)
        assert r["status"] == 0 and len(r["zones"]) == centres, zones
        lo = [z[0] for z in r["zones"]]
        hi = [z[1] for z in r["zones"]]
        assert lo[0] == 0 and hi[-1] == 255
        assert all(hi[i] + 1 == lo[i + 1] for i in range(centres - 1))  # a partition of 0..255


def test_histogram_formulation_matches_item_loop(oracle):
    """The GPU works on the 256-bin histogram; this is the numpy statement of that
    formulation, checked here against the oracle's literal per-item loop."""
    rng = np.random.default_rng(11)
    for case in range(40):
        n = int(rng.integers(1, 4000))
        k = int(rng.integers(1, 6))
        centres = rng.integers(0, 256, k)
        d = np.clip(rng.normal(centres[rng.integers(0, k, n)], rng.integers(1, 30)), 0, 255)
        d = d.astype(np.uint8)
        zones = int(rng.integers(2, 12))
        assert _split_by_histogram(d, zones) == _strip(oracle.depth_split(d, zones)), case


def _strip(r):
    return (r["status"], r["zones"])


def _split_by_histogram(d, zones):
    return _zones_from_counts(np.bincount(d.ravel(), minlength=256), zones)


def _zones_from_counts(hist, zones):
    vals = np.nonzero(hist)[0]
    mn, mx = int(vals[0]), int(vals[-1])
    q = (mx - mn) // (zones - 1)
    if q < 2:
        return (-7, [])
    prev = list(range(mn, mx, q - 1))
    for _ in range(4096):
        members = {c: [] for c in prev}
        for v in vals:
            best = min(prev, key=lambda c: abs(c - int(v)))  # min() keeps the first minimum
            members[best].append(int(v))
        new = sorted(sum(v * int(hist[v]) for v in vs) // max(sum(int(hist[v]) for v in vs), 1)
                     for vs in members.values())
        if all(a == b for a, b in zip(prev, new)):
            if any(not members[c] for c in new):
                return (-8, [])
            return (0, [(min(members[c]), max(members[c])) for c in new])
        prev = new
    return (-9, [])


def test_slice_and_invert_and_apply(oracle):
    d = np.arange(256, dtype=np.uint8).reshape(16, 16)
    assert (oracle.invert_depth(d) == 255 - d).all()
    m = oracle.slice_mask(d, 10, 20)
    assert ((m == 255) == ((d >= 10) & (d <= 20))).all() and set(np.unique(m)) == {0, 255}
    assert (oracle.slice_mask(d, None, 20) == np.where(d <= 20, 255, 0)).all()
    assert (oracle.slice_mask(d, 10, None) == np.where(d >= 10, 255, 0)).all()
    assert (oracle.slice_mask(d, None, None) == 255).all()
    assert (oracle.slice_mask(d, 30, 20) == 0).all()   # from > to: nothing inside
    rgba = np.random.default_rng(3).integers(1, 256, (16, 16, 4)).astype(np.uint8)
    mask = m.copy()
    mask[0, 0] = 7   # only an exact 0 clears (src/mask_image.rs:208 compares with MASK_FALSE)
    out = oracle.apply_mask(rgba, mask)
    assert (out[mask == 0] == 0).all() and (out[mask != 0] == rgba[mask != 0]).all()
