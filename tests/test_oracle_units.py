"""Hand-derived known answers for the CPU oracle (oracle/stepth_oracle.c), i.e. for the
reference semantics of /root/reference/src/helpers.rs:3-54 and src/depth_image.rs:111-129.
No GPU.  Each case states the reference line it pins."""
import numpy as np
import pytest

P = [10, 10, 10]
HIT = [200, 200, 200]


def blank(h, w):
    return np.zeros((h, w, 3), np.uint8)


def scan_order(cx, cy, s):
    """helpers.rs:28-31 written out: rows y+s, y-s (x ascending), then columns x+s, x-s
    (y ascending); duplicates kept."""
    out = []
    for i in (cy + s, cy - s):
        out += [(j, i) for j in range(cx - s, cx + s + 1)]
    for i in (cx + s, cx - s):
        out += [(i, j) for j in range(cy - s, cy + s + 1)]
    return out


def test_distance_truncates(oracle):
    # helpers.rs:3-7: sqrt as f64, `as u32` truncates
    assert oracle.distance_dot_dot((0, 0), (1, 1)) == 1
    assert oracle.distance_dot_dot((0, 0), (2, 2)) == 2
    assert oracle.distance_dot_dot((0, 0), (3, 4)) == 5
    assert oracle.distance_dot_dot((10, 10), (7, 6)) == 5
    assert oracle.distance_dot_dot((0, 0), (254, 254)) == 359  # max possible on the path
    assert oracle.distance_dot_dot((5, 5), (5, 5)) == 0
    for k in range(0, 360):  # perfect squares must not round down
        assert oracle.distance_dot_dot((0, 0), (k, 0)) == k


@pytest.mark.parametrize("s", [1, 2])
def test_single_match_at_every_scan_position(oracle, s):
    # only one cell matches; wherever it sits on ring s it is found, with its distance
    cx = cy = 3
    for (x, y) in set(scan_order(cx, cy, s)):
        a = blank(7, 7)
        a[y, x] = HIT
        res = oracle.distance_dot_array(HIT, a, (cx, cy), 255, P)
        assert res == (int(np.floor(np.hypot(x - cx, y - cy))), (x, y))


@pytest.mark.parametrize("s", [1, 2, 3])
def test_first_hit_is_scan_order_not_nearest(oracle, s):
    # two matches on one ring: the earlier one in scan order wins even when the later one is
    # closer in Euclidean distance (helpers.rs:28-41); pairs enumerated exhaustively
    cx = cy = 4
    order = scan_order(cx, cy, s)
    first_index = {}
    for i, c in enumerate(order):
        first_index.setdefault(c, i)
    cells = sorted(first_index, key=first_index.get)
    for ia in range(0, len(cells), 3):
        for ib in range(ia + 1, len(cells), 2):
            a = blank(9, 9)
            a[cells[ia][1], cells[ia][0]] = HIT
            a[cells[ib][1], cells[ib][0]] = HIT
            res = oracle.distance_dot_array(HIT, a, (cx, cy), 255, P)
            assert res[1] == cells[ia]


def test_row_before_column_and_plus_before_minus(oracle):
    cx = cy = 3
    a = blank(7, 7)
    a[cy + 1, cx] = HIT  # row y+s
    a[cy - 1, cx] = HIT  # row y-s
    a[cy, cx + 1] = HIT  # col x+s
    a[cy, cx - 1] = HIT  # col x-s
    assert oracle.distance_dot_array(HIT, a, (cx, cy), 255, P)[1] == (cx, cy + 1)
    a[cy + 1, cx] = 0
    assert oracle.distance_dot_array(HIT, a, (cx, cy), 255, P)[1] == (cx, cy - 1)
    a[cy - 1, cx] = 0
    assert oracle.distance_dot_array(HIT, a, (cx, cy), 255, P)[1] == (cx + 1, cy)
    a[cy, cx + 1] = 0
    assert oracle.distance_dot_array(HIT, a, (cx, cy), 255, P)[1] == (cx - 1, cy)


def test_inner_ring_beats_outer_ring(oracle):
    a = blank(9, 9)
    a[4 + 3, 4] = HIT      # ring 3, very early in that ring's scan
    a[4 - 2, 4 - 2] = HIT  # ring 2
    assert oracle.distance_dot_array(HIT, a, (4, 4), 255, P) == (2, (2, 2))


def test_probe_counts_keep_duplicates(oracle):
    # ring s costs 8s+4 get2d calls: ring 0 visits the centre 4x, corners twice (helpers.rs:28-31)
    a = blank(21, 21)
    res, probes, cells = oracle.distance_dot_array(HIT, a, (10, 10), 4, P, count=True)
    assert res is None
    assert cells == sum(8 * s + 4 for s in range(4)) == 4 * 4 * 4
    assert probes == cells  # all in bounds


def test_out_of_bounds_cells_are_skipped(oracle):
    # centre at the corner: negative coordinates wrap to huge usize -> None -> continue
    a = blank(5, 5)
    a[2, 0] = HIT
    assert oracle.distance_dot_array(HIT, a, (0, 0), 255, P) == (2, (0, 2))
    res, probes, cells = oracle.distance_dot_array(HIT, blank(3, 3), (0, 0), 255, P, count=True)
    # rings 0,1,2 have in-bounds cells, ring 3 has none -> break (helpers.rs:49-51)
    assert res is None and cells == 4 + 12 + 20 + 28


def test_centre_outside_array_gives_none_at_once(oracle):
    a = blank(4, 4)
    a[:] = HIT
    res, probes, cells = oracle.distance_dot_array(HIT, a, (10, 1), 255, P, count=True)
    assert res is None and probes == 0 and cells == 4  # ring 0 only, then break
    assert oracle.distance_dot_array(HIT, a, (4, 0), 255, P) is None  # x == width
    assert oracle.distance_dot_array(HIT, a, (3, 3), 255, P) == (0, (3, 3))


def test_max_steps_bounds_the_rings(oracle):
    a = blank(21, 21)
    a[10, 15] = HIT  # ring 5
    assert oracle.distance_dot_array(HIT, a, (10, 10), 5, P) is None  # rings 0..4 only
    assert oracle.distance_dot_array(HIT, a, (10, 10), 6, P) == (5, (15, 10))
    assert oracle.distance_dot_array(HIT, a, (10, 10), 0, P) is None


def test_predicate_is_strict_all_channels_absdiff(oracle):
    # disage PixelOpps::{substract,lt} as pinned in SURVEY.md 0.5
    a = blank(1, 1)
    what = [100, 100, 100]
    for cell, expect in [([109, 91, 100], True), ([110, 100, 100], False),
                         ([100, 100, 90], False), ([91, 91, 91], True),
                         ([100, 100, 111], False)]:
        a[0, 0] = cell
        got = oracle.distance_dot_array(what, a, (0, 0), 1, [10, 10, 10])
        assert (got is not None) == expect, cell
    a[0, 0] = [100, 130, 100]
    assert oracle.distance_dot_array(what, a, (0, 0), 1, [1, 31, 1]) is not None
    assert oracle.distance_dot_array(what, a, (0, 0), 1, [1, 30, 1]) is None
    assert oracle.distance_dot_array(what, a, (0, 0), 1, [0, 255, 255]) is None  # nothing < 0


def test_middle_is_half_sum_not_centre(oracle):
    # depth_image.rs:114-117: middle = ((x+w)/2, (y+h)/2) -- u32 floor division
    g = np.zeros(3, oracle.GROUP_DTYPE)
    g["x"], g["y"], g["w"], g["h"] = [10, 11, 0], [20, 21, 0], [1, 1, 7], [1, 1, 4]
    g["value"] = HIT
    a = blank(40, 40)
    a[:] = HIT
    m = oracle.match_groups(g, a, 255, P, threads=oracle.SERIAL)
    assert m["pos"].tolist() == [[5, 10], [6, 11], [3, 2]]
    assert m["raw"].tolist() == [0, 0, 0]


def test_u8_wrap_and_not_found_zero(oracle):
    # depth_image.rs:121 `d as u8`: distance 300 -> 44; None -> 0 (:118-120)
    a = blank(2, 400)
    a[0, 300] = HIT
    g = np.zeros(2, oracle.GROUP_DTYPE)
    g["w"] = g["h"] = 0  # middle = (x/2, y/2)
    g["x"] = [0, 0]
    g["value"][0] = HIT
    g["value"][1] = [1, 2, 3]  # matches nothing (precision 1)
    m = oracle.match_groups(g, a, 400, [1, 1, 1], threads=oracle.SERIAL)
    assert m["raw32"].tolist() == [300, 0] and m["raw"].tolist() == [44, 0]
    assert m["found"].tolist() == [1, 0]


def test_normalise_floor_and_max_zero(oracle):
    rc, out, mx = oracle.normalise(np.array([0, 1, 2, 3, 100, 200], np.uint8))
    assert rc == oracle.OK and mx == 200
    assert out.tolist() == [0, 1 * 255 // 200, 2 * 255 // 200, 3 * 255 // 200, 127, 255]
    rc, out, mx = oracle.normalise(np.zeros(5, np.uint8))
    assert rc == oracle.ERR_MAX_ZERO and mx == 0 and not out.any()


def test_uniform_groups_and_paint(oracle):
    img = np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)
    g = oracle.uniform_groups(img, 3, 2)
    assert len(g) == 3 * 3
    assert g["x"].tolist() == [0, 3, 6] * 3 and g["w"].tolist() == [3, 3, 1] * 3
    assert g["h"].tolist() == [2] * 6 + [1] * 3
    assert g["value"][0].tolist() == (img[0:2, 0:3].reshape(-1, 3).sum(0) // 6).tolist()
    one = oracle.uniform_groups(img, 1, 1)
    assert (one["value"].reshape(5, 7, 3) == img).all()
    depth = oracle.paint(g, np.arange(9, dtype=np.uint8), 7, 5)
    assert depth[0].tolist() == [0, 0, 0, 1, 1, 1, 2] and depth[4].tolist() == [6, 6, 6, 7, 7, 7, 8]


def test_threading_modes_agree(oracle):
    from stepth_amd import synth
    l, r, _ = synth.stereo_pair(96, 64, 32, seed=3)
    a = oracle.depth_from_additional(l, r, [36] * 3, 64, threads=oracle.SERIAL)
    b = oracle.depth_from_additional(l, r, [36] * 3, 64, threads=oracle.REF8)
    c = oracle.depth_from_additional(l, r, [36] * 3, 64, threads=oracle.ALLCORES)
    assert (a["depth"] == b["depth"]).all() and (a["depth"] == c["depth"]).all()
    assert a["probes"] == b["probes"] == c["probes"] and a["max"] == c["max"] > 0


def test_sad_oracle_recovers_planted_disparity(oracle):
    from stepth_amd import synth
    l, r, disp = synth.stereo_pair(160, 48, 24, seed=7)
    raw = oracle.sad_rows(l, r, 24, 5, [0, 0, 0])  # threshold 0 -> pure argmin
    inner = (slice(4, -4), slice(40, -8))
    assert (np.abs(raw[inner].astype(int) - disp[inner]) <= 1).mean() > 0.7  # measured 0.78


def test_gaussian_same_size_known_answers(oracle):
    """gaussian(x, 0.5) taps at offsets -3..3, normalised: [1.5e-8, 2.7e-4, 0.1065, 0.7866, ...]"""
    flat = np.full((5, 7), 200, np.uint8)
    assert (oracle.gaussian_same_size(flat) == 200).all()          # weights sum to 1, also
    imp = np.zeros((9, 9), np.uint8)                                # at the clipped borders
    imp[4, 4] = 255
    out = oracle.gaussian_same_size(imp)
    # vertical pass: round(255*0.7866)=201, round(255*0.1065)=27; then horizontal on those
    assert out[4, 4] == round(201 * 0.78656) and out[4, 3] == out[4, 5] == round(201 * 0.10645)
    assert out[3, 4] == out[5, 4] == round(27 * 0.78656) and out[3, 3] == round(27 * 0.10645)
    assert out[0].sum() == 0 and out[:, 0].sum() == 0
    one = np.array([[77]], np.uint8)
    assert oracle.gaussian_same_size(one)[0, 0] == 77
    row = np.arange(0, 250, 10, dtype=np.uint8)[None, :]
    out = oracle.gaussian_same_size(row)
    assert out.shape == row.shape and abs(int(out[0, 12]) - 120) <= 1


def test_adaptive_groups_rule_by_hand(oracle):
    """Stand-in for disage::open::rgb_discrete (own design, parity unpinned): a 4x2 image,
    left half flat, right half two different columns."""
    img = np.zeros((2, 4, 3), np.uint8)
    img[:, :2] = 10
    img[:, 2] = 100
    img[:, 3] = 200
    # no forced splits, precision 5: root 4x2 not uniform -> cut the longer side (width) into
    # 2x2 | 2x2; left uniform -> one group; right -> 1x2 | 1x2, each uniform
    g = oracle.adaptive_groups(img, [5, 5, 5], 0, 0)
    assert [(r["x"], r["y"], r["w"], r["h"], tuple(r["value"])) for r in g] == [
        (0, 0, 2, 2, (10, 10, 10)), (2, 0, 1, 2, (100, 100, 100)), (3, 0, 1, 2, (200, 200, 200))]
    # forced depth 2: the flat half is cut anyway (2x2 -> 1x2 | 1x2; width >= height cuts width)
    g = oracle.adaptive_groups(img, [5, 5, 5], 2, 0)
    assert sorted((r["x"], r["y"], r["w"], r["h"]) for r in g) == [(0, 0, 1, 2), (1, 0, 1, 2),
                                                                   (2, 0, 1, 2), (3, 0, 1, 2)]
    # max_splits caps the depth even when not uniform: depth 1 -> two 2x2 groups, value = floor mean
    g = oracle.adaptive_groups(img, [5, 5, 5], 0, 1)
    assert [(r["x"], r["w"], tuple(r["value"])) for r in g] == [(0, 2, (10, 10, 10)), (2, 2, (150, 150, 150))]
    # precision 0 is never uniform: single pixels (max_splits = ceil(log2(8)) = 3)
    assert len(oracle.adaptive_groups(img, [0, 0, 0], 0, 0)) == 8
    # strict '<' like the matcher: a range equal to the precision is not uniform
    two = np.array([[[10, 10, 10], [15, 10, 10]]], np.uint8)
    assert len(oracle.adaptive_groups(two, [5, 5, 5], 0, 0)) == 2
    assert len(oracle.adaptive_groups(two, [6, 5, 5], 0, 0)) == 1
