"""GPU parity of the reference-exact mode: the HIP path, called through the C-ABI
(libstepth_amd.so), against the CPU oracle on the same inputs.  Bit-exact everywhere (u8
depth, u32 distances, hit positions); integer work has no tolerance."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P36 = [36, 36, 36]


def scan_cells(cx, cy, s):
    out = []
    for i in (cy + s, cy - s):
        out += [(j, i) for j in range(cx - s, cx + s + 1)]
    for i in (cx + s, cx - s):
        out += [(i, j) for j in range(cy - s, cy + s + 1)]
    return out


def oracle_batch(oracle, array, what, from_xy, max_steps, prec):
    if int(from_xy.max(initial=0)) >= 2 ** 31:  # 2*from would wrap: query one by one
        out = {"found": np.zeros(len(what), np.uint8), "raw32": np.zeros(len(what), np.uint32),
               "pos": np.zeros((len(what), 2), np.uint32)}
        for i in range(len(what)):
            res = oracle.distance_dot_array(what[i], array, tuple(int(v) for v in from_xy[i]),
                                            max_steps, prec)
            if res is not None:
                out["found"][i], out["raw32"][i], out["pos"][i] = 1, res[0], res[1]
        return out
    g = np.zeros(len(what), oracle.GROUP_DTYPE)
    # middle = ((x+w)/2, (y+h)/2): x = 2*fx, w = 0 reproduces an arbitrary `from`
    g["x"], g["y"] = 2 * from_xy[:, 0].astype(np.uint64), 2 * from_xy[:, 1].astype(np.uint64)
    g["value"] = what
    return oracle.match_groups(g, array, max_steps, prec)


def check_batch(engine, oracle, array, what, from_xy, max_steps, prec):
    what = np.asarray(what, np.uint8).reshape(-1, 3)
    from_xy = np.asarray(from_xy, np.uint32).reshape(-1, 2)
    found, dist, pos = engine.distance_dot_array_batch(array, what, from_xy, max_steps, prec)
    ref = oracle_batch(oracle, array, what, from_xy, max_steps, prec)
    np.testing.assert_array_equal(found, ref["found"])
    np.testing.assert_array_equal(dist, ref["raw32"])
    np.testing.assert_array_equal(pos, ref["pos"])
    return found, dist, pos


# ---- helpers::distance_dot_array in isolation (src/helpers.rs:9-54) ----------------------

@pytest.mark.parametrize("s", [1, 2, 7, 8, 9, 31, 32, 33, 40])
def test_single_match_at_every_scan_position(engine, oracle, s):
    """one matching cell, placed at every position of ring s (covers the offset-table rings
    1..7, the table/side-loop seam at 8 and the 64-/128-cell chunk seams at 32/33)"""
    n = 2 * s + 5
    cx = cy = n // 2
    cells = sorted(set(scan_cells(cx, cy, s)))
    if len(cells) > 80:
        cells = cells[:: len(cells) // 80 + 1] + cells[-3:]
    hit = [200, 150, 100]
    for (x, y) in cells:
        a = np.zeros((n, n, 3), np.uint8)
        a[y, x] = hit
        f, d, p = check_batch(engine, oracle, a, [hit], [[cx, cy]], 255, [10, 10, 10])
        assert f[0] == 1 and tuple(p[0]) == (x, y)


@pytest.mark.parametrize("s", [1, 3, 8, 20, 70])
def test_two_matches_first_in_scan_order_wins(engine, oracle, s):
    rng = np.random.default_rng(s)
    n = 2 * s + 3
    cx = cy = n // 2
    order = scan_cells(cx, cy, s)
    first = {}
    for i, c in enumerate(order):  # corners are visited twice: the first visit counts
        first.setdefault(c, i)
    cells = sorted(first, key=first.get)
    for _ in range(24):
        ia, ib = sorted(rng.choice(len(cells), 2, replace=False))
        a = np.zeros((n, n, 3), np.uint8)
        a[cells[ia][1], cells[ia][0]] = 255
        a[cells[ib][1], cells[ib][0]] = 255
        f, d, p = check_batch(engine, oracle, a, [[255] * 3], [[cx, cy]], 255, [5, 5, 5])
        assert tuple(p[0]) == cells[ia]


def test_random_queries_on_noise_and_smooth_arrays(engine, oracle):
    from stepth_amd import synth
    rng = np.random.default_rng(11)
    smooth, _, _ = synth.stereo_pair(301, 203, 32, seed=5)
    noise = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
    for arr, prec in [(smooth, P36), (smooth, [8, 8, 8]), (noise, [40, 40, 40]),
                      (noise, [20, 60, 30])]:
        h, w = arr.shape[:2]
        n = 1500
        what = rng.integers(0, 256, (n, 3), dtype=np.uint8)
        # half the queries take their colour from the array (so that hits exist)
        ys, xs = rng.integers(0, h, n), rng.integers(0, w, n)
        what[::2] = arr[ys[::2], xs[::2]]
        from_xy = np.stack([rng.integers(0, w + 20, n), rng.integers(0, h + 20, n)], 1)
        for ms in (255, 40, 3, 1, 0):
            check_batch(engine, oracle, arr, what, from_xy, ms, prec)


def test_centre_outside_and_degenerate_arrays(engine, oracle):
    a = np.full((4, 6, 3), 77, np.uint8)
    q = [[6, 0], [0, 4], [5, 3], [100000, 2], [2, 4000000000], [0, 0]]
    f, d, p = check_batch(engine, oracle, a, [[77] * 3] * len(q), q, 255, [1, 1, 1])
    assert f.tolist() == [0, 0, 1, 0, 0, 1]
    one = np.array([[[9, 9, 9]]], np.uint8)
    check_batch(engine, oracle, one, [[9, 9, 9], [9, 9, 10]], [[0, 0], [0, 0]], 255, [1, 1, 1])
    row = np.zeros((1, 700, 3), np.uint8)
    row[0, 650] = 50
    f, d, p = check_batch(engine, oracle, row, [[50] * 3] * 2, [[10, 0], [400, 0]], 255, [1] * 3)
    assert f.tolist() == [0, 1] and d[1] == 250
    col = np.zeros((700, 1, 3), np.uint8)
    col[3, 0] = 50
    check_batch(engine, oracle, col, [[50] * 3] * 2, [[0, 200], [0, 300]], 255, [1] * 3)


def test_precision_edges(engine, oracle):
    rng = np.random.default_rng(3)
    arr = rng.integers(0, 256, (40, 40, 3), dtype=np.uint8)
    what = rng.integers(0, 256, (400, 3), dtype=np.uint8)
    what[:50] = [0, 0, 0]
    what[50:100] = [255, 255, 255]
    fxy = rng.integers(0, 40, (400, 2))
    for prec in ([0, 0, 0], [1, 1, 1], [255, 255, 255], [0, 255, 255], [255, 1, 128],
                 [2, 200, 17]):
        check_batch(engine, oracle, arr, what, fxy, 255, prec)


def test_far_hits_and_distance_extremes(engine, oracle):
    """hits at ring 254 (corner: distance 359 -> wraps to 103 as u8) and just beyond max"""
    a = np.zeros((600, 600, 3), np.uint8)
    a[300 + 254, 300 + 254] = 255
    a[300 - 254, 300] = 254
    f, d, p = check_batch(engine, oracle, a, [[255] * 3, [254] * 3, [253] * 3],
                          [[300, 300]] * 3, 255, [1, 1, 1])
    assert f.tolist() == [1, 1, 0] and d.tolist() == [359, 254, 0]
    check_batch(engine, oracle, a, [[255] * 3], [[300, 300]], 254, [1, 1, 1])  # ring 254 cut off


def test_max_steps_beyond_the_reference_255(engine, oracle):
    """`max` is a parameter of helpers::distance_dot_array (helpers.rs:13); beyond 258 the
    block spiral leaves its LDS-staged part and is read from global memory"""
    rng = np.random.default_rng(5)
    a = np.zeros((520, 640, 3), np.uint8)
    a[500, 600] = 200   # ring 590 from (10, 10)
    a[30, 450] = 100    # ring 440
    a[400, 20] = 50     # ring 390
    q = [[10, 10]] * 3 + [[320, 260]] * 3
    what = [[200] * 3, [100] * 3, [50] * 3] * 2
    for ms in (700, 600, 441, 440, 391, 300):
        f, d, p = check_batch(engine, oracle, a, what, q, ms, [1, 1, 1])
    f, d, p = check_batch(engine, oracle, a, what, q, 700, [1, 1, 1])
    assert f.tolist() == [1] * 6 and d[0] == int(np.hypot(590, 490))
    noise = rng.integers(0, 256, (300, 300, 3), dtype=np.uint8)
    what = rng.integers(0, 256, (300, 3), dtype=np.uint8)
    check_batch(engine, oracle, noise, what, rng.integers(0, 300, (300, 2)), 1000, [3, 3, 3])


# ---- the whole path (src/depth_image.rs:96-132, no Gaussian tail) ------------------------

def run_both(engine, oracle, main, add, prec, **kw):
    ref = oracle.depth_from_additional(main, add, prec, kw.get("max_steps", 255),
                                       kw.get("tile", (1, 1)), kw.get("groups"))
    from stepth_amd import MaxZeroError
    try:
        depth, mx = engine.depth_from_additional(main, add, prec, **kw)
        status = 0
    except MaxZeroError:
        status, depth, mx = -4, None, 0
    assert status == ref["status"]
    if status == 0:
        assert mx == ref["max"]
        np.testing.assert_array_equal(depth, ref["depth"])
        raw, mx2 = engine.depth_from_additional(main, add, prec, raw_only=True, **kw)
        np.testing.assert_array_equal(raw, ref["raw"])
        assert mx2 == ref["max"]
    return ref


def test_asset_pair_bit_exact(engine, oracle, asset):
    """config 1 of BASELINE.json: the reference's own 600x400 pair, precision [36;3]"""
    ref = run_both(engine, oracle, asset["main"], asset["additional"], P36)
    assert ref["max"] == 205


@pytest.mark.parametrize("w,h,ms", [(64, 64, 255), (257, 131, 255), (640, 360, 128),
                                     (333, 500, 64), (1, 1, 255), (3, 2, 255), (1, 300, 255),
                                     (300, 1, 255)])
def test_synthetic_pairs_1x1(engine, oracle, w, h, ms):
    from stepth_amd import synth
    l, r, _ = synth.stereo_pair(w, h, 32, seed=w * 7 + h)
    for prec in (P36, [12, 12, 12]):
        run_both(engine, oracle, l, r, prec, max_steps=ms)


def test_rgba_main_and_strided_inputs(engine, oracle):
    from stepth_amd import synth
    l, r, _ = synth.stereo_pair(200, 120, 32, seed=9)
    ref = oracle.depth_from_additional(l, r, P36, 255)
    rgba = np.concatenate([l, np.full((120, 200, 1), 9, np.uint8)], axis=2)  # alpha ignored
    d, mx = engine.depth_from_additional(rgba, r, P36)
    np.testing.assert_array_equal(d, ref["depth"])
    big = np.zeros((120, 260, 3), np.uint8)
    big[:, :200] = r
    lbig = np.zeros((120, 231, 3), np.uint8)
    lbig[:, :200] = l
    out = np.full((120, 300), 0xEE, np.uint8)
    engine.depth_from_additional(lbig[:, :200], big[:, :200], P36, out=out[:, :200])
    np.testing.assert_array_equal(out[:, :200], ref["depth"])
    assert (out[:, 200:] == 0xEE).all()


def test_different_sized_additional(engine, oracle):
    from stepth_amd import synth
    l, _, _ = synth.stereo_pair(160, 100, 16, seed=2)
    _, r, _ = synth.stereo_pair(70, 45, 16, seed=2)   # smaller than main/2 in places
    run_both(engine, oracle, l, r, P36)
    _, r2, _ = synth.stereo_pair(400, 300, 16, seed=3)
    run_both(engine, oracle, l, r2, P36)


@pytest.mark.parametrize("tile", [(5, 5), (2, 3), (16, 16), (7, 1), (64, 64)])
def test_uniform_tiles(engine, oracle, tile):
    from stepth_amd import synth
    l, r, _ = synth.stereo_pair(203, 117, 32, seed=tile[0])
    run_both(engine, oracle, l, r, P36, tile=tile)


def test_explicit_group_list(engine, oracle):
    """a caller-supplied partition (the stand-in for disage::open::rgb_discrete's output)"""
    from stepth_amd import synth
    l, r, _ = synth.stereo_pair(128, 96, 32, seed=4)
    rng = np.random.default_rng(0)
    groups = []
    def split(x, y, w, h, depth):
        if depth == 0 or w < 2 or h < 2 or rng.random() < 0.15:
            v = l[y:y + h, x:x + w].reshape(-1, 3).astype(np.uint64).sum(0) // (w * h)
            groups.append((x, y, w, h, v))
            return
        if w >= h:
            c = int(rng.integers(1, w))
            split(x, y, c, h, depth - 1), split(x + c, y, w - c, h, depth - 1)
        else:
            c = int(rng.integers(1, h))
            split(x, y, w, c, depth - 1), split(x, y + c, w, h - c, depth - 1)
    split(0, 0, 128, 96, 9)
    g = np.zeros(len(groups), oracle.GROUP_DTYPE)
    for i, (x, y, w, h, v) in enumerate(groups):
        g[i]["x"], g[i]["y"], g[i]["w"], g[i]["h"], g[i]["value"] = x, y, w, h, v
    g = g[rng.permutation(len(g))]  # order in the Vec must not matter for results
    run_both(engine, oracle, l, r, P36, groups=g)


def test_max_zero_is_a_defined_outcome(engine, oracle):
    from stepth_amd import MaxZeroError
    flat = np.full((20, 30, 3), 100, np.uint8)
    ref = oracle.depth_from_additional(flat, flat, P36)
    assert ref["status"] == oracle.ERR_MAX_ZERO
    out = np.full((20, 30), 5, np.uint8)
    with pytest.raises(MaxZeroError):
        engine.depth_from_additional(flat, flat, P36, out=out)
    assert not out.any()
    # nothing matches anywhere -> all None -> all 0 -> same outcome
    a, b = np.zeros((8, 8, 3), np.uint8), np.full((8, 8, 3), 255, np.uint8)
    with pytest.raises(MaxZeroError):
        engine.depth_from_additional(a, b, [1, 1, 1])


def test_size_mismatch_and_bad_arguments(engine):
    from stepth_amd import StepthError
    a = np.zeros((8, 8, 3), np.uint8)
    with pytest.raises(StepthError) as ei:
        engine.depth_from_additional(a, a, P36, out=np.zeros((8, 9), np.uint8))
    assert ei.value.status == -2 and "Sizes don't match" in str(ei.value)
    with pytest.raises(StepthError) as ei:
        engine.depth_from_additional(a, a, P36, tile=(0, 1))
    assert ei.value.status == -1
    with pytest.raises(StepthError) as ei:
        engine.depth_from_additional(a, a, P36, rows=(2, 20))
    assert ei.value.status == -1


def test_row_strips_compose_to_the_full_frame(engine, oracle):
    """multi-GPU building block: PASS 1 per strip (RAW_ONLY), max over strips, PASS 2 per strip"""
    from stepth_amd import synth
    from stepth_amd.sharding import strip_bounds
    l, r, _ = synth.stereo_pair(190, 143, 32, seed=12)
    for tile in ((1, 1), (5, 5)):
        ref = oracle.depth_from_additional(l, r, P36, 255, tile)
        for world in (2, 3, 8):
            out = np.zeros((143, 190), np.uint8)
            maxes = []
            for (r0, r1) in strip_bounds(143, world, tile[1]):
                _, mx = engine.depth_from_additional(l, r, P36, tile=tile, rows=(r0, r1),
                                                     raw_only=True, out=out)
                maxes.append(mx)
            np.testing.assert_array_equal(out, ref["raw"])
            assert max(maxes) == ref["max"]
            for (r0, r1) in strip_bounds(143, world, tile[1]):
                if r1 > r0:
                    engine.normalise_rows(out[r0:r1], max(maxes))
            np.testing.assert_array_equal(out, ref["depth"])


def test_strips_with_short_reach_pack_only_part_of_the_additional_image(engine, oracle, asset):
    """a strip only packs the rows its searches can reach (centre rows +- (max_steps-1)); stale
    rows from earlier calls with other images must never leak into the result"""
    from stepth_amd import synth
    from stepth_amd.sharding import strip_bounds
    main, add = asset["main"], asset["additional"]
    other, _, _ = synth.stereo_pair(600, 400, 32, seed=99)
    for ms, tile in ((16, (1, 1)), (40, (1, 1)), (9, (4, 4)), (2, (1, 1))):
        ref = oracle.depth_from_additional(main, add, P36, ms, tile)
        engine.depth_from_additional(main, other, P36, max_steps=255)  # poison T/BB/SB
        out = np.zeros((400, 600), np.uint8)
        maxes = []
        for (r0, r1) in strip_bounds(400, 8, tile[1]):
            _, mx = engine.depth_from_additional(main, add, P36, max_steps=ms, tile=tile,
                                                 rows=(r0, r1), raw_only=True, out=out)
            maxes.append(mx)
        np.testing.assert_array_equal(out, ref["raw"])
        assert max(maxes) == ref["max"]


def test_reaches_beyond_the_dense_search(engine, oracle):
    """max_steps > 263 (the reference itself always searches 255 rings, src/depth_image.rs:119):
    one group per pixel then runs as 1x1 tiles of the prepared-query search, whose per-lane
    ring walk takes any reach -- whole path, batch (+ Gaussian tail), row strips, against the
    oracle; distances beyond 255 wrap like `d as u8`"""
    import torch
    from stepth_amd import synth
    l, r, _ = synth.stereo_pair(720, 520, 200, seed=11)
    tl, tr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
    for prec, ms in (([36] * 3, 420), ([12] * 3, 700), ([12] * 3, 264)):
        ref = oracle.depth_from_additional(l, r, prec, ms)
        d, mx = engine.depth_from_additional(l, r, prec, max_steps=ms)
        assert mx == ref["max"] and np.array_equal(d, ref["depth"]), ms
        outs, mxs = engine.depth_from_additional_batch([tl, tl], [tr, tr], prec, max_steps=ms)
        assert list(mxs) == [ref["max"]] * 2 and np.array_equal(outs[1].cpu().numpy(), ref["depth"]), ms
        out = np.zeros((520, 720), np.uint8)
        for r0, r1 in ((0, 131), (131, 400), (400, 520)):
            engine.depth_from_additional(l, r, prec, max_steps=ms, rows=(r0, r1), raw_only=True, out=out)
        assert np.array_equal(out, ref["raw"]), ms
    assert (ref["raw32"] > 255).any() if "raw32" in ref else True
    outs, _ = engine.depth_from_additional_batch([tl], [tr], [12] * 3, max_steps=264, gaussian_tail=True)
    assert np.array_equal(outs[0].cpu().numpy(), oracle.gaussian_same_size(ref["depth"]))


def test_short_reach_packs_only_the_reachable_columns(engine, oracle, asset):
    """the centres lie in the left half of the additional image (middle = ((x+w)/2, (y+h)/2),
    src/depth_image.rs:114-117): block columns beyond centre + reach are not packed and keep an
    earlier image's cells and bounds, which must never show -- single call, batch, tiles, and
    the adaptive flow (the engine's own groups; a caller's list gets everything packed)"""
    import torch
    from stepth_amd import synth
    main, add = asset["main"][..., :3], asset["additional"][..., :3]
    other = synth.stereo_pair(600, 400, 32, seed=98)[0]
    m = torch.from_numpy(np.ascontiguousarray(main)).cuda()
    a = torch.from_numpy(np.ascontiguousarray(add)).cuda()
    o = torch.from_numpy(np.ascontiguousarray(other)).cuda()
    for ms in (3, 16, 40, 90):
        engine.depth_from_additional_batch([m, m], [o, o], P36, max_steps=255)  # poison everything
        ref = oracle.depth_from_additional(main, add, P36, ms)
        d, mx = engine.depth_from_additional(main, add, P36, max_steps=ms)
        assert mx == ref["max"] and np.array_equal(d, ref["depth"]), ms
        engine.depth_from_additional_batch([m, m], [o, o], P36, max_steps=255)
        outs, mxs = engine.depth_from_additional_batch([m, m], [a, a], P36, max_steps=ms)
        assert list(mxs) == [ref["max"]] * 2 and np.array_equal(outs[1].cpu().numpy(), ref["depth"]), ms
        engine.depth_from_additional_batch([m, m], [o, o], P36, max_steps=255)
        ref = oracle.depth_from_additional(main, add, P36, ms, (5, 3))
        outs, mxs = engine.depth_from_additional_batch([m, m], [a, a], P36, max_steps=ms, tile=(5, 3))
        assert list(mxs) == [ref["max"]] * 2 and np.array_equal(outs[0].cpu().numpy(), ref["depth"]), ms
        engine.depth_from_additional_batch([m, m], [o, o], P36, max_steps=255)
        groups = oracle.adaptive_groups(main, P36, 16, 0)
        ref = oracle.depth_from_additional(main, add, P36, ms, groups=groups)
        outs, mxs = engine.adaptive_depth_batch([m, m], [a, a], P36, max_steps=ms)
        assert list(mxs) == [ref["max"]] * 2 and np.array_equal(outs[1].cpu().numpy(), ref["depth"]), ms
        # a caller's list with centres in the right half (rectangles reaching past the image):
        # everything is packed for it
        g = np.zeros(3, oracle.GROUP_DTYPE)
        g["x"], g["y"], g["w"], g["h"] = [500, 590, 10], [10, 300, 390], [600, 400, 900], [20, 90, 10]
        g["value"] = [main[10, 500], main[300, 590], main[390, 10]]
        engine.depth_from_additional_batch([m, m], [o, o], P36, max_steps=255)
        want = oracle.match_groups(g, add, ms, P36)["raw"]
        gd = torch.from_numpy(g.view(np.uint8).reshape(-1, 20).copy()).cuda()
        outs, _ = engine.depth_from_groups_batch([m], [a], P36, [gd], [3], max_steps=ms, raw_only=True)
        got = outs[0].cpu().numpy()
        assert [int(got[10, 500]), int(got[300, 590]), int(got[390, 10])] == [int(v) for v in want], ms


@pytest.mark.parametrize("tile,block", [((1, 1), 2), ((1, 1), 16), ((1, 1), 6), ((4, 4), 8), ((5, 2), 10)])
def test_interleaved_row_blocks_compose_to_the_full_frame(engine, oracle, tile, block):
    """rank r of N owns row blocks r, r+N, ...: PASS 1 per rank, max over ranks, PASS 2 per
    rank -- every pixel exactly once, result identical to the single call"""
    from stepth_amd import synth
    l, r, _ = synth.stereo_pair(173, 141, 32, seed=31)
    ref = oracle.depth_from_additional(l, r, P36, 255, tile)
    for world in (2, 3, 8):
        out = np.full((141, 173), 0xAB, np.uint8)
        maxes = []
        for rank in range(world):
            before = out.copy()
            _, mx = engine.depth_from_additional(l, r, P36, tile=tile, raw_only=True, out=out,
                                                 row_pattern=(block, world, rank))
            maxes.append(mx)
            ys = np.arange(141)
            other = (ys // block) % world != rank
            np.testing.assert_array_equal(out[other], before[other])  # nothing else touched
        np.testing.assert_array_equal(out, ref["raw"])
        assert max(maxes) == ref["max"]
        for rank in range(world):
            engine.normalise_rows(out, max(maxes), row_pattern=(block, world, rank))
        np.testing.assert_array_equal(out, ref["depth"])
    from stepth_amd import StepthError
    with pytest.raises(StepthError):
        engine.depth_from_additional(l, r, P36, row_pattern=(3, 2, 0))   # odd block
    with pytest.raises(StepthError):
        engine.depth_from_additional(l, r, P36, row_pattern=(4, 2, 2))   # rem >= mod


def test_batch_of_frames_equals_frame_by_frame(engine, oracle):
    """stepth_depth_from_additional_batch: many frames in one set of launches, each frame
    normalised by its own max; also > 32 frames (chunked), strips, row patterns, async"""
    import torch
    from stepth_amd import synth, MaxZeroError
    frames = []
    for i in range(5):
        l, r, _ = synth.stereo_pair(150, 90, 32, seed=40 + i)
        if i == 3:
            r = np.ascontiguousarray(r[::-1])  # a frame with a different max
        frames.append((l, r))
    refs = [oracle.depth_from_additional(l, r, P36, 64) for l, r in frames]
    assert len({ref["max"] for ref in refs}) > 1
    dm = [torch.from_numpy(l).cuda() for l, _ in frames]
    da = [torch.from_numpy(r).cuda() for _, r in frames]
    outs, mx = engine.depth_from_additional_batch(dm, da, P36, max_steps=64)
    for o, m, ref in zip(outs, mx, refs):
        np.testing.assert_array_equal(o.cpu().numpy(), ref["depth"])
        assert m == ref["max"]
    # 40 frames -> two chunks; raw_only; async with device maxima
    many_m, many_a = [dm[i % 5] for i in range(40)], [da[i % 5] for i in range(40)]
    mxd = torch.zeros(40, dtype=torch.int32, device="cuda")
    engine.use_torch_stream()
    try:
        outs, none = engine.depth_from_additional_batch(many_m, many_a, P36, max_steps=64,
                                                        raw_only=True, max_out=mxd,
                                                        asynchronous=True)
        assert none is None
        for i, o in enumerate(outs):
            np.testing.assert_array_equal(o.cpu().numpy(), refs[i % 5]["raw"])
        assert mxd.tolist() == [refs[i % 5]["max"] for i in range(40)]
    finally:
        engine.reset_stream()
    # strips and interleaved patterns compose, per frame
    outs = [torch.full((90, 150), 7, dtype=torch.uint8, device="cuda") for _ in range(5)]
    maxes = np.zeros((3, 5), int)
    for rank in range(3):
        _, mx = engine.depth_from_additional_batch(dm, da, P36, max_steps=64, raw_only=True,
                                                   outs=outs, row_pattern=(4, 3, rank))
        maxes[rank] = mx
    for i, o in enumerate(outs):
        np.testing.assert_array_equal(o.cpu().numpy(), refs[i]["raw"])
    assert maxes.max(axis=0).tolist() == [ref["max"] for ref in refs]
    outs = [torch.zeros((90, 150), dtype=torch.uint8, device="cuda") for _ in range(5)]
    for rows in ((0, 31), (31, 90)):
        engine.depth_from_additional_batch(dm, da, P36, max_steps=64, raw_only=True, outs=outs,
                                           rows=rows)
    for i, o in enumerate(outs):
        np.testing.assert_array_equal(o.cpu().numpy(), refs[i]["raw"])
    # a flat frame in the batch: defined outcome
    flat = torch.full((90, 150, 3), 50, dtype=torch.uint8, device="cuda")
    with pytest.raises(MaxZeroError):
        engine.depth_from_additional_batch([dm[0], flat], [da[0], flat], P36, max_steps=64)


def test_device_memory_async_path(engine, oracle):
    import torch
    from stepth_amd import synth
    l, r, _ = synth.stereo_pair(320, 200, 32, seed=21)
    ref = oracle.depth_from_additional(l, r, P36, 128)
    dl, dr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
    engine.use_torch_stream()
    try:
        mx = torch.zeros(1, dtype=torch.int32, device="cuda")
        out, _ = engine.depth_from_additional(dl, dr, P36, max_steps=128, max_out=mx,
                                              asynchronous=True)
        # no synchronize: torch ops queued behind the engine on the same stream must see it
        np.testing.assert_array_equal(out.clone().cpu().numpy(), ref["depth"])
        assert int(mx.clone().item()) == ref["max"]
        # strip worker flow on device memory: RAW_ONLY -> normalise_rows with a device max
        from stepth_amd.sharding import EngineStripWorker, sharded_depth
        wk = EngineStripWorker(engine, P36, max_steps=128)
        d3, m3 = sharded_depth(wk, [(dl, dr), (dl.flip(0).contiguous(), dr.flip(0).contiguous())],
                               200, 320)
        np.testing.assert_array_equal(d3[0].cpu().numpy(), ref["depth"])
        ref_f = oracle.depth_from_additional(l[::-1].copy(), r[::-1].copy(), P36, 128)
        np.testing.assert_array_equal(d3[1].cpu().numpy(), ref_f["depth"])
        assert m3.tolist() == [ref["max"], ref_f["max"]]
        out2, mx2 = engine.depth_from_additional(dl, dr, P36, max_steps=128)
        np.testing.assert_array_equal(out2.cpu().numpy(), ref["depth"])
        assert mx2 == ref["max"]
    finally:
        engine.reset_stream()


def test_stats_counters(engine, oracle, asset):
    engine.stats_enable(True)
    try:
        engine.depth_from_additional(asset["main"], asset["additional"], P36)
        s = engine.stats()
    finally:
        engine.stats_enable(False)
    assert s["groups"] == 240000 and s["ring0_hits"] == 175117  # SURVEY.md 8c histogram
    assert s["probes"] > 240000


def test_raw_stats_words_of_the_dense_search(engine, asset):
    """stepth_get_stats_raw: words 0..5 are the documented device counters, 6 and 7 the filter
    diagnostics bench.py prints (superblock visits without a candidate block, block probes
    that did not improve a result)"""
    engine.stats_enable(True)
    try:
        engine.depth_from_additional(asset["main"], asset["additional"], P36, max_steps=255)
        s, raw = engine.stats(), engine.stats_raw(8)
    finally:
        engine.stats_enable(False)
    assert raw[0] == s["ring0_hits"] and raw[1] == s["probes"] and raw[5] == s["blocks_probed"]
    assert raw[4] == s["superblock_steps"] > 0
    assert 0 <= raw[6] <= s["superblock_steps"]
    assert 0 < raw[7] < s["blocks_probed"]


def test_depth_image_mirror_end_to_end(engine, oracle, asset, tmp_path):
    """the README flow (Readme.md:11-16) through the host mirror"""
    from PIL import Image
    from stepth_amd import DepthImage
    pm, pa = tmp_path / "main.png", tmp_path / "additional.png"
    Image.fromarray(asset["main"]).save(pm)
    Image.fromarray(asset["additional"]).save(pa)
    img = DepthImage.open(str(pm))
    img.open_depth_from_additional(str(pa), [255 // 7] * 3, engine=engine)
    ref = oracle.depth_from_additional(asset["main"], asset["additional"], P36)
    # the mirror applies the Gaussian tail like the reference (:133-134; parity unpinned)
    np.testing.assert_array_equal(img.depth, oracle.gaussian_same_size(ref["depth"]))
    d = img.depth.astype(float)
    assert np.corrcoef(d.ravel(), asset["depth"].astype(float).ravel())[0, 1] >= 0.985
    img.open_depth_from_additional(str(pa), [36] * 3, engine=engine, gaussian_tail=False)
    np.testing.assert_array_equal(img.depth, ref["depth"])


@pytest.mark.parametrize("w,h", [(600, 400), (1, 1), (3, 2), (2, 9), (64, 7), (257, 131), (4, 1), (4, 5),
                                 (8, 3), (12, 40), (1920, 16)])
def test_gaussian_tail_matches_oracle(engine, oracle, w, h):
    rng = np.random.default_rng(w * 31 + h)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    img[rng.random((h, w)) < 0.5] = 0  # depth maps are mostly flat with sharp steps
    np.testing.assert_array_equal(engine.gaussian_same_size(img), oracle.gaussian_same_size(img))
    strided = np.zeros((h, w + 5), np.uint8)
    strided[:, :w] = img
    out = np.full((h, w + 3), 9, np.uint8)
    engine.gaussian_same_size(strided[:, :w], out=out[:, :w])
    np.testing.assert_array_equal(out[:, :w], oracle.gaussian_same_size(img))
    assert (out[:, w:] == 9).all()
    # device planes: aligned ones take the 4-pixels-per-thread kernels, offset ones the scalar one
    import torch
    t = torch.from_numpy(img).cuda()
    np.testing.assert_array_equal(engine.gaussian_same_size(t).cpu().numpy(), oracle.gaussian_same_size(img))
    wide = torch.zeros((h, w + 8), dtype=torch.uint8, device="cuda")
    wide[:, 1:1 + w] = t
    np.testing.assert_array_equal(engine.gaussian_same_size(wide[:, 1:1 + w]).cpu().numpy(),
                                  oracle.gaussian_same_size(img))


def test_gaussian_tail_flag_and_device_path(engine, oracle):
    import torch
    from stepth_amd import StepthError, synth
    l, r, _ = synth.stereo_pair(210, 130, 32, seed=6)
    ref = oracle.gaussian_same_size(oracle.depth_from_additional(l, r, P36)["depth"])
    d, _ = engine.depth_from_additional(l, r, P36, gaussian_tail=True)
    np.testing.assert_array_equal(d, ref)
    dl, dr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
    dd, _ = engine.depth_from_additional(dl, dr, P36, gaussian_tail=True)
    np.testing.assert_array_equal(dd.cpu().numpy(), ref)
    bare, _ = engine.depth_from_additional(dl, dr, P36)
    np.testing.assert_array_equal(engine.gaussian_same_size(bare).cpu().numpy(), ref)
    with pytest.raises(StepthError):  # strips cannot blur across their seams
        engine.depth_from_additional(l, r, P36, gaussian_tail=True, rows=(0, 64))


def _sampled_raw_equals_oracle(oracle, raw, main, add, prec, max_steps, n=20000, seed=3):
    """`raw` (GPU, one group per pixel) against the oracle's match_groups on n random pixels
    (src/depth_image.rs:112-121): the full-size frames take minutes on the CPU, a sample seconds"""
    h, w = main.shape[:2]
    rng = np.random.default_rng(seed)
    ys, xs = rng.integers(0, h, n), rng.integers(0, w, n)
    g = np.zeros(n, oracle.GROUP_DTYPE)
    g["x"], g["y"], g["w"], g["h"] = xs, ys, 1, 1
    g["value"] = main[ys, xs, :3]
    ref = oracle.match_groups(g, add, max_steps, prec)["raw"]
    bad = np.flatnonzero(ref != raw[ys, xs])
    assert len(bad) == 0, f"{len(bad)} of {n} sampled pixels differ, first (x={xs[bad[0]]}, y={ys[bad[0]]}): " \
                          f"{raw[ys[bad[0]], xs[bad[0]]]} vs oracle {ref[bad[0]]}"


def test_full_size_properties_1080p(engine, oracle):
    """BASELINE full size (1920x1080, d_max=128 -> max_steps) without the oracle: size-
    independent properties -- strips compose, idempotence, max/normalise consistency,
    ring-0 pixels are zero, every raw distance is reachable within max_steps."""
    import torch
    from stepth_amd import synth
    l, r, _ = synth.stereo_pair(1920, 1080, 128, seed=0x5EED)
    raw, mx = engine.depth_from_additional(l, r, P36, max_steps=128, raw_only=True)
    _sampled_raw_equals_oracle(oracle, raw, l, r, P36, 128)
    raw64, _ = engine.depth_from_additional(l, r, P36, max_steps=64, raw_only=True)  # config 2
    _sampled_raw_equals_oracle(oracle, raw64, l, r, P36, 64, seed=4)
    depth, mx2 = engine.depth_from_additional(l, r, P36, max_steps=128)
    depth_b, _ = engine.depth_from_additional(l, r, P36, max_steps=128)
    np.testing.assert_array_equal(depth, depth_b)  # deterministic
    assert mx == mx2 == int(raw.max()) and 0 < mx <= int(np.floor(np.hypot(127, 127)))
    np.testing.assert_array_equal(depth, (raw.astype(np.uint32) * 255 // mx).astype(np.uint8))
    # pixels whose own colour sits at their search centre must be 0
    ys, xs = np.mgrid[0:1080, 0:1920]
    centre = r[(ys + 1) // 2, (xs + 1) // 2].astype(int)
    ring0 = (np.abs(centre - l.astype(int)) < 36).all(axis=2)
    assert not raw[ring0].any()
    # a miss at ring 0 gives d >= 1, or None -> 0 when nothing matches within 128 rings
    assert (raw[~ring0] > 0).mean() > 0.5  # measured 0.81
    # two strips == full frame
    out = np.zeros_like(raw)
    for rows in ((0, 540), (540, 1080)):
        engine.depth_from_additional(l, r, P36, max_steps=128, rows=rows, raw_only=True, out=out)
    np.testing.assert_array_equal(out, raw)


def _big_frame(w, h):
    """a large synthetic pair assembled from mirrored 960x540 tiles (fast to generate)"""
    from stepth_amd import synth
    l, r, _ = synth.stereo_pair(960, 540, 128, seed=0x4B)
    def tile(a):
        row = np.concatenate([a, a[:, ::-1]], axis=1)
        quad = np.concatenate([row, row[::-1]], axis=0)
        reps = (-(-h // quad.shape[0]), -(-w // quad.shape[1]), 1)
        return np.ascontiguousarray(np.tile(quad, reps)[:h, :w])
    return tile(l), tile(r)


@pytest.mark.parametrize("w,h,ms", [(3840, 2160, 128), (7680, 4320, 255), (7680, 4320, 256)])
def test_full_size_properties_4k_8k(engine, oracle, w, h, ms):
    """BASELINE configs 3 and 5 (4K d_max=128; 8K at the reference's reach 255 and at d_max = max_steps = 256,
    BASELINE.json's literal figure): 20 000 sampled
    pixels against the oracle, plus size-independent properties of the whole frame --
    determinism, normalise consistency, ring-0 pixels zero, strips (as dealt to 8 ranks) compose
    to the full frame, the search never reports a distance beyond its reach"""
    from stepth_amd.sharding import strip_bounds
    l, r = _big_frame(w, h)
    raw, mx = engine.depth_from_additional(l, r, P36, max_steps=ms, raw_only=True)
    _sampled_raw_equals_oracle(oracle, raw, l, r, P36, ms)
    depth, mx2 = engine.depth_from_additional(l, r, P36, max_steps=ms)
    assert mx == mx2 == int(raw.max()) and mx > 0
    reach = int(np.floor(np.hypot(ms - 1, ms - 1)))
    assert mx <= min(reach, 255)
    np.testing.assert_array_equal(depth, (raw.astype(np.uint32) * 255 // mx).astype(np.uint8))
    ys, xs = np.mgrid[0:h, 0:w]
    centre = r[(ys + 1) // 2, (xs + 1) // 2].astype(np.int16)
    ring0 = (np.abs(centre - l.astype(np.int16)) < 36).all(axis=2)
    del ys, xs, centre
    assert not raw[ring0].any()
    out = np.zeros_like(raw)
    for rows in strip_bounds(h, 8):
        engine.depth_from_additional(l, r, P36, max_steps=ms, rows=rows, raw_only=True, out=out)
    np.testing.assert_array_equal(out, raw)


def test_images_beyond_the_cell_index_range_are_refused(engine):
    """kernels index cells with 32-bit byte offsets: more than 2^30 cells is a clean error"""
    import ctypes as C
    import torch
    from stepth_amd import _lib as L
    buf = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    big = L.Image(buf.data_ptr(), 40000, 40000, 3, L.MEM_DEVICE, 120000)
    small = L.Image(buf.data_ptr(), 16, 16, 3, L.MEM_DEVICE, 48)
    depth = L.Image(buf.data_ptr(), 16, 16, 1, L.MEM_DEVICE, 16)
    p = L.Params()
    p.struct_size = C.sizeof(L.Params)
    p.precision = (C.c_uint8 * 3)(36, 36, 36)
    p.max_steps, p.tile_w, p.tile_h = 255, 1, 1
    st = engine._lib.stepth_depth_from_additional(engine._h, C.byref(small), C.byref(big),
                                                  C.byref(p), None, 0, 0, C.byref(depth), None)
    assert st == L.ERR_BAD_ARG
    assert b"2^30" in engine._lib.stepth_last_error(engine._h)


def test_randomised_small_cases(engine, oracle):
    """300 random small configurations (sizes, unequal image sizes, precision per channel incl.
    0 and 255, max_steps incl. 0/1 and > 255, tiles, strips, patterns, noise vs smooth) --
    bit-exact against the oracle"""
    from stepth_amd import MaxZeroError, synth
    rng = np.random.default_rng(2024)
    smooth = synth.stereo_pair(96, 80, 16, seed=77)
    for case in range(300):
        w, h = int(rng.integers(1, 48)), int(rng.integers(1, 40))
        wa, ha = (w, h) if rng.random() < 0.6 else (int(rng.integers(1, 60)), int(rng.integers(1, 50)))
        if rng.random() < 0.5:
            main = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            add = rng.integers(0, 256, (ha, wa, 3), dtype=np.uint8)
            if rng.random() < 0.5:  # coarse colours -> plenty of far matches
                main, add = main // 64 * 64, add // 64 * 64
        else:
            main = np.ascontiguousarray(smooth[0][:h, :w])
            add = np.ascontiguousarray(smooth[1][:ha, :wa])
        prec = [int(v) for v in rng.choice([0, 1, 2, 9, 36, 80, 255], 3)] if rng.random() < 0.3 \
            else [int(rng.integers(1, 90))] * 3
        ms = int(rng.choice([0, 1, 2, 3, 8, 9, 24, 25, 26, 64, 255, 300]))
        tile = (1, 1) if rng.random() < 0.6 else (int(rng.integers(1, 7)), int(rng.integers(1, 7)))
        ref = oracle.depth_from_additional(main, add, prec, ms, tile)
        kw = dict(max_steps=ms, tile=tile)
        mode = rng.integers(0, 3)
        if mode == 0:  # whole frame
            try:
                d, mx = engine.depth_from_additional(main, add, prec, **kw)
                assert ref["status"] == 0 and mx == ref["max"], (case, w, h, prec, ms, tile)
                np.testing.assert_array_equal(d, ref["depth"], err_msg=str((case, w, h, prec, ms, tile)))
            except MaxZeroError:
                assert ref["status"] == oracle.ERR_MAX_ZERO, (case, w, h, prec, ms, tile)
        else:
            out = np.zeros((h, w), np.uint8)
            maxes = []
            if mode == 1:  # contiguous strips, tile aligned
                cuts = sorted(set([0, h] + [int(c) // tile[1] * tile[1] for c in rng.integers(0, h + 1, 2)]))
                parts = [dict(rows=(a, b)) for a, b in zip(cuts, cuts[1:])]
            else:          # interleaved blocks
                blk = 2 * tile[1] * int(rng.integers(1, 4))
                mod = int(rng.integers(1, 4))
                parts = [dict(row_pattern=(blk, mod, r)) for r in range(mod)]
            for part in parts:
                _, mx = engine.depth_from_additional(main, add, prec, raw_only=True, out=out,
                                                     **kw, **part)
                maxes.append(mx)
            np.testing.assert_array_equal(out, ref["raw"], err_msg=str((case, w, h, prec, ms, tile, parts)))
            assert max(maxes) == ref["max"]


def test_host_batch_pipeline_matches_oracle(engine, oracle):
    """stepth_depth_from_additional_batch on host memory (pageable and page-locked): chunks are
    uploaded / computed / downloaded on overlapping streams; every frame bit-exact."""
    from stepth_amd import synth
    pairs = [synth.stereo_pair(200 + 0 * i, 120, 24, seed=100 + i)[:2] for i in range(19)]
    mains = [np.ascontiguousarray(l) for l, _ in pairs]
    adds = [np.ascontiguousarray(r) for _, r in pairs]
    want = [oracle.depth_from_additional(m, a, [30, 30, 30], 40, (1, 1)) for m, a in zip(mains, adds)]
    outs, mx = engine.depth_from_additional_batch(mains, adds, [30, 30, 30], max_steps=40)
    for f in range(len(mains)):
        assert mx[f] == want[f]["max"] and np.array_equal(outs[f], want[f]["depth"]), f
    # page-locked inputs/outputs, RGBA mains with padded rows, strided outputs
    pm = [engine.host_array((120, 208, 4))[:, :200] for _ in mains]
    pa = [engine.host_array((120, 200, 3)) for _ in mains]
    po = [engine.host_array((120, 256))[:, 16:216] for _ in mains]
    for f in range(len(mains)):
        pm[f][..., :3] = mains[f]
        pm[f][..., 3] = 255
        pa[f][...] = adds[f]
        po[f][...] = 99
    outs, mx = engine.depth_from_additional_batch(pm, pa, [30, 30, 30], max_steps=40, outs=po)
    for f in range(len(mains)):
        assert mx[f] == want[f]["max"] and np.array_equal(po[f], want[f]["depth"]), f
    # raw-only strips through the same pipeline: only the strip's rows come back, the caller's
    # other rows stay as they were
    outs = [np.full((120, 200), 99, np.uint8) for _ in range(3)]
    outs, mx = engine.depth_from_additional_batch(mains[:3], adds[:3], [30, 30, 30], max_steps=40,
                                                  rows=(30, 91), raw_only=True, outs=outs)
    for f in range(3):
        raw = oracle.match_groups(oracle.uniform_groups(mains[f], 1, 1), adds[f], 40,
                                  [30, 30, 30])["raw"].reshape(120, 200)
        assert np.array_equal(outs[f][30:91], raw[30:91]) and mx[f] == int(raw[30:91].max())
        assert (outs[f][:30] == 99).all() and (outs[f][91:] == 99).all()
    # interleaved row blocks on host memory: the rows of the other ranks come back unchanged
    outs = [np.full((120, 200), 77, np.uint8) for _ in range(3)]
    outs, mx = engine.depth_from_additional_batch(mains[:3], adds[:3], [30, 30, 30], max_steps=40,
                                                  rows=(0, 120), row_pattern=(8, 3, 1), raw_only=True,
                                                  outs=outs)
    own = ((np.arange(120) // 8) % 3) == 1
    for f in range(3):
        raw = oracle.match_groups(oracle.uniform_groups(mains[f], 1, 1), adds[f], 40,
                                  [30, 30, 30])["raw"].reshape(120, 200)
        assert np.array_equal(outs[f][own], raw[own]) and (outs[f][~own] == 77).all()
    # mixing host and device memory is refused
    import torch
    with pytest.raises(Exception):
        engine.depth_from_additional_batch([mains[0], torch.from_numpy(mains[1]).cuda()], adds[:2],
                                           [30, 30, 30], outs=[np.zeros((120, 200), np.uint8)] * 2)


def test_host_batch_flat_frame_reports_max_zero(engine):
    from stepth_amd import MaxZeroError
    flat = np.full((40, 60, 3), 90, np.uint8)
    with pytest.raises(MaxZeroError):
        engine.depth_from_additional_batch([flat, flat], [flat, flat], [36] * 3)


@pytest.mark.parametrize("w,h", [(20000, 5), (5, 20000), (65535, 1), (1, 65535)])
def test_extreme_aspect_ratios(engine, oracle, w, h):
    """the C-ABI admits sides up to 65535: very wide / very tall images (rings clipped on one
    axis almost immediately, block grids of one row or one column)"""
    rng = np.random.default_rng(w * 31 + h)
    main = (rng.integers(0, 4, (h, w, 3)) * 60).astype(np.uint8)
    add = np.roll(main, 7 if w > h else 0, axis=1)
    add = np.roll(add, 7 if h > w else 0, axis=0).copy()
    add[rng.random((h, w)) < 0.02] = 255
    for ms in (40, 255):
        ref = oracle.depth_from_additional(main, add, [36] * 3, ms, (1, 1))
        d, mx = engine.depth_from_additional(main, add, [36] * 3, max_steps=ms)
        assert mx == ref["max"] and np.array_equal(d, ref["depth"]), (w, h, ms)
    g = oracle.adaptive_groups(main, [36] * 3)
    got = engine.adaptive_groups(main, [36] * 3)
    assert np.array_equal(np.sort(g, order=["y", "x"]), np.sort(got, order=["y", "x"]))
    refg = oracle.depth_from_additional(main, add, [36] * 3, 255, groups=g)
    dg, _ = engine.depth_from_additional(main, add, [36] * 3, groups=got)
    assert np.array_equal(dg, refg["depth"])


def test_two_engines_on_two_threads(oracle):
    """different DepthImages may be processed from different threads (the crate
    has no globals) -> the C library is re-entrant per handle: one engine per thread, calls
    overlapping in time (ctypes releases the GIL), each result exact."""
    import threading
    from stepth_amd import Engine, synth
    pairs = [synth.stereo_pair(240, 160, 32, seed=40 + i)[:2] for i in range(2)]
    want = [oracle.depth_from_additional(l, r, [36] * 3, 255)["depth"] for l, r in pairs]
    errors = []

    def work(i):
        try:
            with Engine(0) as eng:
                l, r = pairs[i]
                for _ in range(20):
                    d, _mx = eng.depth_from_additional(l, r, [36] * 3)
                    if not np.array_equal(d, want[i]):
                        errors.append(f"thread {i}: depth differs")
                        return
                    m, _z = eng.select_foreground(255 - d)
                    if not np.array_equal(m, oracle.select_foreground(255 - want[i])["mask"]):
                        errors.append(f"thread {i}: mask differs")
                        return
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {i}: {e!r}")

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not errors, errors


def test_multirank_rehearsal_on_the_engine():
    """tests/tools/rehearse_multirank.py: 3 ranks (gloo) sharing this GPU run the sharded flow
    on the real engine -- work-weighted strips, pipelined gather, steps alternating over two
    streams -- and every frame of every step equals the single-process result"""
    import subprocess
    import sys
    from conftest import ROOT
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "tools", "rehearse_multirank.py")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "mismatching frames 0" in r.stdout and "identical on all ranks: True" in r.stdout


def test_batch_with_gaussian_tail(engine, oracle):
    """STEPTH_FLAG_GAUSSIAN_TAIL in the batch entry: the reference's complete output (depth map +
    same-size Gaussian resample, src/depth_image.rs:125-134) for many frames, device and host"""
    import torch
    from stepth_amd import synth
    pairs = [synth.stereo_pair(161, 90, 24, seed=300 + i)[:2] for i in range(5)]
    want = [oracle.gaussian_same_size(oracle.depth_from_additional(l, r, [36] * 3, 60)["depth"]) for l, r in pairs]
    dl = [torch.from_numpy(np.ascontiguousarray(l)).cuda() for l, _ in pairs]
    dr = [torch.from_numpy(np.ascontiguousarray(r)).cuda() for _, r in pairs]
    outs, _mx = engine.depth_from_additional_batch(dl, dr, [36] * 3, max_steps=60, gaussian_tail=True)
    for f in range(5):
        assert np.array_equal(outs[f].cpu().numpy(), want[f]), f
    hl = [np.ascontiguousarray(l) for l, _ in pairs]
    hr = [np.ascontiguousarray(r) for _, r in pairs]
    outs, _mx = engine.depth_from_additional_batch(hl, hr, [36] * 3, max_steps=60, gaussian_tail=True)
    for f in range(5):
        assert np.array_equal(outs[f], want[f]), f
    with pytest.raises(Exception):  # the tail needs whole normalised frames
        engine.depth_from_additional_batch(dl, dr, [36] * 3, gaussian_tail=True, rows=(0, 40))
    with pytest.raises(Exception):
        engine.depth_from_additional_batch(dl, dr, [36] * 3, gaussian_tail=True, raw_only=True)


def test_batch_rgba_mains_unequal_additional_and_per_channel_precision(engine, oracle):
    """batch entry on what DepthImage really holds -- RGBA mains (src/depth_image.rs:8) with
    padded rows -- against additional images of another size, per-channel precision, 70 frames
    (two chunks of the 64-frame launch)"""
    import torch
    rng = np.random.default_rng(77)
    mains, adds = [], []
    for i in range(7):
        mains.append((rng.integers(0, 256, (41, 67, 3)) // 40 * 40).astype(np.uint8))
        adds.append((rng.integers(0, 256, (58, 35, 3)) // 40 * 40).astype(np.uint8))
    prec = [30, 50, 9]
    refs = [oracle.depth_from_additional(m, a, prec, 255) for m, a in zip(mains, adds)]
    rgba_store = [torch.zeros((41, 72, 4), dtype=torch.uint8, device="cuda") for _ in mains]
    dm = []
    for st, m in zip(rgba_store, mains):
        st[:, :67, :3] = torch.from_numpy(m).cuda()
        st[:, :67, 3] = 200
        dm.append(st[:, :67])                     # row stride 72*4 bytes, 4 channels
    da = [torch.from_numpy(a).cuda() for a in adds]
    many_m, many_a = [dm[i % 7] for i in range(70)], [da[i % 7] for i in range(70)]
    outs, mx = engine.depth_from_additional_batch(many_m, many_a, prec)
    for i in range(70):
        ref = refs[i % 7]
        if ref["status"] == 0:
            assert mx[i] == ref["max"] and np.array_equal(outs[i].cpu().numpy(), ref["depth"]), i


def test_64_frame_1080p_device_batch_equals_per_frame_calls(engine, oracle):
    """BASELINE config 4's per-node workload: 64 x 1920x1080 frames through ONE batched call
    (one set of launches) == 64 single-frame calls, and sampled pixels of two of them == oracle"""
    import torch
    from stepth_amd import synth
    base = [synth.stereo_pair(1920, 1080, 128, seed=0x5EED + i)[:2] for i in range(2)]
    fl = [lambda a: a, lambda a: a[:, ::-1], lambda a: a[::-1], lambda a: a[::-1, ::-1]]
    frames = []
    for i in range(64):
        l, r = base[i % 2]
        f = fl[(i // 2) % 4]
        # beyond the 8 distinct flips: shift the additional image by a few columns, so that all
        # 64 frames differ
        k = i // 8
        frames.append((np.ascontiguousarray(f(l)), np.ascontiguousarray(np.roll(f(r), k, axis=1))))
    mains = [torch.from_numpy(l).cuda() for l, _ in frames]
    adds = [torch.from_numpy(r).cuda() for _, r in frames]
    outs, mx = engine.depth_from_additional_batch(mains, adds, P36, max_steps=128, raw_only=True)
    torch.cuda.synchronize()
    for i in range(64):
        one, m1 = engine.depth_from_additional(mains[i], adds[i], P36, max_steps=128, raw_only=True)
        assert m1 == mx[i] and torch.equal(one, outs[i]), i
    for i in (5, 62):
        _sampled_raw_equals_oracle(oracle, outs[i].cpu().numpy(), frames[i][0], frames[i][1], P36, 128,
                                   n=8000, seed=i)


def test_heavy_first_launch_order_changes_nothing_but_the_order(engine, oracle, asset):
    """STEPTH_FLAG_HEAVY_FIRST (tile_cost / tile_scan / tile_scatter kernels: the search's
    workgroups start heaviest first): same bytes as image order -- a 1080p batch of a natural
    frame (the reference's pair, mirror-tiled: the order is taken) and a synthetic one (most tiles
    search far: the device keeps image order), RGBA mains, a row strip, the single-frame entry,
    reach 255 (the estimate then surveys superblocks) -- and the natural frame against the oracle"""
    import torch
    from stepth_amd import synth
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    from bench import mirror_tile
    W, H = 1920, 1080
    nat = (mirror_tile(asset["main"][..., :3], W, H), mirror_tile(asset["additional"][..., :3], W, H))
    syn = synth.stereo_pair(W, H, 128, seed=0x5EED)[:2]
    dev = torch.device("cuda", 0)
    for ms in (128, 255):
        for frames in ([nat, nat], [syn, nat]):
            mains = [torch.from_numpy(np.ascontiguousarray(l)).to(dev) for l, _ in frames]
            adds = [torch.from_numpy(np.ascontiguousarray(r)).to(dev) for _, r in frames]
            a, ma = engine.depth_from_additional_batch(mains, adds, P36, max_steps=ms)
            b, mb = engine.depth_from_additional_batch(mains, adds, P36, max_steps=ms, heavy_first=True)
            assert ma == mb
            for x, y in zip(a, b):
                assert torch.equal(x, y)
    raw, _ = engine.depth_from_additional_batch(mains[1:], adds[1:], P36, max_steps=255, heavy_first=True, raw_only=True)
    _sampled_raw_equals_oracle(oracle, raw[0].cpu().numpy(), nat[0], nat[1], P36, 255)
    # RGBA main + a row strip + the single-frame entry
    rgba = np.concatenate([nat[0], np.full((H, W, 1), 255, np.uint8)], 2)
    d0, m0 = engine.depth_from_additional(rgba, nat[1], P36, max_steps=128)
    d1, m1 = engine.depth_from_additional(rgba, nat[1], P36, max_steps=128, heavy_first=True)
    assert m0 == m1 and np.array_equal(d0, d1)
    out0, out1 = np.zeros((H, W), np.uint8), np.zeros((H, W), np.uint8)
    engine.depth_from_additional(nat[0], nat[1], P36, max_steps=128, rows=(135, 1080), raw_only=True, out=out0)
    engine.depth_from_additional(nat[0], nat[1], P36, max_steps=128, rows=(135, 1080), raw_only=True, out=out1, heavy_first=True)
    assert np.array_equal(out0, out1) and out0[135:].any()


def test_group_list_with_empty_rectangles(engine, oracle):
    """a group of zero width or height paints nothing (like the reference's collect()) but is
    still matched and counts for the max (src/depth_image.rs:112-124); host and device lists"""
    import torch
    from stepth_amd import synth
    l, r, _ = synth.stereo_pair(96, 64, 24, seed=21)
    g = oracle.uniform_groups(l, 8, 8).copy()
    extra = np.zeros(4, oracle.GROUP_DTYPE)
    extra["x"], extra["y"] = [10, 40, 95, 0], [5, 20, 63, 0]
    extra["w"], extra["h"] = [0, 5, 0, 0], [3, 0, 0, 0]
    extra["value"] = [[250, 5, 5], [5, 250, 5], [1, 2, 3], [200, 200, 200]]  # colours that match far away or never
    g = np.concatenate([g[:20], extra, g[20:]])
    ref = run_both(engine, oracle, l, r, P36, groups=g)
    # the batched entry with the list in device memory (it cannot be validated on the host)
    mains, adds = [torch.from_numpy(l).cuda()], [torch.from_numpy(r).cuda()]
    gd = torch.from_numpy(g.view(np.uint8).reshape(-1, 20).copy()).cuda()
    outs, mx = engine.depth_from_groups_batch(mains, adds, P36, [gd], [len(g)])
    assert mx == [ref["max"]]
    np.testing.assert_array_equal(outs[0].cpu().numpy(), ref["depth"])


def test_explicit_stream_choice_is_kept_and_switches_do_not_block(engine, oracle):
    """set_stream() is an explicit choice: CUDA-tensor calls stay on it whatever torch's current
    stream is, until reset_stream(); binding another stream orders it after the old one's work
    on the device (the result of a call enqueued right after a switch is still the oracle's)"""
    import torch
    from stepth_amd import synth
    l, r, _ = synth.stereo_pair(320, 180, 32, seed=5)
    ref = oracle.depth_from_additional(l, r, P36, 255)
    tl, tr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    try:
        engine.set_stream(s1.cuda_stream)
        with torch.cuda.stream(s2):  # torch's current stream differs: the explicit choice wins
            a, _ = engine.depth_from_additional_batch([tl], [tr], P36, asynchronous=True)
        assert engine._bound == s1.cuda_stream
        s1.synchronize()
        np.testing.assert_array_equal(a[0].cpu().numpy(), ref["depth"])
        for s in (s2, s1, s2):  # alternate: every switch is an event wait, results stay right
            engine.set_stream(s.cuda_stream)
            b, _ = engine.depth_from_additional_batch([tl], [tr], P36, asynchronous=True)
        s2.synchronize()
        np.testing.assert_array_equal(b[0].cpu().numpy(), ref["depth"])
    finally:
        engine.reset_stream()
    with torch.cuda.stream(s2):  # following torch's current stream again
        c, _ = engine.depth_from_additional_batch([tl], [tr], P36, asynchronous=True)
        assert engine._bound == s2.cuda_stream
    s2.synchronize()
    np.testing.assert_array_equal(c[0].cpu().numpy(), ref["depth"])
    engine.reset_stream()


def test_strip_worker_notices_a_frame_list_refilled_in_place(engine, oracle):
    """EngineStripWorker caches the descriptions of a frame set by list object; a caller that
    puts new tensors into the same list must not get the old frames' depth maps"""
    import torch
    from stepth_amd import synth
    from stepth_amd.sharding import EngineStripWorker
    engine.use_torch_stream()
    w = EngineStripWorker(engine, P36, max_steps=64, tile=(1, 1), mode=0)
    pairs = [synth.stereo_pair(160, 96, 24, seed=s)[:2] for s in (1, 2, 3)]
    frames = [(torch.from_numpy(pairs[0][0]).cuda(), torch.from_numpy(pairs[0][1]).cuda()),
              (torch.from_numpy(pairs[1][0]).cuda(), torch.from_numpy(pairs[1][1]).cuda())]
    d0, _ = w.depth_batch(frames)
    first = d0.cpu().numpy().copy()
    frames[1] = (torch.from_numpy(pairs[2][0]).cuda(), torch.from_numpy(pairs[2][1]).cuda())  # same list object
    d1, _ = w.depth_batch(frames)
    got = d1.cpu().numpy()
    for f, p in ((0, pairs[0]), (1, pairs[2])):
        np.testing.assert_array_equal(got[f], oracle.depth_from_additional(p[0], p[1], P36, 64)["depth"])
    np.testing.assert_array_equal(first[1], oracle.depth_from_additional(pairs[1][0], pairs[1][1], P36, 64)["depth"])
    engine.reset_stream()
