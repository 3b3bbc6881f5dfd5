"""Adaptive pixel groups: stepth_adaptive_groups, the engine's stand-in for
disage::open::rgb_discrete (src/depth_image.rs:101-110).  disage's source is absent from the
reference tree, so the grouping rule is the engine's own design -- PARITY WITH DISAGE IS
UNPINNED; these tests pin the GPU against the oracle's statement of the same rule (as sets:
the order of the groups is arbitrary and no result depends on it) and check the properties the
reference's call site fixes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sorted(g):
    g = np.asarray(g)
    return g[np.lexsort((g["x"], g["y"]))]


def _as_groups(t):
    from stepth_amd.engine import GROUP_DTYPE
    return np.frombuffer(t.cpu().numpy().tobytes(), GROUP_DTYPE)


def _cases():
    from stepth_amd import synth
    rng = np.random.default_rng(8)
    yy, xx = np.mgrid[0:90, 0:130]
    flat = np.full((64, 64, 3), 77, np.uint8)
    return {
        "synthetic_320x180": synth.stereo_pair(320, 180, 32, seed=3)[0],
        "noise_97x61": rng.integers(0, 256, (61, 97, 3)).astype(np.uint8),
        "ramp_130x90": np.stack([(xx * 255 // 129), (yy * 255 // 89), ((xx + yy) % 256)], -1).astype(np.uint8),
        "flat_64x64": flat,
        "one_row_300": rng.integers(0, 40, (1, 300, 3)).astype(np.uint8),
        "one_col_300": rng.integers(0, 40, (300, 1, 3)).astype(np.uint8),
        "tiny_3x2": rng.integers(0, 256, (2, 3, 3)).astype(np.uint8),
        "single_pixel": np.array([[[1, 2, 3]]], np.uint8),
    }


@pytest.mark.parametrize("name", list(_cases()))
def test_groups_match_oracle_as_sets(engine, oracle, name):
    import torch
    img = _cases()[name]
    for prec, mn, mx in (([36] * 3, 16, 0), ([8, 20, 60], 4, 0), ([36] * 3, 0, 0), ([36] * 3, 6, 9),
                         ([255] * 3, 3, 0), ([0] * 3, 2, 0), ([36] * 3, 12, 5)):
        want = _sorted(oracle.adaptive_groups(img, prec, mn, mx))
        got = _sorted(engine.adaptive_groups(img, prec, mn, mx))
        assert len(got) == len(want), (name, prec, mn, mx)
        assert (got == want).all(), (name, prec, mn, mx)
        dev = _sorted(_as_groups(engine.adaptive_groups(torch.from_numpy(img).cuda(), prec, mn, mx)))
        assert (dev == want).all(), (name, prec, mn, mx)


def test_what_the_call_site_fixes(engine):
    """src/depth_image.rs:101-110: groups tile the image, >= 2^16 of them on an image of >= 2^16
    pixels (min_splits = 16), single pixels reachable (max_splits = ceil(log2(px)))"""
    from stepth_amd import synth
    img = synth.stereo_pair(640, 360, 64, seed=12)[0]
    g = engine.adaptive_groups(img, [36] * 3)
    cover = np.zeros(img.shape[:2], np.int32)
    for r in g:
        cover[r["y"]:r["y"] + r["h"], r["x"]:r["x"] + r["w"]] += 1
    assert (cover == 1).all()
    assert len(g) >= 1 << 16
    # 640*360 px -> max_splits = 18; only rectangles of <= 2 pixels can sit at that depth (and
    # be groups without being uniform), so every group of >= 3 pixels is uniform
    big = g[(g["w"].astype(int) * g["h"]) >= 3]
    assert len(big) > 1000
    for r in big[:: max(1, len(big) // 500)]:
        px = img[r["y"]:r["y"] + r["h"], r["x"]:r["x"] + r["w"]].reshape(-1, 3).astype(int)
        assert ((px.max(0) - px.min(0)) < 36).all()           # uniform within precision
        assert (r["value"] == px.sum(0) // len(px)).all()     # MeanBrightnessHasher stand-in
    # precision 0 is never uniform: everything is cut down to max_splits = 18 levels, where the
    # uneven halves leave rectangles of one or two pixels
    g0 = engine.adaptive_groups(img, [0] * 3)
    area = g0["w"].astype(int) * g0["h"]
    assert area.max() <= 2 and area.sum() == img.shape[0] * img.shape[1]


def test_reference_flow_with_adaptive_groups(engine, oracle, asset):
    """the whole path the way the reference runs it (grouping -> match -> normalise -> paint),
    groups staying on the device; equals the oracle on the same groups"""
    import torch
    main, add = asset["main"], asset["additional"]
    tm, ta = torch.from_numpy(main).cuda(), torch.from_numpy(add).cuda()
    groups = engine.adaptive_groups(tm, [36] * 3)
    depth, mx = engine.depth_from_additional(tm, ta, [36] * 3, groups=groups)
    want_groups = oracle.adaptive_groups(main, [36] * 3)
    ref = oracle.depth_from_additional(main, add, [36] * 3, 255, groups=want_groups)
    assert ref["status"] == 0 and mx == ref["max"]
    assert np.array_equal(depth.cpu().numpy(), ref["depth"])
    # and it still resembles the reference's shipped result (tolerance, like the 1x1 pin)
    r = np.corrcoef(ref["depth"].astype(float).ravel(), asset["depth"].astype(float).ravel())[0, 1]
    assert r > 0.9, r


@pytest.mark.gpu
def test_batched_group_flow_matches_oracle(engine, oracle):
    """the reference's real flow for several frames in one set of launches
    (stepth_adaptive_groups_batch -> stepth_depth_from_groups_batch): groups as sets, depth
    maps bit for bit, strips, raw-only; frames with different numbers of groups"""
    import torch
    from stepth_amd import synth
    pairs = [synth.stereo_pair(203, 117, 48, seed=40 + i)[:2] for i in range(5)]
    pairs.append((np.full((117, 203, 3), 80, np.uint8), pairs[0][1]))  # a flat frame: 2^16 groups at most
    mains = [torch.from_numpy(np.ascontiguousarray(l)).cuda() for l, _ in pairs]
    adds = [torch.from_numpy(np.ascontiguousarray(r)).cuda() for _, r in pairs]
    prec = [36] * 3
    bufs, counts = engine.adaptive_groups_batch(mains, prec, 12, 0)
    want_groups = [oracle.adaptive_groups(l, prec, 12, 0) for l, _ in pairs]
    for f in range(len(pairs)):
        got = bufs[f, :counts[f]].cpu().numpy().view(oracle.GROUP_DTYPE).ravel()
        got = got[np.lexsort((got["x"], got["y"]))]
        w = want_groups[f][np.lexsort((want_groups[f]["x"], want_groups[f]["y"]))]
        assert counts[f] == len(w) and (got == w).all(), f
    groups = [bufs[f] for f in range(len(pairs))]
    outs, mx = engine.depth_from_groups_batch(mains, adds, prec, groups, counts, max_steps=64)
    for f, (l, r) in enumerate(pairs):
        ref = oracle.depth_from_additional(l, r, prec, 64, groups=want_groups[f])
        assert mx[f] == ref["max"] and np.array_equal(outs[f].cpu().numpy(), ref["depth"]), f
    # a row strip, raw distances: groups whose top row lies in the strip are painted, the rest is 0
    outs, mx = engine.depth_from_groups_batch(mains, adds, prec, groups, counts, max_steps=64, rows=(30, 90),
                                              raw_only=True)
    for f, (l, r) in enumerate(pairs):
        g = want_groups[f]
        sel = g[(g["y"] >= 30) & (g["y"] < 90)]
        m = oracle.match_groups(sel, r, 64, prec)
        want = oracle.paint(sel, m["raw"], 203, 117)
        assert np.array_equal(outs[f].cpu().numpy()[30:90], want[30:90]), f
    # both in one call (stepth_adaptive_depth_batch: the counts come back behind the pack kernels),
    # with and without the Gaussian tail, and the two-call form of the helper
    outs2, mx2 = engine.adaptive_depth_batch(mains, adds, prec, max_steps=64, min_splits=12)
    assert engine.last_group_counts == [int(c) for c in counts]
    outs3, mx3 = engine.adaptive_depth_batch(mains[:3], adds[:3], prec, max_steps=64, min_splits=12, two_calls=True)
    outs4, _ = engine.adaptive_depth_batch(mains[:2], adds[:2], prec, max_steps=64, min_splits=12, gaussian_tail=True)
    for f, (l, r) in enumerate(pairs):
        ref = oracle.depth_from_additional(l, r, prec, 64, groups=want_groups[f])
        assert mx2[f] == ref["max"] and np.array_equal(outs2[f].cpu().numpy(), ref["depth"]), f
        if f < 3:
            assert mx3[f] == ref["max"] and np.array_equal(outs3[f].cpu().numpy(), ref["depth"]), f
        if f < 2:
            assert np.array_equal(outs4[f].cpu().numpy(), oracle.gaussian_same_size(ref["depth"])), f
    # a list with rectangles too large for the search kernel's own lanes (paint_big_kernel)
    big = np.zeros(3, oracle.GROUP_DTYPE)
    big["x"], big["y"], big["w"], big["h"] = [0, 100, 0], [0, 0, 60], [100, 103, 203], [60, 60, 57]
    big["value"] = [pairs[0][0][10, 10], pairs[0][0][20, 150], pairs[0][0][100, 100]]
    gd = torch.from_numpy(big.view(np.uint8).reshape(-1, 20).copy()).cuda()
    outs5, mx5 = engine.depth_from_groups_batch(mains[:1], adds[:1], prec, [gd], [3], max_steps=64)
    ref = oracle.depth_from_additional(pairs[0][0], pairs[0][1], prec, 64, groups=big)
    if ref["status"] == 0:
        assert mx5[0] == ref["max"] and np.array_equal(outs5[0].cpu().numpy(), ref["depth"])


@pytest.mark.gpu
def test_batched_uniform_tiles_match_oracle(engine, oracle):
    """stepth_depth_from_additional_batch with tiles larger than one pixel: all frames in one
    set of launches == the oracle frame by frame; strips; an interleaved row pattern"""
    import torch
    from stepth_amd import synth
    pairs = [synth.stereo_pair(203, 117, 48, seed=70 + i)[:2] for i in range(4)]
    mains = [torch.from_numpy(np.ascontiguousarray(l)).cuda() for l, _ in pairs]
    adds = [torch.from_numpy(np.ascontiguousarray(r)).cuda() for _, r in pairs]
    prec = [36] * 3
    for tile in ((5, 5), (2, 3), (16, 16), (64, 64)):
        outs, mx = engine.depth_from_additional_batch(mains, adds, prec, max_steps=64, tile=tile)
        for f, (l, r) in enumerate(pairs):
            ref = oracle.depth_from_additional(l, r, prec, 64, tile)
            assert mx[f] == ref["max"] and np.array_equal(outs[f].cpu().numpy(), ref["depth"]), (tile, f)
    # a tile larger than the image is the whole image (one group; raw-only: its distance may be 0)
    outs, mx = engine.depth_from_additional_batch(mains, adds, prec, max_steps=64, tile=(300, 4000), raw_only=True)
    for f, (l, r) in enumerate(pairs):
        ref = oracle.depth_from_additional(l, r, prec, 64, (203, 117))
        assert np.array_equal(outs[f].cpu().numpy(), ref["raw"]), f
    outs = [torch.full((117, 203), 7, dtype=torch.uint8, device="cuda") for _ in pairs]
    engine.depth_from_additional_batch(mains, adds, prec, max_steps=64, tile=(5, 5), rows=(20, 60),
                                       raw_only=True, outs=outs)
    for f, (l, r) in enumerate(pairs):
        raw = oracle.depth_from_additional(l, r, prec, 64, (5, 5))["raw"]
        got = outs[f].cpu().numpy()
        assert np.array_equal(got[20:60], raw[20:60]) and (got[:20] == 7).all() and (got[60:] == 7).all(), f


@pytest.mark.gpu
def test_long_reach_list_order_changes_nothing(engine, oracle, asset):
    """reach > 160 with >= 8192 waves of queries: ring_list_kernel's waves start heaviest first
    (list_cost / list_rank / tile_scan / tile_scatter kernels).  Same bytes as the unordered
    launch (the statistics variant keeps list order), one natural frame against the oracle"""
    import sys
    import torch
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    from bench import mirror_tile
    from stepth_amd import synth
    W, H = 1920, 1080
    nat = (mirror_tile(asset["main"][..., :3], W, H), mirror_tile(asset["additional"][..., :3], W, H))
    frames = [nat if f % 2 else synth.stereo_pair(W, H, 128, seed=0xA0 + f)[:2] for f in range(8)]
    mains = [torch.from_numpy(np.ascontiguousarray(l)).cuda() for l, _ in frames]
    adds = [torch.from_numpy(np.ascontiguousarray(r)).cuda() for _, r in frames]
    prec = [36] * 3
    a, ma = engine.adaptive_depth_batch(mains, adds, prec, max_steps=255)
    counts = engine.last_group_counts
    assert sum((c + 63) // 64 for c in counts) >= 8192, counts  # else the order is not taken
    engine.stats_enable(True)
    try:
        b, mb = engine.adaptive_depth_batch(mains, adds, prec, max_steps=255)
    finally:
        engine.stats_enable(False)
    assert list(ma) == list(mb)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    ref = oracle.depth_from_additional(nat[0], nat[1], prec, 255, groups=oracle.adaptive_groups(nat[0], prec, 16, 0))
    assert ref["status"] == 0 and ma[1] == ref["max"]
    assert np.array_equal(a[1].cpu().numpy(), ref["depth"])


@pytest.mark.gpu
def test_deepest_trees_of_the_one_launch_grouping(engine, oracle):
    """adaptive_tree_kernel at its limits: seeds of 16x16 pixels (8 levels below them) that all
    split down to single pixels (noise, precision 1: 256 seeds x 511 nodes per workgroup, far
    more gos than levels), the same with max_splits cutting the tree short, and one level more
    (the seed + level launches take over)"""
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    img[64:128, :96] = 77  # and a flat patch: leaves at every depth
    for prec, mn, mx in (([1] * 3, 8, 0), ([1] * 3, 8, 13), ([9, 1, 200], 9, 0), ([1] * 3, 7, 0), ([1] * 3, 6, 14)):
        want = oracle.adaptive_groups(img, prec, mn, mx)
        got = engine.adaptive_groups(img, prec, mn, mx)
        got = np.ascontiguousarray(got).view(oracle.GROUP_DTYPE).ravel()
        assert len(got) == len(want), (prec, mn, mx, len(got), len(want))
        k = lambda g: g[np.lexsort((g["h"], g["w"], g["x"], g["y"]))]
        assert (k(got) == k(np.ascontiguousarray(want))).all(), (prec, mn, mx)
