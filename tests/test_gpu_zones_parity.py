"""GPU parity for the consumers of a finished depth map through the C-ABI:
stepth_invert_depth / stepth_depth_split / stepth_slice / stepth_select_foreground(_batch) /
stepth_apply_mask against the oracle (src/depth_image.rs:162-247, src/mask_image.rs:205-213).
Byte work: everything is compared bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rgba(rgb):
    return np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=2)


def _depth_maps():
    """name -> [H,W] u8: shapes with ragged tails, flat areas, noise, few levels"""
    rng = np.random.default_rng(2024)
    yy, xx = np.mgrid[0:173, 0:251]
    maps = {
        "noise_251x173": rng.integers(0, 256, (173, 251)).astype(np.uint8),
        "ramp_251x173": ((xx * 255) // 250).astype(np.uint8),
        "blobs_640x360": np.clip(40 + 180 * np.exp(-(((np.mgrid[0:360, 0:640][1] - 300) / 90.0) ** 2 +
                                                       ((np.mgrid[0:360, 0:640][0] - 200) / 70.0) ** 2))
                                 + rng.normal(0, 6, (360, 640)), 0, 255).astype(np.uint8),
        "three_levels_33x17": rng.choice(np.array([5, 90, 250], np.uint8), (17, 33)),
        "mostly_zero_512x64": np.where(rng.random((64, 512)) < 0.71, 0,
                                       rng.integers(1, 256, (64, 512))).astype(np.uint8),
        "one_row_1000": rng.integers(0, 256, (1, 1000)).astype(np.uint8),
        "one_col_1000": rng.integers(0, 256, (1000, 1)).astype(np.uint8),
        "single_pixel_pair": np.array([[3, 250]], np.uint8),
    }
    return maps


def test_readme_second_half_on_the_gpu(engine, oracle, asset):
    """Readme.md:18-24 through the host mirror: open_depth, invert_depth, select_foreground,
    apply_mask -- bit-exact against the oracle (which is pinned to assets/foreground.jpg)."""
    from stepth_amd import DepthImage
    img = DepthImage.from_image(asset["main"])
    img.load_depth(asset["depth"].copy())
    img.invert_depth(engine=engine)
    want_d = oracle.invert_depth(asset["depth"])
    assert np.array_equal(img.depth, want_d)
    mask = img.select_foreground(engine=engine)
    want = oracle.select_foreground(want_d)
    assert np.array_equal(mask.mask, want["mask"])
    mask.apply_mask(engine=engine)
    assert np.array_equal(mask.image, oracle.apply_mask(_rgba(asset["main"]), want["mask"]))
    assert img.depth_split(2, engine=engine) == [(0, 201), (202, 255)]


@pytest.mark.parametrize("name", list(_depth_maps()))
def test_depth_split_matches_oracle(engine, oracle, name):
    from stepth_amd import SplitError
    d = _depth_maps()[name]
    for zones in (0, 1, 2, 3, 4, 5, 8, 13, 32, 100, 255):
        ref = oracle.depth_split(d, zones)
        try:
            got = (0, engine.depth_split(d, zones))
        except SplitError as e:
            got = (e.status, [])
        assert got == (ref["status"], ref["zones"]), (name, zones)


def test_split_statuses_where_the_reference_panics(engine, oracle):
    from stepth_amd import SplitError
    cases = [(np.full((9, 31), 7, np.uint8), 2, -7), (np.array([[3, 4]], np.uint8), 2, -7),
             (np.array([[0, 255]], np.uint8), 200, -7), (np.array([[100, 250]], np.uint8), 3, -8)]
    for d, zones, status in cases:
        assert oracle.depth_split(d, zones)["status"] == status
        with pytest.raises(SplitError) as ei:
            engine.depth_split(d, zones)
        assert ei.value.status == status
        if zones == 2:
            with pytest.raises(SplitError):
                engine.select_foreground(d)
    with pytest.raises(Exception):
        engine.depth_split(np.zeros((4, 4), np.uint8), 256)  # zones is a u8


@pytest.mark.parametrize("name", list(_depth_maps()))
def test_slice_invert_select_host_and_device(engine, oracle, name):
    import torch
    d = _depth_maps()[name]
    for rng_ in ((None, None), (10, 200), (None, 77), (128, None), (200, 10), (0, 0), (255, 255)):
        want = oracle.slice_mask(d, *rng_)
        assert np.array_equal(engine.slice(d, *rng_), want), (name, rng_)
        assert np.array_equal(engine.slice(torch.from_numpy(d).cuda(), *rng_).cpu().numpy(), want)
    h = d.copy()
    engine.invert_depth(h)
    assert np.array_equal(h, oracle.invert_depth(d))
    t = torch.from_numpy(d).cuda()
    engine.invert_depth(t)
    assert np.array_equal(t.cpu().numpy(), 255 - d)
    ref = oracle.select_foreground(d)
    if ref["status"] == 0:
        m, zone = engine.select_foreground(d)
        assert zone == ref["zone"] and np.array_equal(m, ref["mask"])
        m, zone = engine.select_foreground(torch.from_numpy(d).cuda())
        assert zone == ref["zone"] and np.array_equal(m.cpu().numpy(), ref["mask"])


def test_strided_and_unaligned_device_planes(engine, oracle):
    """rows with padding and a base pointer that is not 16-byte aligned take the byte path"""
    import torch
    rng = np.random.default_rng(5)
    big = torch.from_numpy(rng.integers(0, 256, (97, 300)).astype(np.uint8)).cuda()
    for view in (big[:, 1:250], big[3:, 16:272], big[:, :299], big[5:6, 7:]):
        d = view.cpu().numpy()
        out = torch.zeros((97, 320), dtype=torch.uint8, device="cuda")[: view.shape[0], 3:3 + view.shape[1]]
        engine.slice(view, 50, 180, out=out)
        assert np.array_equal(out.cpu().numpy(), oracle.slice_mask(d, 50, 180))
        assert engine.depth_split(view, 3) == oracle.depth_split(d, 3)["zones"]
        m, zone = engine.select_foreground(view)
        assert np.array_equal(m.cpu().numpy(), oracle.select_foreground(d)["mask"])
        before = big.clone()
        engine.invert_depth(view)
        after = big.cpu().numpy()
        expect = before.cpu().numpy()
        sl = (slice(view.storage_offset() // 300, view.storage_offset() // 300 + view.shape[0]),
              slice(view.storage_offset() % 300, view.storage_offset() % 300 + view.shape[1]))
        expect[sl] = 255 - expect[sl]
        assert np.array_equal(after, expect)  # nothing outside the view was touched
        engine.invert_depth(view)


def test_apply_mask_shapes(engine, oracle):
    import torch
    rng = np.random.default_rng(9)
    for (h, w) in ((1, 1), (3, 5), (17, 64), (40, 203), (360, 640)):
        rgba = rng.integers(1, 256, (h, w, 4)).astype(np.uint8)
        mask = rng.choice(np.array([0, 0, 255, 255, 255, 7], np.uint8), (h, w))
        if h > 8:
            mask[4:8, :] = 0      # whole quads cleared
            mask[0:2, :] = 255    # whole quads kept
        want = oracle.apply_mask(rgba, mask)
        got = rgba.copy()
        engine.apply_mask(got, mask)
        assert np.array_equal(got, want), (h, w)
        t = torch.from_numpy(rgba).cuda()
        engine.apply_mask(t, torch.from_numpy(mask).cuda())
        assert np.array_equal(t.cpu().numpy(), want), (h, w)
    with pytest.raises(Exception):
        engine.apply_mask(np.zeros((4, 4, 4), np.uint8), np.zeros((4, 5), np.uint8))
    with pytest.raises(Exception):
        engine.slice(np.zeros((4, 4), np.uint8), 1, 2, out=np.zeros((5, 4), np.uint8))


def test_select_foreground_batch_with_failing_frames(engine, oracle):
    import torch
    from stepth_amd import SplitError
    rng = np.random.default_rng(17)
    frames = [rng.integers(0, 256, (90, 160)).astype(np.uint8) for _ in range(5)]
    frames[1] = np.full((90, 160), 9, np.uint8)                  # split fails: -7
    frames[3] = (frames[3] // 64 * 64).astype(np.uint8)          # few levels
    frames += [np.clip(rng.normal(100, 40, (90, 160)), 0, 255).astype(np.uint8) for _ in range(36)]
    ts = [torch.from_numpy(f).cuda() for f in frames]            # 41 frames: two launches' worth
    outs = [torch.full_like(t, 77) for t in ts]
    with pytest.raises(SplitError):
        engine.select_foreground_batch(ts, [o.clone() for o in outs])
    masks, zones, statuses = engine.select_foreground_batch(ts, outs, raise_on_error=False)
    for f, d in enumerate(frames):
        ref = oracle.select_foreground(d)
        assert statuses[f] == ref["status"], f
        if ref["status"] == 0:
            assert zones[f] == ref["zone"] and np.array_equal(masks[f].cpu().numpy(), ref["mask"]), f
        else:
            assert zones[f] is None and (masks[f] == 77).all()   # mask left as it was
    # asynchronous form on the engine's stream, then sync
    outs2 = [torch.zeros_like(t) for t in ts]
    engine.select_foreground_batch(ts, outs2, asynchronous=True)
    engine.sync()
    for f in (0, 2, 40):
        assert np.array_equal(outs2[f].cpu().numpy(), masks[f].cpu().numpy())


def test_randomised_zone_cases(engine, oracle):
    from stepth_amd import SplitError
    rng = np.random.default_rng(99)
    for case in range(150):
        h, w = int(rng.integers(1, 80)), int(rng.integers(1, 300))
        kind = case % 4
        if kind == 0:
            d = rng.integers(0, 256, (h, w))
        elif kind == 1:
            k = int(rng.integers(1, 5))
            c = rng.integers(0, 256, k)
            d = np.clip(rng.normal(c[rng.integers(0, k, (h, w))], rng.integers(1, 25)), 0, 255)
        elif kind == 2:
            lv = rng.integers(0, 256, int(rng.integers(1, 6)))
            d = lv[rng.integers(0, len(lv), (h, w))]
        else:
            lo = int(rng.integers(0, 250))
            d = rng.integers(lo, min(256, lo + int(rng.integers(1, 60))), (h, w))
        d = d.astype(np.uint8)
        zones = int(rng.integers(0, 20))
        ref = oracle.depth_split(d, zones)
        try:
            got = (0, engine.depth_split(d, zones))
        except SplitError as e:
            got = (e.status, [])
        assert got == (ref["status"], ref["zones"]), (case, h, w, zones)
        sf = oracle.select_foreground(d)
        if sf["status"] == 0:
            m, zone = engine.select_foreground(d)
            assert zone == sf["zone"] and np.array_equal(m, sf["mask"]), case


def test_properties_at_8k(engine):
    """BASELINE's largest size (7680x4320): size-independent properties instead of the oracle."""
    import torch
    # outputs are torch allocations that get recycled while torch kernels of the previous
    # trip may still read them: share torch's stream so that everything is stream-ordered
    engine.use_torch_stream()
    try:
        _properties_at_8k(engine)
    finally:
        engine.reset_stream()


def _properties_at_8k(engine):
    import torch
    g = torch.Generator(device="cuda").manual_seed(3)
    d = torch.randint(0, 256, (4320, 7680), dtype=torch.uint8, device="cuda", generator=g)
    d[:1000] = d[:1000] // 3                         # a skewed histogram
    ref = d.clone()
    engine.invert_depth(d)
    assert bool((d == 255 - ref).all())
    engine.invert_depth(d)
    assert bool((d == ref).all())                    # involution
    zones = engine.depth_split(d, 4)
    assert zones[0][0] == int(d.min()) and zones[-1][1] == int(d.max())
    covered = torch.zeros_like(d, dtype=torch.int32)
    for lo, hi in zones:                             # zones partition the value range
        m = engine.slice(d, lo, hi)
        assert bool(((m == 255) == ((d >= lo) & (d <= hi))).all())
        covered += (m == 255).int()
    assert bool((covered == 1).all())
    m, (lo, hi) = engine.select_foreground(d)
    assert (lo, hi) == engine.depth_split(d, 2)[0]
    assert bool(((m == 255) == ((d >= lo) & (d <= hi))).all())
    # the histogram behind the split counts every pixel once: zone sizes add up
    torch_hist = torch.bincount(d.flatten().int(), minlength=256)
    assert int(torch_hist.sum()) == d.numel()


def test_histogram_halves_compose_to_depth_split(engine, oracle):
    """stepth_depth_histogram + stepth_zones_from_histogram (the row-sharded form): strips of a
    depth map dealt to 8 'ranks', counts summed, zones == depth_split of the whole map."""
    import torch
    from stepth_amd.sharding import EngineZoneWorker, sharded_select_foreground, strip_bounds
    for name, d in _depth_maps().items():
        if d.shape[0] < 8:
            continue
        whole = engine.depth_histogram(d)
        assert np.array_equal(whole, np.bincount(d.ravel(), minlength=256)), name
        t = torch.from_numpy(d).cuda()
        total = torch.zeros(256, dtype=torch.int32, device="cuda")
        for r0, r1 in strip_bounds(d.shape[0], 8):
            if r1 > r0:
                total += engine.depth_histogram(t[r0:r1])
        assert np.array_equal(total.cpu().numpy(), np.bincount(d.ravel(), minlength=256)), name
        for zones in (1, 2, 3, 7):
            ref = oracle.depth_split(d, zones)
            if ref["status"] == 0:
                assert engine.zones_from_histogram(total, zones) == ref["zones"], (name, zones)
                assert engine.zones_from_histogram(whole, zones) == ref["zones"], (name, zones)
        # world-size-1 run of the sharded helper (the collective is skipped)
        ref = oracle.select_foreground(d)
        if ref["status"] == 0:
            engine.use_torch_stream()
            try:
                mask, zone = sharded_select_foreground(EngineZoneWorker(engine), t)
                torch.cuda.synchronize()
            finally:
                engine.reset_stream()
            assert tuple(zone) == ref["zone"] and np.array_equal(mask.cpu().numpy(), ref["mask"]), name
