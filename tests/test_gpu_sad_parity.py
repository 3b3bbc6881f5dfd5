"""GPU parity of the windowed-SAD mode against the engine's own CPU statement
(oracle_sad_rows).  This mode does not exist in the reference: parity here
is self-consistency, bit-exact on the u8 disparity map."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run_sad(engine, l, r, d_max, win, prec, **kw):
    from stepth_amd import MODE_SAD_WINDOW
    return engine.depth_from_additional(l, r, prec, mode=MODE_SAD_WINDOW, d_max=d_max, win=win,
                                        **kw)


@pytest.mark.parametrize("w,h,d_max,win", [(160, 48, 24, 5), (200, 90, 64, 5), (131, 77, 128, 5),
                                            (96, 40, 32, 9), (64, 17, 256, 3), (70, 33, 16, 1),
                                            (5, 4, 8, 5)])
def test_sad_raw_matches_cpu_statement(engine, oracle, w, h, d_max, win):
    from stepth_amd import synth
    l, r, _ = synth.stereo_pair(w, h, min(d_max, 64), seed=w + d_max)
    for prec in ([0, 0, 0], [4, 4, 4], [36, 36, 36]):
        ref = oracle.sad_rows(l, r, d_max, win, prec)
        raw, mx = run_sad(engine, l, r, d_max, win, prec, raw_only=True)
        np.testing.assert_array_equal(raw, ref)
        assert mx == int(ref.max())
        if mx:
            depth, _ = run_sad(engine, l, r, d_max, win, prec)
            np.testing.assert_array_equal(depth, (ref.astype(np.uint32) * 255 // mx).astype(np.uint8))


def test_sad_rgba_left_and_strips(engine, oracle):
    from stepth_amd import synth
    l, r, _ = synth.stereo_pair(150, 61, 32, seed=5)
    ref = oracle.sad_rows(l, r, 32, 5, [3, 3, 3])
    rgba = np.concatenate([l, np.zeros((61, 150, 1), np.uint8)], axis=2)
    raw, _ = run_sad(engine, rgba, r, 32, 5, [3, 3, 3], raw_only=True)
    np.testing.assert_array_equal(raw, ref)
    out = np.zeros_like(ref)
    for rows in ((0, 20), (20, 21), (21, 61)):
        run_sad(engine, l, r, 32, 5, [3, 3, 3], raw_only=True, rows=rows, out=out)
    np.testing.assert_array_equal(out, ref)


def test_sad_device_images_at_any_alignment(engine, oracle):
    """the search fetches each pixel as the aligned dword pair around its three bytes (csrc/
    sad_kernels.h): device images whose first byte sits at every offset inside a dword, row
    pitches that are not multiples of 4 (the offsets then change from row to row), an RGBA main
    image, widths below a wave and below a dword -- each a view that ends where its buffer ends"""
    import torch
    from stepth_amd import MODE_SAD_WINDOW, synth
    rng = np.random.default_rng(11)

    def view(a, off):  # a device view at byte offset `off` of a buffer that ends with it
        b = torch.zeros(a.size + off, dtype=torch.uint8, device="cuda")
        v = b[off:].view(*a.shape)
        v.copy_(torch.from_numpy(a))
        return v

    for w, h, d_max, win in ((150, 33, 40, 5), (131, 20, 24, 9), (67, 19, 16, 3), (5, 7, 6, 5), (1, 9, 4, 3)):
        l, r, _ = synth.stereo_pair(max(w, 8), h, min(d_max, 16), seed=w)
        l, r = np.ascontiguousarray(l[:, :w]), np.ascontiguousarray(r[:, :w])
        want = oracle.sad_rows(l, r, d_max, win, [5, 5, 5])
        for off_l in range(4):
            off_r = (off_l * 3 + 1) & 3
            lm = np.concatenate([l, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], axis=2) if off_l & 1 else l
            outs, mx = engine.depth_from_additional_batch([view(lm, off_l)], [view(r, off_r)], [5, 5, 5],
                                                          mode=MODE_SAD_WINDOW, d_max=d_max, win=win, raw_only=True)
            np.testing.assert_array_equal(outs[0].cpu().numpy(), want, err_msg=f"{w}x{h} offsets {off_l},{off_r}")
            assert list(mx) == [int(want.max())]


def test_sad_larger_frame_band_and_planted_disparity(engine, oracle):
    from stepth_amd import synth
    l, r, disp = synth.stereo_pair(640, 360, 64, seed=2)
    raw, _ = run_sad(engine, l, r, 64, 5, [0, 0, 0], raw_only=True)
    ref = oracle.sad_rows(l, r, 64, 5, [0, 0, 0], 170, 190)
    np.testing.assert_array_equal(raw[170:190], ref[170:190])
    inner = (slice(8, -8), slice(72, -8))
    # piecewise-flat texture is ambiguous for a 5x5 window; measured 0.48
    assert (np.abs(raw[inner].astype(int) - disp[inner]) <= 1).mean() > 0.35


def test_sad_randomised_small_cases(engine, oracle):
    """80 random configurations: sizes (also narrower than a wave / the window), d_max not a
    multiple of the disparity chunk, every odd window 1..11, thresholds, strips"""
    rng = np.random.default_rng(7)
    for case in range(80):
        w, h = int(rng.integers(1, 150)), int(rng.integers(1, 40))
        d_max = int(rng.choice([1, 2, 7, 16, 17, 33, 64, 100, 128, 256]))
        win = int(rng.choice([1, 3, 5, 7, 9, 11]))
        l = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        shift = int(rng.integers(0, min(d_max, w)))
        r = np.roll(l, -shift, axis=1)
        r = np.clip(r.astype(int) + rng.integers(-3, 4, r.shape), 0, 255).astype(np.uint8)
        prec = [int(rng.choice([0, 2, 10, 60]))] * 3
        ref = oracle.sad_rows(l, r, d_max, win, prec)
        if rng.random() < 0.5:
            raw, mx = run_sad(engine, l, r, d_max, win, prec, raw_only=True)
        else:
            raw = np.zeros((h, w), np.uint8)
            cut = int(rng.integers(0, h + 1))
            mx = 0
            for rows in ((0, cut), (cut, h)):
                _, m = run_sad(engine, l, r, d_max, win, prec, raw_only=True, rows=rows, out=raw)
                mx = max(mx, m)
        np.testing.assert_array_equal(raw, ref, err_msg=str((case, w, h, d_max, win, prec)))
        assert mx == int(ref.max())


def test_sad_batch_equals_frame_by_frame(engine, oracle):
    """stepth_depth_from_additional_batch in SAD mode: all frames packed, searched and painted
    in one launch each -- whole frames, strips (planes hold only the strip's rows), host batch"""
    import torch
    from stepth_amd import MODE_SAD_WINDOW, synth
    pairs = [synth.stereo_pair(203, 97, 24, seed=60 + i)[:2] for i in range(5)]
    kw = dict(mode=MODE_SAD_WINDOW, d_max=24, win=5)
    want = [oracle.sad_rows(l, r, 24, 5, [4, 4, 4]) for l, r in pairs]
    dl = [torch.from_numpy(np.ascontiguousarray(l)).cuda() for l, _ in pairs]
    dr = [torch.from_numpy(np.ascontiguousarray(r)).cuda() for _, r in pairs]
    outs, mx = engine.depth_from_additional_batch(dl, dr, [4, 4, 4], raw_only=True, **kw)
    for f in range(5):
        assert np.array_equal(outs[f].cpu().numpy(), want[f]) and mx[f] == int(want[f].max()), f
    outs, mx = engine.depth_from_additional_batch(dl, dr, [4, 4, 4], **kw)  # normalised per frame
    for f in range(5):
        m = int(want[f].max())
        assert np.array_equal(outs[f].cpu().numpy(), (want[f].astype(np.uint32) * 255 // m).astype(np.uint8)), f
    for rows in ((0, 31), (31, 70), (70, 97), (96, 97)):
        outs, mx = engine.depth_from_additional_batch(dl, dr, [4, 4, 4], raw_only=True, rows=rows, **kw)
        for f in range(5):
            assert np.array_equal(outs[f].cpu().numpy()[rows[0]:rows[1]], want[f][rows[0]:rows[1]]), (f, rows)
            assert mx[f] == int(want[f][rows[0]:rows[1]].max())
    hl = [np.ascontiguousarray(l) for l, _ in pairs]
    hr = [np.ascontiguousarray(r) for _, r in pairs]
    outs, mx = engine.depth_from_additional_batch(hl, hr, [4, 4, 4], raw_only=True, win=9, mode=MODE_SAD_WINDOW, d_max=24)
    for f in range(5):
        assert np.array_equal(outs[f], oracle.sad_rows(hl[f], hr[f], 24, 9, [4, 4, 4])), f
    with pytest.raises(Exception):  # interleaved patterns are a ring-mode feature
        engine.depth_from_additional_batch(dl, dr, [4, 4, 4], row_pattern=(2, 2, 0), **kw)


@pytest.mark.parametrize("w,h,d_max,win", [(1920, 1080, 64, 5), (3840, 2160, 128, 5), (7680, 4320, 256, 9)])
def test_sad_full_size_band_against_oracle(engine, oracle, w, h, d_max, win):
    """the SAD mode at BASELINE's full sizes (configs 2, 3, 5): a 16-row band of the full-frame
    result against oracle.sad_rows (the whole frame would take the CPU minutes), the band as a
    strip call, and determinism of the full frame"""
    from stepth_amd import synth
    l0, r0, _ = synth.stereo_pair(960, 540, min(d_max, 128), seed=0x5AD)
    def tile(a):
        row = np.concatenate([a, a[:, ::-1]], axis=1)
        quad = np.concatenate([row, row[::-1]], axis=0)
        reps = (-(-h // quad.shape[0]), -(-w // quad.shape[1]), 1)
        return np.ascontiguousarray(np.tile(quad, reps)[:h, :w])
    l, r = tile(l0), tile(r0)
    prec = [4, 4, 4]
    raw, mx = run_sad(engine, l, r, d_max, win, prec, raw_only=True)
    raw2, _ = run_sad(engine, l, r, d_max, win, prec, raw_only=True)
    np.testing.assert_array_equal(raw, raw2)
    y0 = h // 2 - 8
    ref = oracle.sad_rows(l, r, d_max, win, prec, y0, y0 + 16)
    np.testing.assert_array_equal(raw[y0:y0 + 16], ref[y0:y0 + 16])
    band = np.zeros_like(raw)
    run_sad(engine, l, r, d_max, win, prec, raw_only=True, rows=(y0, y0 + 16), out=band)
    np.testing.assert_array_equal(band[y0:y0 + 16], raw[y0:y0 + 16])
    assert mx == int(raw.max()) and mx < d_max
