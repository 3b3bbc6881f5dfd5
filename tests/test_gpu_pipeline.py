"""File-level pipeline: decode threads + batched engine calls; results equal
the oracle on the decoded bytes, in order, for PNG and JPEG files."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_depth_from_files_matches_oracle(engine, oracle, tmp_path):
    from PIL import Image
    from stepth_amd import InvalidInput, synth
    from stepth_amd.pipeline import depth_from_files
    pairs, want = [], []
    for i in range(11):
        l, r, _ = synth.stereo_pair(160, 96, 24, seed=500 + i)
        ext = "png" if i % 2 == 0 else "jpg"
        pm, pa = tmp_path / f"m{i}.{ext}", tmp_path / f"a{i}.{ext}"
        Image.fromarray(l).save(pm, quality=92)
        Image.fromarray(r).save(pa, quality=92)
        pairs.append((str(pm), str(pa)))
        dl = np.asarray(Image.open(pm).convert("RGB"))  # what the pipeline decodes
        dr = np.asarray(Image.open(pa).convert("RGB"))
        ref = oracle.depth_from_additional(dl, dr, [36] * 3, 255)
        want.append((oracle.gaussian_same_size(ref["depth"]), ref["max"]))
    got = list(depth_from_files(pairs, [36] * 3, engine=engine, chunk=4, workers=3))
    assert len(got) == len(want)
    for i, ((d, mx), (wd, wmx)) in enumerate(zip(got, want)):
        assert mx == wmx and np.array_equal(d, wd), i
    bare = list(depth_from_files(pairs[:3], [36] * 3, engine=engine, chunk=8, gaussian_tail=False))
    for i in range(3):
        dl = np.asarray(Image.open(pairs[i][0]).convert("RGB"))
        dr = np.asarray(Image.open(pairs[i][1]).convert("RGB"))
        assert np.array_equal(bare[i][0], oracle.depth_from_additional(dl, dr, [36] * 3, 255)["depth"])
    with pytest.raises(InvalidInput):
        list(depth_from_files([(str(tmp_path / "missing.png"), pairs[0][1])], [36] * 3, engine=engine))
    assert list(depth_from_files([], [36] * 3, engine=engine)) == []
