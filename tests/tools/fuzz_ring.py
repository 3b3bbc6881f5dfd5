"""one-off differential fuzz of the ring path against the oracle: larger images than the suite's
randomised test (far phase, superblock steps, shells), many seeds"""
import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np
from stepth_amd import Engine, MaxZeroError, synth
from oracle import oracle as O
eng = Engine(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
t0 = time.time(); n = 0; bad = 0
while time.time() - t0 < budget:
    n += 1
    kind = rng.integers(0, 4)
    w, h = int(rng.integers(40, 420)), int(rng.integers(40, 300))
    if kind == 0:
        l, r, _ = synth.stereo_pair(w, h, int(rng.integers(4, 100)), seed=int(rng.integers(0, 1 << 30)))
    elif kind == 1:   # blocky noise: coherent regions, far hits and none
        cell = int(rng.choice([4, 8, 16, 32, 64]))
        lv = rng.integers(0, 256, ((h + cell - 1) // cell, (w + cell - 1) // cell, 3), dtype=np.uint8) // 48 * 48
        l = np.repeat(np.repeat(lv, cell, 0), cell, 1)[:h, :w].copy()
        r = np.roll(l, (int(rng.integers(-40, 40)), int(rng.integers(-90, 90))), (0, 1)).copy()
        r = (r.astype(int) + rng.integers(-6, 7, r.shape)).clip(0, 255).astype(np.uint8)
    elif kind == 2:   # unequal sizes
        l = rng.integers(0, 256, (h, w, 3), dtype=np.uint8) // 32 * 32
        r = rng.integers(0, 256, (int(rng.integers(20, 300)), int(rng.integers(20, 400)), 3), dtype=np.uint8) // 32 * 32
    else:             # sparse matches: mostly mismatching colours with a few islands
        l = np.full((h, w, 3), 200, np.uint8); l[::7, ::5] = 20
        r = np.full((h, w, 3), 90, np.uint8)
        for _ in range(int(rng.integers(0, 12))):
            y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
            r[y:y + int(rng.integers(1, 9)), x:x + int(rng.integers(1, 9))] = rng.choice([200, 20])
    prec = [int(rng.integers(1, 60))] * 3 if rng.random() < 0.7 else [int(v) for v in rng.integers(0, 256, 3)]
    ms = int(rng.choice([5, 24, 25, 26, 40, 64, 100, 128, 200, 255, 259, 300]))
    ref = O.depth_from_additional(l, r, prec, ms, (1, 1))
    try:
        mode = rng.integers(0, 3)
        hf = bool(rng.integers(0, 2))  # STEPTH_FLAG_HEAVY_FIRST: the pooled-probe variant of the dense search
        if mode == 0:
            d, mx = eng.depth_from_additional(l, r, prec, max_steps=ms, heavy_first=hf)
            ok = ref["status"] == 0 and mx == ref["max"] and np.array_equal(d, ref["depth"])
        elif mode == 1:
            outs, mxs = eng.depth_from_additional_batch([l, l[::-1].copy()], [r, r[::-1].copy()], prec, max_steps=ms, raw_only=True, heavy_first=hf)
            ref2 = O.depth_from_additional(l[::-1].copy(), r[::-1].copy(), prec, ms, (1, 1))
            ok = np.array_equal(outs[0], ref["raw"]) and np.array_equal(outs[1], ref2["raw"]) and mxs == [ref["max"], ref2["max"]]
        else:
            blk, mod = 2 * int(rng.integers(1, 9)), int(rng.integers(1, 5))
            out = np.zeros(l.shape[:2], np.uint8); mm = []
            for rem in range(mod):
                _, mx = eng.depth_from_additional(l, r, prec, max_steps=ms, raw_only=True, out=out, row_pattern=(blk, mod, rem), heavy_first=hf); mm.append(mx)
            ok = np.array_equal(out, ref["raw"]) and max(mm) == ref["max"]
    except MaxZeroError:
        ok = ref["status"] == O.ERR_MAX_ZERO
    if not ok:
        bad += 1
        print("MISMATCH case", n, kind, w, h, prec, ms, flush=True)
        np.savez(f"gpurun_out/fuzz_bad_{n}.npz", l=l, r=r, prec=prec, ms=ms)
print(f"fuzz: {n} cases, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
