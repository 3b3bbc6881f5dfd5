"""one-off differential fuzz of the prepared-query paths (explicit group lists in any order,
uniform tiles, single calls and batches, strips) against the oracle: random rectangles with
random values on synthetic / blocky / sparse pairs, all reaches incl. those beyond the dense search (> 263)"""
import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np
import torch
from stepth_amd import Engine, MaxZeroError, synth
from oracle import oracle as O
eng = Engine(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
t0 = time.time(); n = 0; bad = 0
while time.time() - t0 < budget:
    n += 1
    w, h = int(rng.integers(30, 380)), int(rng.integers(30, 260))
    kind = rng.integers(0, 3)
    if kind == 0:
        l, r, _ = synth.stereo_pair(w, h, int(rng.integers(4, 100)), seed=int(rng.integers(0, 1 << 30)))
    elif kind == 1:
        cell = int(rng.choice([4, 8, 16, 32, 64]))
        lv = rng.integers(0, 256, ((h + cell - 1) // cell, (w + cell - 1) // cell, 3), dtype=np.uint8) // 48 * 48
        l = np.repeat(np.repeat(lv, cell, 0), cell, 1)[:h, :w].copy()
        r = np.roll(l, (int(rng.integers(-40, 40)), int(rng.integers(-90, 90))), (0, 1)).copy()
    else:
        l = np.full((h, w, 3), 200, np.uint8); l[::7, ::5] = 20
        r = np.full((h, w, 3), 90, np.uint8)
        for _ in range(int(rng.integers(0, 12))):
            y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
            r[y:y + int(rng.integers(1, 9)), x:x + int(rng.integers(1, 9))] = rng.choice([200, 20])
    prec = [int(rng.integers(1, 60))] * 3
    ms = int(rng.choice([5, 25, 26, 64, 128, 255, 263, 264, 300]))
    mode = rng.integers(0, 4)
    try:
        if mode == 0:  # adaptive groups in shuffled order, single call
            g = O.adaptive_groups(l, prec, int(rng.integers(4, 14)), 0)
            g = g[rng.permutation(len(g))]
            ref = O.depth_from_additional(l, r, prec, ms, groups=g)
            d, mx = eng.depth_from_additional(l, r, prec, max_steps=ms, groups=g)
            ok = ref["status"] == 0 and mx == ref["max"] and np.array_equal(d, ref["depth"])
        elif mode == 1:  # random rectangles (overlaps allowed: compare raw distances per group)
            k = int(rng.integers(1, 400))
            g = np.zeros(k, O.GROUP_DTYPE)
            g["x"], g["y"] = rng.integers(0, w, k), rng.integers(0, h, k)
            g["w"], g["h"] = rng.integers(1, 40, k), rng.integers(1, 40, k)
            g["value"] = rng.integers(0, 256, (k, 3))
            m = O.match_groups(g, r, ms, prec)
            what = np.ascontiguousarray(g["value"])
            cen = np.stack([(g["x"] + g["w"]) // 2, (g["y"] + g["h"]) // 2], 1).astype(np.uint32)
            fnd, dist, pos = eng.distance_dot_array_batch(r, what, cen, ms, prec)
            ok = np.array_equal(fnd.astype(bool), m["found"].astype(bool)) and np.array_equal(dist, m["raw32"]) and \
                np.array_equal(pos[m["found"].astype(bool)], m["pos"][m["found"].astype(bool)])
        elif mode == 2:  # tiles, batched, a strip
            tile = (int(rng.integers(2, 20)), int(rng.integers(2, 20)))
            mains = [torch.from_numpy(l).cuda(), torch.from_numpy(l[::-1].copy()).cuda()]
            adds = [torch.from_numpy(r).cuda(), torch.from_numpy(r[::-1].copy()).cuda()]
            r0 = (int(rng.integers(0, h)) // tile[1]) * tile[1]
            outs, mxs = eng.depth_from_additional_batch(mains, adds, prec, max_steps=ms, tile=tile, raw_only=True,
                                                        rows=(r0, h))
            ok = True
            for f, (ll, rr) in enumerate([(l, r), (l[::-1].copy(), r[::-1].copy())]):
                ref = O.depth_from_additional(ll, rr, prec, ms, tile)
                ok &= np.array_equal(outs[f].cpu().numpy()[r0:], ref["raw"][r0:])
        else:  # batched group lists on the device
            mains = [torch.from_numpy(l).cuda(), torch.from_numpy(l[:, ::-1].copy()).cuda()]
            adds = [torch.from_numpy(r).cuda(), torch.from_numpy(r[:, ::-1].copy()).cuda()]
            gs = [O.adaptive_groups(m.cpu().numpy(), prec, 10, 0) for m in mains]
            gs = [g[rng.permutation(len(g))] for g in gs]
            gd = [torch.from_numpy(np.ascontiguousarray(g).view(np.uint8).reshape(-1, 20)).cuda() for g in gs]
            outs, mxs = eng.depth_from_groups_batch(mains, adds, prec, gd, [len(g) for g in gs], max_steps=ms, raw_only=True)
            ok = True
            for f in range(2):
                m = O.match_groups(gs[f], adds[f].cpu().numpy(), ms, prec)
                ok &= np.array_equal(outs[f].cpu().numpy(), O.paint(gs[f], m["raw"], w, h)) and mxs[f] == int(m["raw"].max())
    except MaxZeroError:
        ok = ref["status"] == O.ERR_MAX_ZERO
    if not ok:
        bad += 1
        print("MISMATCH case", n, mode, kind, w, h, prec, ms, flush=True)
print(f"fuzz lists: {n} cases, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
