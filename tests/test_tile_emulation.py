"""CPU check of the "lane = query" searches (stepth_amd/csrc/tile_kernels.h): the dense one for
one group per pixel and the per-lane one for prepared queries (group lists, tiles, dot arrays).

tests/tools/tile_emulate.cpp compiles the kernel's own per-lane headers (tile_lane.h,
cell_pack.h) for the host and mirrors the wave-level control flow of ring_tile_kernel with
plain loops; its raw distance map must equal the oracle's (the restatement of
/root/reference/src/helpers.rs:9-54 + src/depth_image.rs:112-123) bit for bit.  The GPU suite
(test_gpu_ring_parity.py) checks the kernel itself; this one pins the algorithm without a GPU.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("tile_emu") / "tile_emulate.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-std=c++17",
                    "-I", os.path.join(ROOT, "stepth_amd", "csrc"),
                    os.path.join(ROOT, "tests", "tools", "tile_emulate.cpp"), "-o", out], check=True)
    lib = C.CDLL(out)
    lib.tile_emulate.restype = C.c_int
    lib.tile_emulate.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                 C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                 C.c_void_p, C.c_void_p]

    lib.cell_pack_selfcheck.restype = C.c_int
    lib.list_emulate.restype = C.c_int
    lib.list_emulate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                 C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p]

    def run(main, add, prec, max_steps, rows=None):
        main = np.ascontiguousarray(main)
        add = np.ascontiguousarray(add[..., :3])
        H, W, ch = main.shape
        r0, r1 = rows if rows else (0, H)
        raw = np.full((r1 - r0, W), 0xEE, np.uint8)
        mx = C.c_uint32(0)
        st = (C.c_uint64 * 6)()
        pr = (C.c_uint8 * 3)(*prec)
        rc = lib.tile_emulate(main.ctypes.data, W, H, ch, add.ctypes.data, add.shape[1], add.shape[0],
                              C.cast(pr, C.c_void_p), max_steps, r0, r1, raw.ctypes.data,
                              C.cast(C.byref(mx), C.c_void_p), C.cast(st, C.c_void_p))
        assert rc == 0, f"tile_emulate failed with {rc}"
        return raw, mx.value, list(st)
    run.lib = lib
    return run


def _check(emu, oracle, main, add, prec, ms, rows=None):
    """both variants of the dense search: every lane probing its own blocks (POOL = false) and
    the pooled probe tasks (POOL = true: kPoolFlushMin 16, kPoolCap 128)"""
    ref = oracle.depth_from_additional(main[..., :3], add, prec, ms, (1, 1))
    want = ref["raw"] if rows is None else ref["raw"][rows[0]:rows[1]]
    for flush_min in (0, 16):
        emu.lib.tile_emulate_policy(flush_min, 128)
        raw, mx, _ = emu(main, add, prec, ms, rows)
        bad = np.argwhere(raw != want)
        assert len(bad) == 0, f"policy {flush_min}: {len(bad)} pixels differ, first at {bad[0]}: {raw[tuple(bad[0])]} vs {want[tuple(bad[0])]}"
        if rows is None:
            assert mx == ref["max"]
    emu.lib.tile_emulate_policy(0, 128)


def test_asset_pair(emu, oracle, asset):
    _check(emu, oracle, asset["main"], asset["additional"], [36] * 3, 255)


@pytest.mark.parametrize("w,h,ms,prec,seed", [
    (200, 120, 255, 36, 1), (203, 117, 64, 36, 2), (96, 64, 255, 6, 3), (160, 96, 9, 36, 4),
    (320, 180, 128, 36, 5), (64, 40, 2, 36, 6), (130, 70, 26, 20, 7), (17, 9, 255, 36, 8),
])
def test_synthetic_pairs(emu, oracle, w, h, ms, prec, seed):
    from stepth_amd import synth
    l, r, _ = synth.stereo_pair(w, h, min(ms, 64), seed=seed)
    _check(emu, oracle, l, r, [prec] * 3, ms)


def test_long_reach_sparse_matches(emu, oracle):
    """few matching cells, far away: the superblock phase, rings up to 254, `d as u8` wraps"""
    rng = np.random.default_rng(11)
    main = np.full((300, 520, 3), 200, np.uint8)
    add = rng.integers(0, 60, (300, 520, 3), dtype=np.uint8)
    ys, xs = rng.integers(0, 300, 40), rng.integers(0, 520, 40)
    add[ys, xs] = 200
    _check(emu, oracle, main, add, [36] * 3, 255)
    _check(emu, oracle, main, add, [36] * 3, 100)


def test_unequal_sizes_rgba_and_strips(emu, oracle):
    rng = np.random.default_rng(5)
    main = rng.integers(0, 256, (90, 150, 4), dtype=np.uint8)
    main[..., :3] //= 3
    add = (rng.integers(0, 256, (37, 61, 3), dtype=np.uint8) // 3).astype(np.uint8)
    _check(emu, oracle, main, add, [30] * 3, 255)  # centres beyond the small array: None -> 0
    _check(emu, oracle, main, add, [30] * 3, 255, rows=(30, 61))
    _check(emu, oracle, main, add, [30, 0, 30], 255)  # precision 0 on a channel: nothing matches
    _check(emu, oracle, main, add, [255] * 3, 255)  # everything matches (empty borders must not)


def test_scan_order_ties(emu, oracle):
    """blocky images make many equal-ring candidates: side and position order decide"""
    rng = np.random.default_rng(9)
    coarse = rng.integers(0, 4, (12, 20, 3), dtype=np.uint8) * 70
    add = np.repeat(np.repeat(coarse, 9, axis=0), 9, axis=1)[:100, :170]
    main = np.repeat(np.repeat(rng.integers(0, 4, (24, 40, 3), dtype=np.uint8) * 70, 9, axis=0), 9, axis=1)[:200, :340]
    _check(emu, oracle, main, add, [36] * 3, 255)
    _check(emu, oracle, main, add, [36] * 3, 40)


def test_cell_format_arithmetic_is_the_reference_predicate(emu):
    """cell_pack.h: the box test as two additions (cell + lo, cell + hi, A & ~B & guard), its
    64-bit form for {max, min} bounds, the line test and the no-cell / empty values, against
    `|v - what| < precision` on every channel (/root/reference/src/helpers.rs:35): every value
    pair per channel for precisions 0..255, plus 200 000 random three-channel lines"""
    assert emu.lib.cell_pack_selfcheck() == 0


def _list_check(emu, oracle, add, cx, cy, what, prec, ms, lanes):
    """ring_list_kernel's emulation against oracle.match_groups on the same queries (a group
    {x = 2 cx, y = 2 cy, w = h = 0} has middle (cx, cy), src/depth_image.rs:114-117)"""
    add = np.ascontiguousarray(add[..., :3])
    cx, cy = np.ascontiguousarray(cx, np.uint32), np.ascontiguousarray(cy, np.uint32)
    what = np.ascontiguousarray(what, np.uint8)
    n = len(cx)
    d, found, pos = np.zeros(n, np.uint32), np.zeros(n, np.uint8), np.zeros((n, 2), np.uint32)
    st = (C.c_uint64 * 6)()
    pr = (C.c_uint8 * 3)(*prec)
    rc = emu.lib.list_emulate(cx.ctypes.data, cy.ctypes.data, what.ctypes.data, n, add.ctypes.data,
                              add.shape[1], add.shape[0], C.cast(pr, C.c_void_p), ms, lanes,
                              d.ctypes.data, found.ctypes.data, pos.ctypes.data, C.cast(st, C.c_void_p))
    assert rc == 0, f"list_emulate failed with {rc}"
    g = np.zeros(n, oracle.GROUP_DTYPE)
    g["x"], g["y"], g["value"] = 2 * cx, 2 * cy, what
    ref = oracle.match_groups(g, add, ms, prec)
    assert np.array_equal(found, ref["found"]), np.argwhere(found != ref["found"])[:5]
    assert np.array_equal(d, ref["raw32"]), np.argwhere(d != ref["raw32"])[:5]
    hit = found.astype(bool)
    assert np.array_equal(pos[hit], ref["pos"][hit])
    return list(st)


@pytest.mark.parametrize("lanes", [16, 32, 64])
def test_list_search_clustered_and_spread_queries(emu, oracle, lanes):
    """queries sorted by place share one window of superblocks per wave; shuffled ones are spread
    too far for its slots and walk their own windows -- both paths must give the oracle's answer"""
    from stepth_amd import synth
    rng = np.random.default_rng(100 + lanes)
    l, r, _ = synth.stereo_pair(700, 500, 60, seed=21)
    n = 1500
    cx, cy = rng.integers(0, 700, n), rng.integers(0, 500, n)
    what = l[rng.integers(0, 500, n), rng.integers(0, 700, n)]
    order = np.lexsort((cx // 32, cy // 32))  # bins of 32x32 cells, as the engine sorts group lists
    st = _list_check(emu, oracle, r, cx[order], cy[order], what[order], [36] * 3, 128, lanes)
    st2 = _list_check(emu, oracle, r, cx, cy, what, [36] * 3, 255, lanes)
    assert st[0] > 0 and st2[1] > 0, (st, st2)  # the shared and the per-lane far phase both ran


def test_list_search_edges(emu, oracle):
    """centres outside the array end at once (src/helpers.rs:49-51), rare matches far away,
    precision 0, reaches of 1 and 2, a one-row array"""
    rng = np.random.default_rng(5)
    add = np.full((300, 420, 3), 90, np.uint8)
    add[17:19, 400:403] = 200
    add[290:293, 3:9] = 20
    n = 400
    cx, cy = rng.integers(0, 460, n), rng.integers(0, 330, n)  # some beyond 420 x 300
    what = rng.choice([200, 20, 90, 150], (n, 1)).repeat(3, 1).astype(np.uint8)
    for ms in (1, 2, 25, 26, 255, 263):
        _list_check(emu, oracle, add, cx, cy, what, [36] * 3, ms, 64)
    _list_check(emu, oracle, add, cx, cy, what, [0, 36, 36], 255, 32)
    _list_check(emu, oracle, add, cx, cy, what, [255] * 3, 255, 16)
    row = rng.integers(0, 256, (1, 500, 3)).astype(np.uint8) // 64 * 64
    _list_check(emu, oracle, row, rng.integers(0, 500, 200), np.zeros(200, int),
                row[0, rng.integers(0, 500, 200)], [20] * 3, 200, 64)
