"""N>1 path on CPU: world_size-2 and -3 `gloo` process groups run the row-strip flow of
stepth_amd/sharding.py (all_reduce(MAX) + all_gather) with the oracle standing in for the
per-strip compute, and every rank must end up with the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


class OracleStripWorker:
    """test double for EngineStripWorker: same contract, CPU oracle inside"""

    def __init__(self, prec, max_steps, tile):
        from oracle import oracle as O
        self.O, self.prec, self.max_steps, self.tile = O, prec, max_steps, tile

    def match_strip(self, frame, rows, max_out=None, pattern=None):
        main, add = frame
        O = self.O
        h, w = main.shape[:2]
        g = O.uniform_groups(main, *self.tile)
        keep = (g["y"] >= rows[0]) & (g["y"] < rows[1])
        if pattern:
            keep &= (g["y"] // pattern[0]) % pattern[1] == pattern[2]
        sel = g[keep]
        out = np.zeros((h, w), np.uint8)
        mx = 0
        if len(sel):
            m = O.match_groups(sel, add, self.max_steps, self.prec, threads=O.SERIAL)
            out = O.paint(sel, m["raw"], w, h)
            mx = int(m["raw"].max())
        return torch.from_numpy(out[rows[0]:rows[1]].copy()), torch.tensor([mx], dtype=torch.int32)

    def normalise(self, rows, max_word, first_row=0, pattern=None):
        mx = int(max_word.item())
        if rows.numel():
            v = rows.to(torch.int32) * 255 // max(mx, 1) if mx else rows.to(torch.int32) * 0
            if pattern:
                ys = torch.arange(first_row, first_row + rows.shape[0])
                own = (ys // pattern[0]) % pattern[1] == pattern[2]
                rows[own] = v.to(torch.uint8)[own]
            else:
                rows.copy_(v.to(torch.uint8))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tile, q, dst=None, interleave=0):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stepth_amd import synth
        from stepth_amd.sharding import sharded_depth
        frames = []
        for seed in (1, 2) if dst != "frames" else (1, 2, 3, 4):
            l, r, _ = synth.stereo_pair(97, 61, 16, seed=seed)
            frames.append((l, r))
        wk = OracleStripWorker([36] * 3, 64, tile)
        depth, mx = sharded_depth(wk, frames, 61, 97, tile_h=tile[1], dst=dst,
                                  interleave=interleave)
        if dst != "frames":
            assert (depth is None) == (dst is not None and rank != dst)
        q.put((rank, None if depth is None else depth.numpy(), mx.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tile,dst,interleave", [(2, (1, 1), None, 0), (3, (1, 1), 0, 0),
                                                       (2, (5, 5), 1, 0), (3, (1, 1), 0, 8),
                                                       (2, (2, 2), None, 4)])
def test_sharded_equals_single_process(world, tile, dst, interleave):
    from oracle import oracle as O
    from stepth_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tile, q, dst, interleave))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for seed_i, seed in enumerate((1, 2)):
        l, r, _ = synth.stereo_pair(97, 61, 16, seed=seed)
        ref = O.depth_from_additional(l, r, [36] * 3, 64, tile)
        for rank, depth, mx in results:
            if depth is not None:
                np.testing.assert_array_equal(depth[seed_i], ref["depth"])
            assert int(mx[seed_i]) == ref["max"]
    assert sum(d is not None for _r, d, _m in results) == (world if dst is None else 1)


def test_single_process_path_without_init():
    from oracle import oracle as O
    from stepth_amd import synth
    from stepth_amd.sharding import sharded_depth
    l, r, _ = synth.stereo_pair(50, 40, 16, seed=4)
    wk = OracleStripWorker([36] * 3, 64, (1, 1))
    depth, mx = sharded_depth(wk, [(l, r)], 40, 50)
    ref = O.depth_from_additional(l, r, [36] * 3, 64)
    np.testing.assert_array_equal(depth[0].numpy(), ref["depth"])


# ---- select_foreground on a row-sharded depth map: all_reduce(SUM) of 256 counts ---------
class NumpyZoneWorker:
    """test double for EngineZoneWorker: numpy histogram / histogram k-means / threshold.
    The k-means is the histogram formulation the GPU uses (tests/test_oracle_zones.py checks it
    against the oracle's per-item loop); the end result is compared with the oracle."""

    def histogram(self, rows):
        return torch.from_numpy(np.bincount(rows.numpy().ravel(), minlength=256).astype(np.int64))

    def zones(self, counts, n_zones):
        from test_oracle_zones import _zones_from_counts
        status, zones = _zones_from_counts(counts.numpy(), n_zones)
        assert status == 0
        return zones

    def slice(self, rows, lo, hi):
        return torch.where((rows >= lo) & (rows <= hi), 255, 0).to(torch.uint8)


def _zone_rank(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stepth_amd.sharding import sharded_select_foreground, strip_bounds
        depth = _zone_depth_map()
        r0, r1 = strip_bounds(depth.shape[0], world)[rank]
        mask, zone = sharded_select_foreground(NumpyZoneWorker(), torch.from_numpy(depth[r0:r1].copy()))
        q.put((rank, r0, r1, mask.numpy(), zone))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _zone_depth_map():
    rng = np.random.default_rng(21)
    yy, xx = np.mgrid[0:5, 0:83]  # 5 rows over 3 ranks: one rank gets a single row
    d = np.where(xx < 40, 20 + rng.integers(0, 30, (5, 83)), 200 + rng.integers(0, 40, (5, 83)))
    d[0, :] = 3  # the global minimum lives on rank 0 only
    return d.astype(np.uint8)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_select_foreground_equals_single_process(world):
    from oracle import oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_zone_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    depth = _zone_depth_map()
    ref = O.select_foreground(depth)
    assert ref["status"] == 0
    for rank, r0, r1, mask, zone in results:
        assert tuple(zone) == ref["zone"]
        np.testing.assert_array_equal(mask, ref["mask"][r0:r1])


# ---- work-weighted strips: cut points move between steps, results must not -----------------
def test_strip_balancer_moves_cuts_towards_equal_time():
    from stepth_amd.sharding import StripBalancer
    b = StripBalancer(1080, 4, align=2, damping=1.0)
    assert b.bounds == [(0, 270), (270, 540), (540, 810), (810, 1080)]
    b.timed_bounds = list(b.bounds)
    b.update([100, 100, 300, 100])          # rank 2's rows cost three times as much
    # total 600 -> 150 each: cut 1 at 270 + 50/100*270 = 405, cut 2 inside strip 2 at
    # 540 + (300 - 200)/300 * 270 = 630, cut 3 at 540 + 250/300 * 270 = 765
    cuts = [r0 for r0, _r1 in b.bounds[1:]]
    assert all(abs(c - w) <= 2 for c, w in zip(cuts, (405, 630, 765))) and all(c % 2 == 0 for c in cuts)
    assert b.bounds[0][0] == 0 and b.bounds[-1][1] == 1080
    assert all(b.bounds[k][1] == b.bounds[k + 1][0] for k in range(3))
    before = list(b.bounds)
    b.update([0, 1, 1, 1])                   # a missing time: leave the cuts alone
    assert b.bounds == before
    b.timed_bounds = list(b.bounds)
    b.update([1, 1, 1, 10 ** 6])             # never squeeze a strip below min_rows
    assert all(r1 - r0 >= 8 for r0, r1 in b.bounds) and b.bounds[-1][1] == 1080


def _balanced_rank(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        from stepth_amd import synth
        from stepth_amd.sharding import StripBalancer, sharded_depth
        l, r, _ = synth.stereo_pair(97, 61, 16, seed=5)

        class Slow(OracleStripWorker):  # rank 1 pretends to be three times slower
            def match_strip(self, frame, rows, max_out=None, pattern=None):
                out = super().match_strip(frame, rows, max_out, pattern)
                time.sleep(0.03 * (3 if rank == 1 else 1) * max(rows[1] - rows[0], 1) / 30)
                return out

        wk = Slow([36] * 3, 64, (1, 1))
        bal = StripBalancer(61, world, align=1, min_rows=4)
        history, depths = [], []
        for _step in range(5):
            history.append(list(bal.bounds))
            depth, mx = sharded_depth(wk, [(l, r)], 61, 97, balancer=bal)
            depths.append(depth.numpy().copy())
        q.put((rank, history, depths, mx.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_balanced_strips_give_the_same_depth_and_shift_work():
    from oracle import oracle as O
    from stepth_amd import synth
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_balanced_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    l, r, _ = synth.stereo_pair(97, 61, 16, seed=5)
    ref = O.depth_from_additional(l, r, [36] * 3, 64)
    hist0 = results[0][1]
    for rank, history, depths, mx in results:
        assert history == hist0                      # every rank derived the same cut points
        for d in depths:
            np.testing.assert_array_equal(d[0], ref["depth"])
        assert int(mx[0]) == ref["max"]
    first, last = hist0[0], hist0[-1]
    assert first != last                             # the cuts moved ...
    assert (last[1][1] - last[1][0]) < (first[1][1] - first[1][0])  # ... away from the slow rank


# ---- pipelined gather: results lag one call, flush() delivers the last ----------------------
def _pipelined_rank(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stepth_amd import synth
        from stepth_amd.sharding import GatherPipeline, StripBalancer, sharded_depth
        wk = OracleStripWorker([36] * 3, 64, (1, 1))
        pipe, bal = GatherPipeline(), StripBalancer(61, world, align=1, min_rows=4)
        got = []
        for seed in (1, 2, 3):
            l, r, _ = synth.stereo_pair(97, 61, 16, seed=seed)
            depth, mx = sharded_depth(wk, [(l, r)], 61, 97, dst=0, balancer=bal, pipeline=pipe)
            got.append((None if depth is None else depth.numpy().copy(), mx.numpy().copy()))
        last = pipe.flush()
        q.put((rank, got, None if last is None else last.numpy().copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_lags_one_call_and_flushes():
    from oracle import oracle as O
    from stepth_amd import synth
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipelined_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    refs = [O.depth_from_additional(*synth.stereo_pair(97, 61, 16, seed=s)[:2], [36] * 3, 64) for s in (1, 2, 3)]
    for rank, got, last in results:
        for i, (depth, mx) in enumerate(got):
            assert int(mx[0]) == refs[i]["max"]          # the maxima are this call's
            if rank == 0 and i > 0:
                np.testing.assert_array_equal(depth[0], refs[i - 1]["depth"])  # maps lag one call
            else:
                assert depth is None
        if rank == 0:
            np.testing.assert_array_equal(last[0], refs[2]["depth"])
        else:
            assert last is None


@pytest.mark.parametrize("world,interleave", [(3, 0), (2, 4)])
def test_every_rank_is_a_sink_frame_f_gathers_to_rank_f_mod_n(world, interleave):
    """dst="frames": no rank receives more than its share of the depth maps"""
    from oracle import oracle as O
    from stepth_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, (1, 1), q, "frames", interleave))
             for r in range(world)]
    for p in procs:
        p.start()
    results = {rank: (depth, mx) for rank, depth, mx in [q.get(timeout=180) for _ in range(world)]}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    refs = [O.depth_from_additional(*synth.stereo_pair(97, 61, 16, seed=s)[:2], [36] * 3, 64) for s in (1, 2, 3, 4)]
    for rank in range(world):
        depth, mx = results[rank]
        mine = list(range(rank, 4, world))
        assert depth.shape[0] == len(mine)
        for k, f in enumerate(mine):
            np.testing.assert_array_equal(depth[k], refs[f]["depth"])
        assert [int(v) for v in mx] == [r["max"] for r in refs]  # every rank knows every max


def _moving_slow_rank(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        from stepth_amd import synth
        from stepth_amd.sharding import StripBalancer, sharded_depth
        pairs = [synth.stereo_pair(97, 61, 16, seed=11 + k)[:2] for k in range(3)]
        step_no = [0]

        class Moving(OracleStripWorker):  # the slow rank is a different one every step
            def match_strip(self, frame, rows, max_out=None, pattern=None):
                out = super().match_strip(frame, rows, max_out, pattern)
                slow = step_no[0] % world
                time.sleep(0.02 * (4 if rank == slow else 1) * max(rows[1] - rows[0], 1) / 30)
                return out

        wk = Moving([36] * 3, 64, (1, 1))
        bal = StripBalancer(61, world, align=1, min_rows=4)
        history, depths = [], []
        for step in range(7):  # content changes every step too: three frame sets cycled
            step_no[0] = step
            history.append(list(bal.bounds))
            depth, mx = sharded_depth(wk, [pairs[step % 3]], 61, 97, balancer=bal)
            depths.append(depth.numpy().copy())
        q.put((rank, history, depths))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_balancer_with_a_slow_rank_that_moves_every_step():
    """frames AND the slow rank change from step to step (the balancer chases a moving target):
    every rank still derives the same cut points, the cuts stay valid partitions of the rows,
    and every step's depth map is the oracle's"""
    from oracle import oracle as O
    from stepth_amd import synth
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_moving_slow_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    refs = [O.depth_from_additional(*synth.stereo_pair(97, 61, 16, seed=11 + k)[:2], [36] * 3, 64) for k in range(3)]
    hist0 = results[0][1]
    assert len({tuple(map(tuple, h)) for h in hist0}) > 1  # the cuts did move
    for rank, history, depths in results:
        assert history == hist0
        for step, d in enumerate(depths):
            np.testing.assert_array_equal(d[0], refs[step % 3]["depth"])
    for bounds in hist0:
        assert bounds[0][0] == 0 and bounds[-1][1] == 61
        assert all(bounds[k][1] == bounds[k + 1][0] and bounds[k][1] - bounds[k][0] >= 4 for k in range(world - 1))


def test_balancer_cut_points_keep_their_phase():
    """StripBalancer(align=8, phase=7): cut points at pixel rows 8k - 1, the row pattern of the
    dense search's waves (stepth_amd/csrc/tile_kernels.h), initially and after every update,
    strips never below min_rows, the same on every rank (pure function of the all-reduced times)"""
    from stepth_amd.sharding import StripBalancer
    rng = np.random.default_rng(7)
    for world in (2, 3, 4, 8):
        bals = [StripBalancer(1080, world, align=8, phase=7) for _ in range(2)]
        for it in range(40):
            times = [int(t) for t in rng.integers(200, 3000, world)]
            if it % 5 == 4:  # one rank far slower than the rest: its neighbours' strips hit min_rows
                times[int(rng.integers(0, world))] *= 400
            for b in bals:
                b.update(times, b.bounds)
            assert bals[0].bounds == bals[1].bounds
            cuts = [a for a, _ in bals[0].bounds[1:]]
            assert all(c % 8 == 7 for c in cuts), cuts
            assert bals[0].bounds[0][0] == 0 and bals[0].bounds[-1][1] == 1080
            assert all(b - a >= 8 for a, b in bals[0].bounds)
            assert all(bals[0].bounds[k][1] == bals[0].bounds[k + 1][0] for k in range(world - 1))


def _frame_split_rank(rank, world, port, q, dst):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stepth_amd import synth
        from stepth_amd.sharding import frame_sharded_depth
        frames = [synth.stereo_pair(97, 61, 16, seed=s)[:2] for s in (1, 2, 3, 4, 5)]
        wk = OracleStripWorker([36] * 3, 64, (1, 1))
        depth, mx = frame_sharded_depth(wk, frames, 61, 97, dst=dst)
        q.put((rank, None if depth is None else depth.numpy(), mx.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dst", [(2, None), (3, None), (3, 1)])
def test_frame_split_whole_frames_per_rank(world, dst):
    """frame_sharded_depth: frame f wholly on rank f % world (BASELINE config 4's independent
    pairs): no halo, no cross-rank max; every rank ends up with the oracle's maps of its own
    frames, or one rank with all of them"""
    from oracle import oracle as O
    from stepth_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_frame_split_rank, args=(r, world, port, q, dst)) for r in range(world)]
    for p in procs:
        p.start()
    results = {rank: (depth, mx) for rank, depth, mx in [q.get(timeout=180) for _ in range(world)]}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    refs = [O.depth_from_additional(*synth.stereo_pair(97, 61, 16, seed=s)[:2], [36] * 3, 64) for s in (1, 2, 3, 4, 5)]
    for rank in range(world):
        depth, mx = results[rank]
        mine = list(range(rank, 5, world))
        if dst is None:
            assert depth.shape[0] == len(mine)
            for k, f in enumerate(mine):
                np.testing.assert_array_equal(depth[k], refs[f]["depth"])
            assert [int(v) for v in mx] == [refs[f]["max"] for f in mine]
        elif rank == dst:
            assert depth.shape[0] == 5
            for f in range(5):
                np.testing.assert_array_equal(depth[f], refs[f]["depth"])
            assert [int(v) for v in mx] == [r["max"] for r in refs]
        else:
            assert depth is None


def _pipelined_frames_rank(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stepth_amd import synth
        from stepth_amd.sharding import GatherPipeline, sharded_depth
        sets = [[synth.stereo_pair(97, 61, 16, seed=10 * k + s)[:2] for s in (1, 2, 3)] for k in range(3)]
        wk = OracleStripWorker([36] * 3, 64, (1, 1))
        pipe = GatherPipeline()
        got = []
        for frames in sets:
            depth, mx = sharded_depth(wk, frames, 61, 97, dst="frames", pipeline=pipe)
            got.append((None if depth is None else depth.numpy().copy(), mx.numpy().copy()))
        last = pipe.flush()
        q.put((rank, got, last.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_with_every_rank_a_sink():
    """dst="frames" + GatherPipeline: the gathers of a step run behind the next step's PASS 1;
    a call hands out the rank's own frames of the PREVIOUS call, flush() the last ones"""
    from oracle import oracle as O
    from stepth_amd import synth
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipelined_frames_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    refs = [[O.depth_from_additional(*synth.stereo_pair(97, 61, 16, seed=10 * k + s)[:2], [36] * 3, 64) for s in (1, 2, 3)]
            for k in range(3)]
    for rank, got, last in results:
        mine = list(range(rank, 3, world))
        for k, (depth, mx) in enumerate(got):
            assert [int(v) for v in mx] == [r["max"] for r in refs[k]]  # the maxima are this call's
            if k == 0:
                assert depth is None
            else:
                for j, f in enumerate(mine):
                    np.testing.assert_array_equal(depth[j], refs[k - 1][f]["depth"])  # maps lag one call
        for j, f in enumerate(mine):
            np.testing.assert_array_equal(last[j], refs[2][f]["depth"])
