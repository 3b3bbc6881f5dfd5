"""Row-strip sharding of one frame across the GPUs of a node (one process per GPU).

The reference is single-process (rayon threads, src/depth_image.rs:111-112); groups are
independent in PASS 1 and coupled only through the global max (src/depth_image.rs:124).  So
a frame shards by main-image row strips with exactly one real exchange:

    PASS 1 on the rank's strip  ->  all_reduce(MAX) of one word per frame
                                ->  PASS 2 on the strip (stepth_normalise_rows)
                                ->  gather (or all_gather) of the u8 depth rows

`torch.distributed` is plumbing (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  The two compute steps are injected as callables so that the collective
logic is testable without a GPU; `EngineStripWorker` is the product implementation on top of
the HIP engine.  The additional image is replicated on every rank: a strip's search window
lives in half-coordinate space ((y+1)/2 - 254 ... ) and is not a thin halo.
"""
from __future__ import annotations

import collections
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def strip_bounds(height: int, world: int, tile_h: int = 1) -> List[Tuple[int, int]]:
    """Contiguous, tile-aligned, near-equal row strips; later ranks may get empty strips."""
    if height <= 0 or world <= 0 or tile_h <= 0:
        raise ValueError("height, world and tile_h must be positive")
    tiles = (height + tile_h - 1) // tile_h
    out, t0 = [], 0
    for r in range(world):
        t1 = t0 + tiles // world + (1 if r < tiles % world else 0)
        out.append((min(t0 * tile_h, height), min(t1 * tile_h, height)))
        t0 = t1
    return out


class StripBalancer:
    """Work-weighted contiguous row strips from measured per-rank times.

    The cost of a row is data dependent (one contiguous 1/8 strip of the bench frame holds
    27 % of its work), so equal strips leave most ranks waiting for the slowest.  Frames of a
    stream resemble their predecessors: every step each rank times its own share, the times
    travel with the all_reduce of the maxima (no extra collective), and all ranks derive the
    same new cut points -- equal shares of the work under a piecewise-constant cost per row.
    Results do not depend on where the cuts are."""

    def __init__(self, height: int, world: int, align: int = 2, min_rows: int = 8,
                 damping: float = 0.7, lag: int = 1, phase: int = 0):
        """`lag` = calls in flight at once (steps alternating over that many streams): the
        measurement read at call k is the one of call k - lag, so reading it never waits for
        work that is meant to overlap with call k.  Cut points are `align * k + phase`: the dense
        search gives a wave the pixel rows 8k-1 .. 8k+6 (two rows per search centre, the first
        centre row holds row 0 alone), so with align 8, phase 7 no wave straddles two ranks."""
        self.lag = max(1, lag)
        self.height, self.world, self.align = height, world, max(1, align)
        self.phase = phase % self.align
        self.min_rows, self.damping = max(min_rows, self.align), damping
        self.bounds = strip_bounds(height, world, self.align)
        if self.phase:
            cuts = [0] + [self._snap(a) for a, _ in strip_bounds(height, world)[1:]] + [height]
            if all(cuts[k] + self.min_rows <= cuts[k + 1] for k in range(world)):
                self.bounds = [(cuts[k], cuts[k + 1]) for k in range(world)]
        self.last_ms: Optional[float] = None  # this rank's most recent finished measurement
        self.timed_bounds = None   # ... and the strips it was taken with
        self._open = None          # measurement being taken
        self._timings = collections.deque()  # GPU measurements not read yet
        self._pending = collections.deque()  # (all-reduced times, their bounds, event) per call
        self._updates = 0

    def _snap(self, y: float) -> int:
        return int(round((y - self.phase) / self.align)) * self.align + self.phase

    # -- timing of this rank's share (GPU events on the current stream, or the wall clock) --
    def start(self, on_gpu: bool):
        if on_gpu:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            self._open = (ev, list(self.bounds))
        else:
            import time
            self._open = (time.perf_counter(), list(self.bounds))

    def stop(self):
        if self._open is None:
            return
        mark, bounds = self._open
        self._open = None
        if isinstance(mark, tuple):
            mark[1].record()
            self._timings.append((mark, bounds))  # read later, once the GPU has got there
        else:
            import time
            self.last_ms, self.timed_bounds = (time.perf_counter() - mark) * 1e3, bounds

    def previous_us(self) -> int:
        """the previous step's measurement in microseconds (0 = none yet) and, in
        `timed_bounds`, the strips it was taken with.  Every rank reads the SAME step's
        measurement (the cut points must come out identical everywhere), so a GPU measurement
        still in flight is waited for -- it ends before that step's normalise and gather, which
        are what keeps the GPU busy meanwhile."""
        if len(self._timings) >= self.lag:
            (a, b), bounds = self._timings.popleft()
            b.synchronize()
            self.last_ms, self.timed_bounds = a.elapsed_time(b), bounds
        return 0 if self.last_ms is None else max(1, int(self.last_ms * 1e3))

    def push_pending(self, times: torch.Tensor, timed_bounds) -> None:
        """the all-reduced times of this call (a tensor still being produced on the current
        stream) and the strips they were measured with; read `lag` calls later"""
        ev = None
        if times.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
        self._pending.append((times, timed_bounds, ev))

    def apply_pending(self) -> None:
        """new cut points from the times all-reduced `lag` calls ago.  The event makes the
        read safe when the calls alternate over streams (the reading stream is not the one that
        ran the reduction) and identical on every rank."""
        if len(self._pending) >= self.lag:
            times, tb, ev = self._pending.popleft()
            if ev is not None:
                ev.synchronize()
            self.update([int(t) for t in times.tolist()], tb)

    def update(self, times_us: Sequence[int], timed_bounds=None) -> None:
        """new cut points from every rank's time of one step, measured with `timed_bounds`;
        every rank calls this with the same numbers"""
        tb = timed_bounds if timed_bounds is not None else self.timed_bounds
        if tb is None or len(times_us) != self.world or any(t <= 0 for t in times_us):
            return
        dens = [t / max(b - a, 1) for (a, b), t in zip(tb, times_us)]  # cost per row
        total = float(sum(times_us))
        cuts, acc, r = [0], 0.0, 0
        for k in range(1, self.world):
            want = total * k / self.world
            while r < self.world - 1 and acc + times_us[r] < want:
                acc += times_us[r]
                r += 1
            a = tb[r][0]
            y = a + (want - acc) / max(dens[r], 1e-9)
            old = self.bounds[k][0]
            # the first correction goes all the way (equal strips are far off), later ones
            # are damped against noise
            y = old + (1.0 if self._updates == 0 else self.damping) * (y - old)
            cuts.append(self._snap(y))
        cuts.append(self.height)
        for k in range(1, self.world):  # keep every strip at least min_rows high (cuts stay in phase)
            lo = cuts[k - 1] + self.min_rows
            cuts[k] = max(cuts[k], lo + (self.phase - lo) % self.align)
        for k in range(self.world - 1, 0, -1):
            hi = cuts[k + 1] - self.min_rows
            cuts[k] = min(cuts[k], hi - (hi - self.phase) % self.align)
        if all(cuts[k] < cuts[k + 1] for k in range(self.world)):
            self.bounds = [(cuts[k], cuts[k + 1]) for k in range(self.world)]
            self._updates += 1


class GatherPipeline:
    """Lets the gather of one step's rows run behind the next step's kernels.

    With `pipeline=` sharded_depth only STARTS the gather to the root (async collective) and
    returns the depth maps of the PREVIOUS call: the root waits for and assembles them after
    it has enqueued its own PASS 1 of the next step, so the xGMI transfers hide behind ~4 ms of
    kernels instead of stalling every rank at the end of each step.  `flush()` hands out the
    maps of the last call (and must be called before the results are used or the clock is
    stopped).  Contiguous strips; gather-to-root (`dst=r`) or every rank a sink (`dst="frames"`:
    the maps handed out are then the rank's own frames)."""

    def __init__(self):
        self._inflight = None  # (work, recv, send, bounds, n, height, width, device)
        self.depth = None      # depth maps of the most recent finished step (root only)

    def finish_previous(self) -> None:
        if self._inflight is None:
            return
        if self._inflight[0] == "frames":  # dst="frames": one gather per destination rank
            _tag, works, recv, bounds, n_own, height, width, device = self._inflight
            self._inflight = None
            for work, _part in works:
                work.wait()
            own = torch.empty((n_own, height, width), dtype=torch.uint8, device=device)
            if recv is not None:
                recv = recv.to(device)
                for r, (a, b) in enumerate(bounds):
                    if b > a:
                        own[:, a:b] = recv[r, :, : b - a]
            self.depth = own
            return
        work, recv, _send, bounds, n, height, width, device = self._inflight
        self._inflight = None
        if work is not None:
            work.wait()  # the current stream waits for the collective; the host does not
        if recv is None:
            self.depth = None
            return
        recv = recv.to(device)
        depth = torch.empty((n, height, width), dtype=torch.uint8, device=device)
        for r, (a, b) in enumerate(bounds):
            if b > a:
                depth[:, a:b] = recv[r, :, : b - a]
        self.depth = depth

    def flush(self):
        """-> depth maps of the last call (None on the other ranks)"""
        self.finish_previous()
        return self.depth


class StripWorker:
    """What a rank must be able to do for one frame."""

    def match_strip(self, frame, rows: Tuple[int, int], max_out: Optional[torch.Tensor] = None,
                    pattern: Optional[Tuple[int, int, int]] = None
                    ) -> Tuple[torch.Tensor, torch.Tensor]:
        """PASS 1 on rows [r0,r1): -> (raw u8 [r1-r0, W], max int32 [1]), same device.
        `max_out`, when given, is a zeroed int32 [1] view to write the max into.  With
        `pattern` = (row_block, row_mod, row_rem) only rows y of the range with
        (y // row_block) % row_mod == row_rem are computed (the others are left untouched)."""
        raise NotImplementedError

    def normalise(self, rows: torch.Tensor, max_word: torch.Tensor, first_row: int = 0,
                  pattern: Optional[Tuple[int, int, int]] = None) -> None:
        """PASS 2 in place: rows = rows*255/max (0 when max == 0), owned rows only."""
        raise NotImplementedError


class EngineStripWorker(StripWorker):
    """Strips on the HIP engine; frames are (main, additional) CUDA uint8 tensors."""

    def __init__(self, engine, precision: Sequence[int], max_steps: int = 255,
                 tile: Tuple[int, int] = (1, 1), mode: int = 0, d_max: int = 0, win: int = 0):
        self.engine = engine
        self.kw = dict(max_steps=max_steps, tile=tile, mode=mode, d_max=d_max, win=win)
        self.precision = list(precision)
        self._sets = {}  # id(frames) -> described frame set (see _described)

    def match_strip(self, frame, rows, max_out=None, pattern=None):
        main, add = frame
        h, w = main.shape[0], main.shape[1]
        full = torch.empty((h, w), dtype=torch.uint8, device=main.device)
        mx = max_out if max_out is not None else torch.zeros(1, dtype=torch.int32,
                                                             device=main.device)
        if rows[1] > rows[0]:
            self.engine.depth_from_additional(main, add, self.precision, rows=rows,
                                              row_pattern=pattern, raw_only=True, out=full,
                                              max_out=mx, asynchronous=True, **self.kw)
        return full[rows[0]:rows[1]], mx

    def normalise(self, rows, max_word, first_row=0, pattern=None):
        if rows.shape[0]:
            self.engine.normalise_rows(rows, max_word, asynchronous=True, first_row=first_row,
                                       row_pattern=pattern)

    # -- all frames of a step in one set of launches (ring mode with 1x1 tiles, SAD mode) -----
    def can_batch(self) -> bool:
        return self.kw["mode"] == 1 or tuple(self.kw["tile"]) == (1, 1)

    def _described(self, frames):
        """ctypes descriptions of a frame set and of the output planes that go with it, built
        once per frame set (a step of 64 frames is 192 images; describing them on every call
        costs the host more than the kernels take).  The output planes are reused from call to
        call: everything that reads them is enqueued on the same stream before the next call."""
        from .engine import PlaneList
        key = id(frames)
        hit = self._sets.get(key)
        # a hit must be the same tensors, not just the same list object: a caller that refills
        # one list in place (frames[i] = new pair, the natural video loop) gets new descriptions
        if hit is not None and not (hit[0] is frames and len(hit[6]) == len(frames) and
                                    all(f[0] is p[0] and f[1] is p[1] for f, p in zip(frames, hit[6]))):
            hit = None
        if hit is None:
            mains = PlaneList([f[0] for f in frames], (3, 4), "main")
            adds = PlaneList([f[1] for f in frames], (3,), "additional")
            h, w = frames[0][0].shape[0], frames[0][0].shape[1]
            base = torch.empty((len(frames), h, w), dtype=torch.uint8, device=frames[0][0].device)
            outs = PlaneList([base[i] for i in range(len(frames))], (1,), "depth")
            if len(self._sets) > 8:
                self._sets.clear()
            hit = self._sets[key] = (frames, mains, adds, base, outs, {}, [(f[0], f[1]) for f in frames])
        return hit

    def match_batch(self, frames, rows, pattern=None):
        """PASS 1 for every frame -> (list of raw views [r1-r0, W], max int32 [F])."""
        _f, mains, adds, base, fulls, _views, _pairs = self._described(frames)
        dev = base.device
        mx = torch.zeros(len(frames), dtype=torch.int32, device=dev)
        if rows[1] > rows[0]:
            self.engine.depth_from_additional_batch(mains, adds, self.precision,
                                                    max_steps=self.kw["max_steps"],
                                                    mode=self.kw["mode"], d_max=self.kw["d_max"],
                                                    win=self.kw["win"], rows=rows,
                                                    row_pattern=pattern, raw_only=True,
                                                    outs=fulls, max_out=mx, asynchronous=True)
        self.last_base = base  # [F, H, W]; lets the caller gather rows of all frames at once
        views = _views.get(rows)
        if views is None:  # the strips as a described list (normalise_batch), per cut
            from .engine import PlaneList
            if len(_views) > 16:
                _views.clear()
            views = _views[rows] = PlaneList([base[i, rows[0]:rows[1]] for i in range(len(frames))], (1,), "rows") \
                if rows[1] > rows[0] else [base[i, rows[0]:rows[1]] for i in range(len(frames))]
        return views, mx

    def depth_batch(self, frames):
        """the whole path for every frame in one batched call (one rank: nothing to exchange
        between PASS 1 and PASS 2) -> (depth u8 [F, H, W], max int32 [F])"""
        _f, mains, adds, base, outs, _views, _pairs = self._described(frames)
        mx = torch.zeros(len(frames), dtype=torch.int32, device=base.device)
        self.engine.depth_from_additional_batch(mains, adds, self.precision,
                                                max_steps=self.kw["max_steps"], mode=self.kw["mode"],
                                                d_max=self.kw["d_max"], win=self.kw["win"],
                                                outs=outs, max_out=mx, asynchronous=True)
        return base, mx

    def normalise_batch(self, rows_list, max_all, first_row=0, pattern=None):
        if len(rows_list) and rows_list[0].shape[0]:
            self.engine.normalise_rows_batch(rows_list, max_all, asynchronous=True,
                                             first_row=first_row, row_pattern=pattern)


_OWNED_CACHE: dict = {}


def _owned_rows(height: int, block: int, world: int) -> List[torch.Tensor]:
    """row indices owned by each rank under the interleaved pattern (cached)"""
    key = (height, block, world)
    if key not in _OWNED_CACHE:
        ys = torch.arange(height)
        _OWNED_CACHE[key] = [ys[(ys // block) % world == r] for r in range(world)]
    return _OWNED_CACHE[key]


_OWNED_DEV: dict = {}


def _owned_on(owned: List[torch.Tensor], r: int, device) -> torch.Tensor:
    key = (id(owned), r, str(device))
    if key not in _OWNED_DEV:
        _OWNED_DEV[key] = owned[r].to(device)
    return _OWNED_DEV[key]


def sharded_depth(worker: StripWorker, frames: Sequence, height: int, width: int,
                  tile_h: int = 1, group: Optional[dist.ProcessGroup] = None,
                  device: Optional[torch.device] = None,
                  sync: Optional[Callable[[], None]] = None,
                  dst: Optional[int] = None,
                  interleave: int = 0,
                  balancer: Optional[StripBalancer] = None,
                  pipeline: Optional[GatherPipeline] = None
                  ) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
    """Depth maps of `frames`, every frame row-strip sharded over the ranks of `group`.

    Returns (depth u8 [F, H, W], max int32 [F]).  With one rank and an EngineStripWorker the
    maps are the worker's own output planes for this frame list: valid until its next call
    with the same list (copy them to keep them).  Collectives per call (not per frame): one
    all_reduce(MAX) of F words and one exchange of the u8 rows -- `dst=None`: all_gather, every
    rank gets the depth maps; `dst=r`: gather to rank r only (the other ranks return None;
    each sends its own F*rows*W bytes over its own xGMI link to r, nothing else moves);
    `dst="frames"`: frame f gathers to rank f % world, every rank returns its own frames
    [n_owned, H, W] (all ranks are sinks, no link carries more than 1/world of the maps).
    `sync` is called before each collective when the worker computes on a side stream.
    `interleave` = B > 0 deals the rows in blocks of B (even, a multiple of tile_h): rank r owns
    blocks r, r+world, ... of every frame -- the work per row is data dependent and very
    uneven (one contiguous 1/8 strip of the bench frame holds 27 % of its work), interleaved
    blocks give every rank the same mix.  `balancer` (contiguous strips only) moves the cut
    points between steps according to the ranks' measured times instead.  `pipeline`
    (GatherPipeline; needs `dst`, contiguous strips) overlaps the gather with the next call's
    kernels: the depth returned is then the PREVIOUS call's (None the first time)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = len(frames)
    pattern = None
    use_balancer = False
    if interleave and world > 1:
        pattern = (interleave, world, rank)
        owned = _owned_rows(height, interleave, world)
        bounds = [(0, height)] * world
        r0, r1 = 0, height
        rows_max = max(len(o) for o in owned)
    else:
        use_balancer = balancer is not None and world > 1 and balancer.world == world and \
            balancer.height == height
        if use_balancer:
            balancer.apply_pending()  # times exchanged during the previous call
        bounds = balancer.bounds if use_balancer else strip_bounds(height, world, tile_h)
        r0, r1 = bounds[rank]
        rows_max = max(b - a for a, b in bounds)

    batched = getattr(worker, "can_batch", lambda: False)() and n > 0 and \
        hasattr(frames[0][0], "is_cuda") and frames[0][0].is_cuda
    if world == 1 and batched and hasattr(worker, "depth_batch"):
        return worker.depth_batch(frames)  # one rank: PASS 1 -> max -> PASS 2 inside one call
    strips, maxes = [], []
    max_buf = None
    prev_us = 0
    if use_balancer:
        prev_us = balancer.previous_us()  # the previous step's share, measured with ITS bounds
        balancer.start(n > 0 and hasattr(frames[0][0], "is_cuda") and frames[0][0].is_cuda)
    if batched:
        strips, max_buf = worker.match_batch(frames, (r0, r1), pattern)
        maxes = [max_buf]
    for i, f in enumerate([] if batched else frames):
        kw = {"pattern": pattern} if pattern else {}
        if i == 0:
            raw, mx = worker.match_strip(f, (r0, r1), **kw)
            if mx.is_cuda and n > 1:  # one zeroed buffer for the other frames' maxima
                max_buf = torch.zeros(n, dtype=torch.int32, device=mx.device)
        elif max_buf is not None:
            raw, mx = worker.match_strip(f, (r0, r1), max_out=max_buf[i:i + 1], **kw)
        else:
            raw, mx = worker.match_strip(f, (r0, r1), **kw)
        strips.append(raw)
        maxes.append(mx)
    use_pipeline = pipeline is not None and world > 1 and isinstance(dst, int) and pattern is None
    use_pipeline_f = pipeline is not None and world > 1 and dst == "frames" and pattern is None
    if use_pipeline or use_pipeline_f:
        pipeline.finish_previous()  # its transfers ran behind the kernels enqueued above
    if use_balancer:
        balancer.stop()  # PASS 1 plus, on the root, the assembly of the previous step's rows
    if device is None:
        device = maxes[0].device
    if batched:
        max_all = max_buf
    elif max_buf is not None:
        max_buf[0:1] = maxes[0].reshape(1)
        max_all = max_buf
    else:
        max_all = torch.cat([m.reshape(1) for m in maxes]).to(torch.int32)
    # gloo (CPU rehearsals of the N > 1 path) cannot gather device tensors: stage through host
    via_host = world > 1 and dist.get_backend(group) == "gloo" and max_all.is_cuda
    if use_balancer:
        # the ranks' times of the previous step ride along: slot n + r belongs to rank r, the
        # others leave it 0, so the MAX reduction hands every rank all of them
        ext = torch.zeros(n + world, dtype=torch.int32, device=max_all.device)
        ext[:n] = max_all
        ext[n + rank] = prev_us
        max_all = ext
    if world > 1:
        if sync:
            sync()
        if via_host:
            m_host = max_all.cpu()
            dist.all_reduce(m_host, op=dist.ReduceOp.MAX, group=group)
            max_all = m_host.to(device)
        else:
            dist.all_reduce(max_all, op=dist.ReduceOp.MAX, group=group)  # src/depth_image.rs:124
    if use_balancer:
        # read `lag` calls later, when the reduction has long finished: no host sync here
        balancer.push_pending(max_all[n:], balancer.timed_bounds)
        max_all = max_all[:n]
    if batched:  # src/depth_image.rs:125-129
        worker.normalise_batch(strips, max_all, first_row=r0, pattern=pattern)
    for i, raw in enumerate([] if batched else strips):
        if pattern:
            worker.normalise(raw, max_all[i:i + 1], first_row=r0, pattern=pattern)
        else:
            worker.normalise(raw, max_all[i:i + 1])

    if world == 1:
        if batched and getattr(worker, "last_base", None) is not None:
            return worker.last_base[:, r0:r1], max_all
        return torch.stack([s for s in strips]), max_all

    # rows beyond this rank's own are padding that nobody reads
    send = torch.empty((n, rows_max, width), dtype=torch.uint8, device=device)
    base = getattr(worker, "last_base", None) if batched else None
    if pattern:
        mine = _owned_on(owned, rank, device)
        if base is not None:
            send[:, : len(mine)] = base.index_select(1, mine)
        else:
            for i, raw in enumerate(strips):
                send[i, : len(mine)] = raw.index_select(0, mine)
    elif r1 > r0:
        if base is not None:
            send[:, : r1 - r0] = base[:, r0:r1]
        else:
            for i, raw in enumerate(strips):
                send[i, : r1 - r0] = raw
    if sync:
        sync()
    if dst == "frames":
        # every rank is a sink: frame f's depth map gathers to rank f % world (one gather per
        # destination, all ranks' links busy at once instead of rank 0's alone).  Returns this
        # rank's frames rank, rank + world, ... as [n_owned, H, W].
        own = None
        works, recv_mine = [], None
        for d in range(world):
            idx = list(range(d, n, world))
            if not idx:
                continue
            part = send[idx].contiguous()
            part_c = part.cpu() if via_host else part
            recv_d = (torch.empty((world, len(idx), rows_max, width), dtype=torch.uint8,
                                  device=torch.device("cpu") if via_host else device)
                      if rank == d else None)
            if use_pipeline_f:  # only start the gathers; the next call (or flush) assembles
                works.append((dist.gather(part_c, list(recv_d.unbind(0)) if rank == d else None, dst=d,
                                          group=group, async_op=True), part_c))
                if rank == d:
                    recv_mine = recv_d
                continue
            dist.gather(part_c, list(recv_d.unbind(0)) if rank == d else None, dst=d, group=group)
            if rank == d:
                recv_d = recv_d.to(device)
                own = torch.empty((len(idx), height, width), dtype=torch.uint8, device=device)
                if pattern:
                    for r in range(world):
                        if len(owned[r]):
                            own[:, _owned_on(owned, r, device)] = recv_d[r, :, : len(owned[r])]
                else:
                    for r, (a, b) in enumerate(bounds):
                        if b > a:
                            own[:, a:b] = recv_d[r, :, : b - a]
        if use_pipeline_f:
            pipeline._inflight = ("frames", works, recv_mine, bounds, len(range(rank, n, world)), height, width, device)
            return pipeline.depth, max_all
        if own is None:
            own = torch.empty((0, height, width), dtype=torch.uint8, device=device)
        return own, max_all
    if dst is not None:
        coll_dev = torch.device("cpu") if via_host else device
        # the root receives straight into one [world, n, rows, W] buffer (views as the list)
        recv = (torch.empty((world, n, rows_max, width), dtype=torch.uint8, device=coll_dev)
                if rank == dst else None)
        send_c = send.cpu() if via_host else send
        if use_pipeline:
            work = dist.gather(send_c, list(recv.unbind(0)) if rank == dst else None, dst=dst,
                               group=group, async_op=True)
            pipeline._inflight = (work, recv, send_c, bounds, n, height, width, device)
            return pipeline.depth, max_all
        dist.gather(send_c, list(recv.unbind(0)) if rank == dst else None, dst=dst, group=group)
        if rank != dst:
            return None, max_all
        recv = recv.to(device)
    else:
        recv = torch.empty((world, n, rows_max, width), dtype=torch.uint8, device=device)
        if via_host:
            r_host = torch.empty(recv.shape, dtype=torch.uint8)
            dist.all_gather_into_tensor(r_host.view(-1), send.cpu().view(-1), group=group)
            recv.copy_(r_host)
        else:
            dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
    depth = torch.empty((n, height, width), dtype=torch.uint8, device=device)
    if pattern:
        for r in range(world):
            if len(owned[r]):
                depth[:, _owned_on(owned, r, device)] = recv[r, :, : len(owned[r])]
        return depth, max_all
    for r, (a, b) in enumerate(bounds):
        if b > a:
            depth[:, a:b] = recv[r, :, : b - a]
    return depth, max_all


def frame_sharded_depth(worker: StripWorker, frames: Sequence, height: int, width: int,
                        group: Optional[dist.ProcessGroup] = None,
                        device: Optional[torch.device] = None,
                        sync: Optional[Callable[[], None]] = None,
                        dst=None, own_frames: Optional[Sequence] = None
                        ) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
    """Depth maps of `frames` with WHOLE frames dealt to the ranks: frame f belongs to rank
    f % world (BASELINE config 4: 64 independent pairs over 8 GPUs).  No halo -- a rank packs
    only the additional images of its own frames --, no cross-rank max (the max is per frame,
    src/depth_image.rs:124), no balancer; the only exchange is the optional one of the finished
    maps: `dst=None` every rank keeps the maps of its own frames (returns [n_owned, H, W],
    max [n_owned]); `dst=r` gathers all maps to rank r (returns [F, H, W] there, None elsewhere).
    A single 8K frame cannot be split this way -- row strips (sharded_depth) remain the only
    option there and stay the default.  `own_frames`: the rank's frames as a list object that
    stays the same from call to call (EngineStripWorker describes a frame set once per list)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = len(frames)
    idx = list(range(rank, n, world))
    own = own_frames if own_frames is not None else [frames[i] for i in idx]
    batched = getattr(worker, "can_batch", lambda: False)() and len(own) > 0 and \
        hasattr(own[0][0], "is_cuda") and own[0][0].is_cuda and hasattr(worker, "depth_batch")
    if batched:
        depth, mx = worker.depth_batch(own)  # PASS 1 -> max -> PASS 2 inside one call
    else:
        maps, maxes = [], []
        for f in own:
            raw, m = worker.match_strip(f, (0, height))
            worker.normalise(raw, m)
            maps.append(raw)
            maxes.append(m.reshape(1))
        if device is None and maps:
            device = maps[0].device
        depth = torch.stack(maps) if maps else torch.empty((0, height, width), dtype=torch.uint8, device=device)
        mx = torch.cat(maxes).to(torch.int32) if maxes else torch.empty(0, dtype=torch.int32, device=device)
    if dst is None or world == 1:
        return depth, mx
    if device is None:
        device = depth.device
    via_host = dist.get_backend(group) == "gloo" and depth.is_cuda
    n_max = (n + world - 1) // world
    send = torch.zeros((n_max, height, width), dtype=torch.uint8, device=device)
    send[: len(idx)] = depth
    send_m = torch.zeros(n_max, dtype=torch.int32, device=device)
    send_m[: len(idx)] = mx
    if sync:
        sync()
    coll_dev = torch.device("cpu") if via_host else device
    recv = torch.empty((world, n_max, height, width), dtype=torch.uint8, device=coll_dev) if rank == dst else None
    recv_m = torch.empty((world, n_max), dtype=torch.int32, device=coll_dev) if rank == dst else None
    dist.gather(send.cpu() if via_host else send, list(recv.unbind(0)) if rank == dst else None, dst=dst, group=group)
    dist.gather(send_m.cpu() if via_host else send_m, list(recv_m.unbind(0)) if rank == dst else None, dst=dst, group=group)
    if rank != dst:
        return None, mx
    recv, recv_m = recv.to(device), recv_m.to(device)
    full = torch.empty((n, height, width), dtype=torch.uint8, device=device)
    full_m = torch.empty(n, dtype=torch.int32, device=device)
    for r in range(world):
        k = len(range(r, n, world))
        if k:
            full[r::world] = recv[r, :k]
            full_m[r::world] = recv_m[r, :k]
    return full, full_m


# ---- consumers of a row-sharded depth map (src/depth_image.rs:162-247) ---------------------
class ZoneWorker:
    """What a rank must be able to do on its rows of a depth map for select_foreground."""

    def histogram(self, rows: torch.Tensor) -> torch.Tensor:
        """256 counts (int32/int64 tensor on the collective's device) of the u8 rows."""
        raise NotImplementedError

    def zones(self, counts: torch.Tensor, n_zones: int):
        """depth_split's zones [(lo, hi), ...] from the summed counts (same on every rank)."""
        raise NotImplementedError

    def slice(self, rows: torch.Tensor, lo: int, hi: int) -> torch.Tensor:
        """mask rows: 255 where lo <= depth <= hi else 0"""
        raise NotImplementedError


class EngineZoneWorker(ZoneWorker):
    """The three steps on the HIP engine (CUDA uint8 row tensors)."""

    def __init__(self, engine):
        self.engine = engine

    def histogram(self, rows):
        return self.engine.depth_histogram(rows, asynchronous=True)

    def zones(self, counts, n_zones):
        return self.engine.zones_from_histogram(counts, n_zones)

    def slice(self, rows, lo, hi):
        return self.engine.slice(rows, lo, hi, asynchronous=True)


def sharded_select_foreground(worker: ZoneWorker, rows: torch.Tensor,
                              group: Optional[dist.ProcessGroup] = None,
                              sync: Optional[Callable[[], None]] = None):
    """DepthImage::select_foreground (src/depth_image.rs:221-224) on a depth map whose rows are
    sharded over the ranks of `group`; `rows` = this rank's [r, W] u8 rows (may be empty).

    depth_split's only pass over pixels is a histogram, so the one exchange is an
    all_reduce(SUM) of 256 counts; every rank then derives the same zones and slices its own
    rows.  Returns (mask rows [r, W], (from, to)); gather the mask rows like the depth rows.
    Inputs on which the reference panics raise the engine's SplitError on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rows.numel():
        counts = worker.histogram(rows)
    else:
        counts = torch.zeros(256, dtype=torch.int32, device=rows.device)
    if world > 1:
        if sync:
            sync()
        if dist.get_backend(group) == "gloo" and counts.is_cuda:
            host = counts.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            counts = host.to(counts.device)
        else:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    lo, hi = worker.zones(counts, 2)[0]
    mask = worker.slice(rows, lo, hi) if rows.numel() else torch.empty_like(rows)
    return mask, (lo, hi)
