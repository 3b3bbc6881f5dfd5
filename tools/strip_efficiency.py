"""Parallel efficiency of the row-strip sharding, measured on ONE GPU by running each rank's
share of a step alone: a step of N ranks holds 8 N frames, rank r
computes its rows of all of them in one batched call; efficiency = time of 8 full frames /
time of the slowest rank's share.  No interconnect is involved (the gather overlaps the next
step's kernels in the real flow), so this is the compute-balance part of the scaling only.

Whole frames per rank (frame f on rank f % N, `bench.py --shard frames`) need no table: a rank's
share IS 8 full frames of the undivided step -- no halo rows to pack, no cross-rank max, no
balancer -- 1.00 by construction, printed below from the same measurement for reference.

Three ways of dealing the rows: equal contiguous strips, contiguous strips whose cut points
follow the measured times (StripBalancer, with the two-step lag of the real flow), interleaved
16-row blocks -- each with the SAME frames every step and with `--rotate K` distinct frame sets
cycled step by step (content that changes under the balancer's feet).

    python tools/strip_efficiency.py [--rotate 3] [--steps 12]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from stepth_amd import MODE_RING_FIRST_HIT, Engine, synth
from stepth_amd.sharding import EngineStripWorker, StripBalancer, strip_bounds

ap = argparse.ArgumentParser()
ap.add_argument("--rotate", type=int, default=3)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--ranks", type=int, nargs="*", default=[2, 4, 8])
args = ap.parse_args()
W, H, D = 1920, 1080, 128
dev = torch.device("cuda", 0)
eng = Engine(0)
eng.use_torch_stream()
worker = EngineStripWorker(eng, [36] * 3, max_steps=D, tile=(1, 1), mode=MODE_RING_FIRST_HIT)
fl = [lambda a: a, lambda a: a[:, ::-1], lambda a: a[::-1], lambda a: a[::-1, ::-1]]
sets = []
for k in range(max(1, args.rotate)):
    base = [synth.stereo_pair(W, H, D, seed=0x5EED + 2 * k + i)[:2] for i in range(2)]
    eight = [(np.ascontiguousarray(fl[(i // 2) % 4](base[i % 2][0])), np.ascontiguousarray(fl[(i // 2) % 4](base[i % 2][1])))
             for i in range(8)]
    sets.append([(torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)) for l, r in eight])


def timed(frames, rows, pattern=None, reps=3):
    """GPU time of one share: `reps` calls enqueued back to back (the host's argument marshalling
    for 8 N frames then overlaps the previous call's kernels, as in the pipelined real flow)"""
    worker.match_batch(frames, rows, pattern)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        worker.match_batch(frames, rows, pattern)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


for s in sets:  # warm up buffers
    timed(s, (0, H))
full = [np.median([timed(s, (0, H)) for _ in range(5)]) for s in sets]
print(f"8 full frames: {[round(t, 3) for t in full]} ms per frame set")
print("whole frames per rank (frame f on rank f % N): a rank's share = the 8 full frames above -> 1.000 at any N "
      "(no halo, no all_reduce; the optional gather of finished maps is the only traffic)")
print("ranks | frames rotate | equal strips | balanced strips (cut points follow times, lag 2) | interleaved 16-row blocks")
for n in args.ranks:
    for rot in sorted({1, max(1, args.rotate)}):
        res = {}
        equal_bounds = StripBalancer(H, n, align=8, phase=7).bounds  # equal strips, wave-aligned cuts
        for mode in ("equal", "balanced", "interleave16"):
            bal = StripBalancer(H, n, align=8, phase=7)
            pending = []  # (times, bounds) waiting for their two-step lag
            effs = []
            for step in range(args.steps):
                k = step % rot
                frames = sets[k] * n  # 8 n frames of this step's content
                if mode == "balanced" and len(pending) >= 2:
                    t, b = pending.pop(0)
                    bal.update([max(1, int(x * 1e3)) for x in t], b)
                bounds = bal.bounds if mode == "balanced" else equal_bounds
                if mode == "interleave16":
                    times = [timed(frames, (0, H), (16, n, r)) for r in range(n)]
                else:
                    times = [timed(frames, bounds[r]) for r in range(n)]
                pending.append((times, list(bounds)))
                if step >= args.steps // 2:
                    effs.append(full[k] / max(times))
            res[mode] = float(np.mean(effs))
        print(f"{n:5d} | {rot:13d} | {res['equal']:12.3f} | {res['balanced']:15.3f} | {res['interleave16']:12.3f}")
eng.close()
