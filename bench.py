#!/usr/bin/env python3
"""bench.py -- Mpixels/s of depth output for stepth's open_depth_from_additional hot path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched under
torch.distributed.run, one rank per GPU.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json metric: "Mpixels/sec depth output, 1080p stereo d_max=128"):
  a step = one pass of the hot path over a batch of `8 * N` synthetic 1920x1080 stereo pairs
  (inputs resident in HBM), reference-exact mode (square-ring first hit,
  /root/reference/src/helpers.rs:9-54), one group per pixel, max_steps = d_max = 128,
  precision [36;3] (Readme.md:14).  All frames of a step go through ONE set of launches
  (stepth_depth_from_additional_batch): pack additional -> dense ring search (ring 0, search,
  distance, max per frame: ring_tile_kernel) -> normalise + paint.  With N > 1 every frame is
  sharded by row strips over the N ranks, with one all_reduce(MAX) and one gather of the depth
  rows to rank 0 per step (RCCL); per-GPU work is constant in N (weak scaling).
  value = frames * 1920*1080 / step time, whole job.

roofline: HBM-bound by contract (scan/compare, no MFMA).  achieved = algorithmic bytes per
  ring_tile_kernel launch (7 B per output pixel: 3 B main RGB + 3 B additional RGB + 1 B depth,
  SURVEY.md 8d) / that kernel's mean launch time, measured live with HIP events on the
  engine's stream; peak = 8 TB/s.  The ring search is data dependent, so the number of items
  examined per pixel is printed next to it.  `traffic` = HBM bytes per launch from the
  committed rocprofv3 PMC passes (profiles/r3_hbm_traffic.json: means over the headline step's
  launches only), valid for the profiled shape; they count requests leaving the L2 -- the headline
  cycles one frame set that fits the 256 MB Infinity Cache, so part of it never reaches HBM.
cpu_baseline: the C restatement of the reference algorithm (oracle/, "port" -- NOT the Rust
  binary) on one frame of the same workload, all host cores with dynamic scheduling, plus the
  reference's own static 8-chunk partition (src/depth_image.rs:111-112) as `ref8_value`.

Extra objects (N = 1, outside the timed region, each a few launches): `two_streams`,
`config2_1080p_d64`, `config3_4k_d128` (north_star's target configuration),
`config5_8k_d256_1gpu`, `asset_tiled_1080p` (the reference's own image pair, mirror-tiled),
`adaptive_flow` (grouping -> match -> normalise -> paint, the way the reference runs),
`end_to_end` (host-resident frames, PCIe-inclusive), `sad_mode` and `sad_config*` (the windowed-SAD
mode on the headline frames and on BASELINE's configurations read literally), `with_gaussian_tail`,
`foreground_tail`.  Every ring leg carries its own kernel time, Mpixels/s, HBM fraction and a
bit-exact flag against the oracle (whole frame 0 at 1080p, 20 000 sampled pixels above).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# With several HIP streams in use (--streams 2) the runtime's default of 4 hardware queues
# sometimes puts two streams on one queue and serialises them; 8 do not.  Must be set before
# the HIP runtime starts, i.e. before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
ALGO_BYTES_PER_PIXEL = 7.0
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r3_hbm_traffic.json")
DOMINANT_KERNEL = "ring_tile_kernel"  # tests/test_bench_contract.py checks the profile names it


def flips():
    return [lambda a: a, lambda a: a[:, ::-1], lambda a: a[::-1], lambda a: a[::-1, ::-1]]


def make_frames(width, height, d_max, count, seed0=0x5EED, bases=2):
    """`count` distinct synthetic pairs from `bases` seeded base pairs and their flips."""
    from stepth_amd import synth
    base = [synth.stereo_pair(width, height, d_max, seed=seed0 + i)[:2]
            for i in range(min(bases, count))]
    frames = []
    for i in range(count):
        l, r = base[i % len(base)]
        f = flips()[(i // len(base)) % 4]
        frames.append((np.ascontiguousarray(f(l)), np.ascontiguousarray(f(r))))
    return frames


def mirror_tile(img, width, height):
    """`img` continued by mirroring to width x height (seamless: no artificial edges)"""
    a = np.concatenate([img, img[:, ::-1]], 1)
    a = np.concatenate([a, a[::-1]], 0)
    ry, rx = -(-height // a.shape[0]), -(-width // a.shape[1])
    return np.ascontiguousarray(np.tile(a, (ry, rx, 1))[:height, :width])


def asset_frames(width, height, count):
    """the reference's own pair (assets/main.jpg + additional.jpg, committed decoded as
    tests/golden/asset_*.png) mirror-tiled to width x height, `count` frames by flips"""
    from PIL import Image
    g = os.path.join(ROOT, "tests", "golden")
    m = np.asarray(Image.open(os.path.join(g, "asset_main.png")))[..., :3]
    a = np.asarray(Image.open(os.path.join(g, "asset_additional.png")))[..., :3]
    m, a = mirror_tile(m, width, height), mirror_tile(a, width, height)
    return [(np.ascontiguousarray(flips()[i % 4](m)), np.ascontiguousarray(flips()[i % 4](a)))
            for i in range(count)]


def load_oracle():
    from oracle import oracle as O
    try:  # a host-tuned build of the same C file, kept out of the tree
        lib_path = O.build(force=True, lib_path=os.path.join("/tmp", f"stepth_oracle_native_{os.getpid()}.so"),
                           march="native")
        O._lib = O.lib(lib_path)
    except Exception:
        O.lib()
    O.set_threads(O.usable_cpus())
    return O


def cpu_baseline(O, frame, args):
    """Times the oracle (test infrastructure; allowed here as the CPU baseline) on one frame."""
    l, r = frame
    prec = [args.precision] * 3
    if args.mode == "ring":
        t0 = time.perf_counter()
        ref = O.depth_from_additional(l, r, prec, args.d_max, (1, 1), threads=O.ALLCORES)
        t_all = time.perf_counter() - t0
        t0 = time.perf_counter()
        O.depth_from_additional(l, r, prec, args.d_max, (1, 1), threads=O.REF8)
        t_ref8 = time.perf_counter() - t0
        px = l.shape[0] * l.shape[1]
        out = {"value": round(px / t_all / 1e6, 4), "unit": "Mpixels/s",
               "cores": O.hardware_threads(), "kind": "port",
               "sample": f"1 frame {l.shape[1]}x{l.shape[0]}, all cores, dynamic schedule; "
                         "C restatement of the reference algorithm, not the Rust binary",
               "seconds": round(t_all, 3),
               "ref8_value": round(px / t_ref8 / 1e6, 4),
               "ref8_note": "static len/8 chunks like src/depth_image.rs:111-112 (9 threads max)",
               "reference_probes_per_pixel": round(ref["probes"] / px, 1)}
        return out, ref["depth"]
    rows = min(l.shape[0], 32)  # SAD on CPU is ~1e3x slower: a 32-row band
    t0 = time.perf_counter()
    raw = O.sad_rows(l, r, args.d_max, args.win, prec, 0, rows, threads=O.ALLCORES)
    t_all = time.perf_counter() - t0
    out = {"value": round(rows * l.shape[1] / t_all / 1e6, 4), "unit": "Mpixels/s",
           "cores": O.hardware_threads(), "kind": "port",
           "sample": f"rows 0..{rows} of 1 frame {l.shape[1]}x{l.shape[0]}, all cores",
           "seconds": round(t_all, 3)}
    return out, raw[:rows]


def sampled_raw_check(O, raw_gpu, main, add, prec, max_steps, n=20000, seed=1):
    """GPU raw distances == oracle on `n` random pixels (1x1 groups, src/depth_image.rs:112-121)"""
    h, w = main.shape[:2]
    rng = np.random.default_rng(seed)
    ys, xs = rng.integers(0, h, n), rng.integers(0, w, n)
    g = np.zeros(n, O.GROUP_DTYPE)
    g["x"], g["y"], g["w"], g["h"] = xs, ys, 1, 1
    g["value"] = main[ys, xs, :3]
    res = O.match_groups(g, add, max_steps, prec)
    return bool(np.array_equal(res["raw"], raw_gpu[ys, xs])), res["probes"] / n


def sad_leg(eng, O, frames_np, prec, d_max, win, reps, dev, label):
    """the engine's windowed-SAD mode (north_star's literal wording; not in the reference) on
    device-resident frames, one batched call per repetition; a 16-row band of frame 0 is
    compared with the CPU statement of the same mode (oracle.sad_rows)"""
    from stepth_amd import MODE_SAD_WINDOW
    from stepth_amd.engine import PlaneList
    h, w = frames_np[0][0].shape[:2]
    mains = [torch.from_numpy(l).to(dev) for l, _ in frames_np]
    adds = [torch.from_numpy(r).to(dev) for _, r in frames_np]
    outs = [torch.empty((h, w), dtype=torch.uint8, device=dev) for _ in frames_np]
    n = len(frames_np)
    pm, pa, po = PlaneList(mains, (3, 4), "main"), PlaneList(adds, (3,), "additional"), PlaneList(outs, (1,), "depth")

    def call(**kw):
        return eng.depth_from_additional_batch(pm, pa, prec, mode=MODE_SAD_WINDOW, d_max=d_max, win=win,
                                               outs=po, asynchronous=True, **kw)
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    eng.kernel_time_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    kms, kl = eng.kernel_time_read()
    eng.kernel_time_enable(False)
    k_avg = kms / max(kl, 1)
    px = n * w * h
    out = {"value": round(px / dt / 1e6, 1), "unit": "Mpixels/s", "workload": label,
           "frames_per_launch": n // max(kl // reps, 1), "ms_per_call": round(dt * 1e3, 4),
           "kernel": "sad_search_kernel", "kernel_avg_ms": round(k_avg, 4),
           "hbm_frac": round(ALGO_BYTES_PER_PIXEL * px / max(kl // reps, 1) / (k_avg * 1e-3) / 1e9
                             / HBM_PEAK_GBS, 6)}
    r0 = (h // 2) & ~15
    raw, _ = eng.depth_from_additional_batch(mains[:1], adds[:1], prec, mode=MODE_SAD_WINDOW, d_max=d_max,
                                             win=win, raw_only=True)
    ref = O.sad_rows(frames_np[0][0], frames_np[0][1], d_max, win, prec, r0, r0 + 16)
    out["parity_16_row_band_bit_exact"] = bool(np.array_equal(raw[0][r0:r0 + 16].cpu().numpy(), ref[r0:r0 + 16]))
    return out


def cpu_band_ms(O, main, add, prec, max_steps, rows=16, best_of=3):
    """the oracle on a band of `rows` pixel rows in the middle of a frame, one group per pixel:
    best of `best_of` runs, all cores (dynamic schedule) and the reference's 8 static chunks
    (src/depth_image.rs:111-112) -> (ms_all, ms_ref8, band pixels).  Whole 4K / 8K frames take
    minutes on the CPU; the band is the stated sample."""
    h, w = main.shape[:2]
    y0 = (h // 2) & ~15
    ys, xs = np.mgrid[y0:y0 + rows, 0:w]
    g = np.zeros(rows * w, O.GROUP_DTYPE)
    g["x"], g["y"], g["w"], g["h"] = xs.ravel(), ys.ravel(), 1, 1
    g["value"] = main[ys.ravel(), xs.ravel(), :3]
    best = {}
    for name, thr in (("all", O.ALLCORES), ("ref8", O.REF8)):
        t = []
        for _ in range(best_of):
            t0 = time.perf_counter()
            O.match_groups(g, add, max_steps, prec, threads=thr)
            t.append(time.perf_counter() - t0)
        best[name] = min(t) * 1e3
    return best["all"], best["ref8"], rows * w


def ring_leg(eng, O, frames_np, prec, max_steps, reps, check, dev, label, heavy_first=False, cpu_rows=True):
    """one batched call per repetition on device-resident frames; -> dict for the JSON line"""
    from stepth_amd.engine import PlaneList
    h, w = frames_np[0][0].shape[:2]
    mains = [torch.from_numpy(l).to(dev) for l, _ in frames_np]
    adds = [torch.from_numpy(r).to(dev) for _, r in frames_np]
    outs = [torch.empty((h, w), dtype=torch.uint8, device=dev) for _ in frames_np]
    n = len(frames_np)
    # described once, like a caller that streams frames through fixed buffers would
    pm, pa, po = PlaneList(mains, (3, 4), "main"), PlaneList(adds, (3,), "additional"), PlaneList(outs, (1,), "depth")

    def call(**kw):
        return eng.depth_from_additional_batch(pm, pa, prec, max_steps=max_steps, outs=po,
                                               asynchronous=True, heavy_first=heavy_first, **kw)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    eng.kernel_time_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    kms, kl = eng.kernel_time_read()
    eng.kernel_time_enable(False)
    k_avg = kms / max(kl, 1)
    px = n * w * h
    out = {"value": round(px / dt / 1e6, 1), "unit": "Mpixels/s", "workload": label,
           "frames_per_launch": n // max(kl // reps, 1), "ms_per_call": round(dt * 1e3, 4),
           "kernel": DOMINANT_KERNEL, "kernel_avg_ms": round(k_avg, 4),
           "hbm_frac": round(ALGO_BYTES_PER_PIXEL * px / max(kl // reps, 1) / (k_avg * 1e-3) / 1e9
                             / HBM_PEAK_GBS, 6)}
    eng.stats_enable(True)
    call()
    st = eng.stats()
    raw = eng.stats_raw(8)
    eng.stats_enable(False)
    out["gpu_items_examined_per_pixel"] = round(st["probes"] / max(st["groups"], 1), 1)
    out["gpu_blocks_probed_per_pixel"] = round(st["blocks_probed"] / max(st["groups"], 1), 3)
    out["ring0_hit_fraction"] = round(st["ring0_hits"] / max(st["groups"], 1), 4)
    # how sharp the filters are: block probes that did not improve the query's result, and
    # superblock visits in which no block passed its bound test for any query of the wave
    out["block_probes_without_gain"] = round(raw[7] / max(st["blocks_probed"], 1), 3)
    out["superblock_steps_without_candidate"] = round(raw[6] / max(st["superblock_steps"], 1), 3)
    if heavy_first:
        out["launch_order"] = "heavy first (STEPTH_FLAG_HEAVY_FIRST: three small launches estimate and sort the tiles)"
    if cpu_rows and O is not None:
        # the CPU rows of BASELINE.md section 3 on a stated sample: 16 pixel rows of frame 0
        ms_all, ms_ref8, band_px = cpu_band_ms(O, frames_np[0][0], frames_np[0][1], prec, max_steps)
        out["cpu_ms_16_row_band"] = round(ms_all, 2)
        out["cpu_ref8_ms_16_row_band"] = round(ms_ref8, 2)
        out["cpu_mpixels_s"] = round(band_px / ms_all / 1e3, 4)
        out["cpu_ref8_mpixels_s"] = round(band_px / ms_ref8 / 1e3, 4)
        out["cpu_sample"] = (f"oracle (C restatement, not the Rust binary) on rows {(h // 2) & ~15}..+16 of frame 0, "
                             f"{O.usable_cpus()} cores dynamic / 8 static chunks, best of 3")
    if check == "full":
        depth, _ = eng.depth_from_additional_batch(mains[:1], adds[:1], prec, max_steps=max_steps,
                                                   heavy_first=heavy_first)
        ref = O.depth_from_additional(frames_np[0][0], frames_np[0][1], prec, max_steps, (1, 1))
        out["parity_frame0_bit_exact"] = bool(np.array_equal(depth[0].cpu().numpy(), ref["depth"]))
        out["reference_probes_per_pixel"] = round(ref["probes"] / (w * h), 1)
    elif check == "sample":
        raw, _ = eng.depth_from_additional_batch(mains[:1], adds[:1], prec, max_steps=max_steps,
                                                 raw_only=True, heavy_first=heavy_first)
        ok, ppp = sampled_raw_check(O, raw[0].cpu().numpy(), frames_np[0][0], frames_np[0][1], prec,
                                    max_steps)
        out["parity_20000_sampled_pixels_bit_exact"] = ok
        out["reference_probes_per_pixel"] = round(ppp, 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=["ring", "sad"], default="ring")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--d-max", dest="d_max", type=int, default=128)
    ap.add_argument("--win", type=int, default=5)
    ap.add_argument("--precision", type=int, default=36)
    ap.add_argument("--frames-per-gpu", dest="fpg", type=int, default=8,
                    help="frames per step and GPU (BASELINE config 4: 64 pairs over 8 GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="N = 1: only the headline (skips the other configurations)")
    ap.add_argument("--interleave", type=int, default=0,
                    help="N > 1, ring mode: deal rows in interleaved blocks of this many rows "
                         "(0 = contiguous strips; blocks of 16 balance the data-dependent work "
                         "whatever the image content)")
    ap.add_argument("--streams", type=int, default=1,
                    help="steps alternate over this many engines/HIP streams (1 = strictly one "
                         "step after the other, the headline; 2 is reported as `two_streams`)")
    ap.add_argument("--rotate-frames", dest="rotate", type=int, default=3,
                    help="N > 1: cycle through this many distinct frame sets, so that the strip "
                         "balancer faces content that changes from step to step (1 = the same "
                         "frames every step)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: finish every step's gather before the next step starts")
    ap.add_argument("--no-balance", action="store_true",
                    help="N > 1: keep equal row strips instead of work-weighted ones")
    ap.add_argument("--dst", default="frames",
                    help="N > 1, strips: where the depth rows gather: 'frames' = frame f to rank f %% N, every "
                         "rank a sink (default: rank 0's N-1 inbound links were the exposed part), or a rank "
                         "number (gather to that rank)")
    ap.add_argument("--shard", choices=["strips", "frames"], default="strips",
                    help="N > 1: what `value` is measured on -- row strips of every frame (north_star's "
                         "partition, the only one for a single big frame) or whole frames per rank (frame f "
                         "on rank f %% N: no halo, no all_reduce).  The other split is timed too and "
                         "reported beside it.")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                         "the multi-rank path with several ranks on one GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run "
                     "(--nproc-per-node N)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hot path has no CPU fallback")
    if args.backend != "nccl":  # rehearsal: ranks share whatever GPUs exist
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from stepth_amd import MODE_RING_FIRST_HIT, MODE_SAD_WINDOW, Engine
    from stepth_amd.sharding import (EngineStripWorker, GatherPipeline, StripBalancer, frame_sharded_depth,
                                     sharded_depth, strip_bounds)

    W, H = args.width, args.height
    n_frames = args.fpg * world
    # N > 1: `rotate` distinct frame sets (different seeds), cycled step by step
    n_sets = max(1, args.rotate) if world > 1 else 1
    sets_np = [make_frames(W, H, args.d_max, n_frames, seed0=0x5EED + 2 * k) for k in range(n_sets)]
    frames_np = sets_np[0]
    frame_sets = [[(torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)) for l, r in s]
                  for s in sets_np]
    prec = [args.precision] * 3

    dst = None if world == 1 else (args.dst if args.dst == "frames" else int(args.dst))
    # whole frames per rank: the rank's own frames of every set, as lists that stay the same
    own_sets = [[fs[i] for i in range(rank, n_frames, world)] for fs in frame_sets]

    def run_timed(n_slots, steps, warmup, shard="strips"):
        """`steps` timed steps alternating over n_slots engines / HIP streams
        -> (seconds, kernel ms, kernel launches, depth of the last step, max, balancer)"""
        torch.cuda.synchronize()
        engines = [Engine(local_rank) for _ in range(n_slots)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(n_slots)] if n_slots > 1 else [None]
        for e_, s_ in zip(engines, streams):
            if s_ is None:
                e_.use_torch_stream()
            else:
                with torch.cuda.stream(s_):
                    e_.use_torch_stream()

        def make_worker(e_):
            if args.mode == "ring":
                return EngineStripWorker(e_, prec, max_steps=args.d_max, tile=(1, 1), mode=MODE_RING_FIRST_HIT)
            return EngineStripWorker(e_, prec, mode=MODE_SAD_WINDOW, d_max=args.d_max, win=args.win)

        workers = [make_worker(e_) for e_ in engines]
        # N > 1, contiguous strips: the cut points follow the ranks' measured times from step
        # to step (the cost of a row is data dependent); the warm-up steps let them settle
        # (lag 2: a step reads the times of the step before the previous one, which has long
        # finished -- reading the previous step's would stall the host until its kernels end)
        balancer = StripBalancer(H, world, align=8, phase=7, lag=max(2, n_slots)) if world > 1 and not args.interleave \
            and not args.no_balance and shard == "strips" else None
        # N > 1: the gather of a step's rows to rank 0 runs behind the following kernels; the
        # outstanding gathers are flushed inside the timed region
        pipelines = [GatherPipeline() if world > 1 and not args.interleave and not args.no_overlap and shard == "strips"
                     else None for _ in range(n_slots)]
        step_no = [0]

        def step():
            slot = step_no[0] % n_slots
            frames = frame_sets[step_no[0] % n_sets]
            step_no[0] += 1

            own = own_sets[(step_no[0] - 1) % n_sets]

            def run():
                if shard == "frames" and world > 1:  # frame f wholly on rank f % N, maps stay there
                    return frame_sharded_depth(workers[slot], frames, H, W, own_frames=own)
                return sharded_depth(workers[slot], frames, H, W, dst=dst,
                                     interleave=args.interleave if args.mode == "ring" else 0,
                                     balancer=balancer, pipeline=pipelines[slot])
            if streams[slot] is None:
                return run()
            with torch.cuda.stream(streams[slot]):
                return run()

        def flush():
            """finishes the gathers still in flight, oldest first; -> depth of the last step"""
            last = None
            for k in range(n_slots):
                slot = (step_no[0] + k) % n_slots  # the slot of the oldest outstanding step first
                if pipelines[slot] is None:
                    continue
                if streams[slot] is None:
                    d = pipelines[slot].flush()
                else:
                    with torch.cuda.stream(streams[slot]):
                        d = pipelines[slot].flush()
                if slot == (step_no[0] - 1) % n_slots:
                    last = d
            return last

        def fence():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        depth = mx = None
        for _ in range(warmup):
            depth, mx = step()
        if pipelines[0] is not None:
            flush()
        fence()
        for e_ in engines:
            e_.kernel_time_enable(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            depth, mx = step()
        if pipelines[0] is not None:
            depth = flush()  # the last step's rows; part of the timed work
        fence()
        dt = time.perf_counter() - t0
        kms = kl = 0
        for e_ in engines:
            k_, l_ = e_.kernel_time_read()
            kms, kl = kms + k_, kl + l_
            e_.kernel_time_enable(False)
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        last_set = (step_no[0] - 1) % n_sets
        for e_ in engines[1:]:
            e_.close()
        return float(t.item()), kms, kl, depth, mx, balancer, engines[0], workers[0], last_set

    dt, kms, klaunches, depth, _mx, balancer, eng, worker, last_set = run_timed(
        max(1, args.streams), args.steps, args.warmup, args.shard)
    other = None
    if world > 1 and args.mode == "ring":
        # the other split of the same step, timed the same way, reported beside `value`
        o_shard = "frames" if args.shard == "strips" else "strips"
        o_dt, o_kms, o_kl, _d, _m, _b, o_eng, _w, _ls = run_timed(max(1, args.streams), args.steps, args.warmup, o_shard)
        o_eng.close()
        other = (o_shard, o_dt, o_kms / max(o_kl, 1))
    eng.use_torch_stream()  # the extra legs below run on the default stream

    # items examined per pixel on this rank's strips (data-dependent work), outside the timed region
    probes_pp = None
    if args.mode == "ring":
        r0, r1 = (balancer.bounds if balancer is not None else strip_bounds(H, world))[rank]
        if world > 1 and args.shard == "frames":
            r0, r1 = 0, H
        pat = None
        if args.interleave and world > 1:
            r0, r1, pat = 0, H, (args.interleave, world, rank)
        eng.stats_enable(True)
        tot_p = tot_g = 0
        for f in frame_sets[0][: min(n_frames, 8)]:
            worker.match_strip(f, (r0, r1), pattern=pat)
            s = eng.stats()
            tot_p += s["probes"]
            tot_g += s["groups"]
        eng.stats_enable(False)
        probes_pp = tot_p / max(tot_g, 1)

    if rank == 0:
        px_step = n_frames * W * H
        value = px_step * args.steps / dt / 1e6
        strip_rows = balancer.bounds[0] if balancer is not None else strip_bounds(H, world)[0]
        # one launch carries this rank's rows of up to 64 frames of the step (batched entry point)
        frames_per_launch = min(n_frames, 64)
        if world > 1 and args.shard == "frames":  # whole frames: the rank's own ones in one launch
            strip_rows, frames_per_launch = (0, H), min(len(own_sets[0]), 64)
        px_launch = (strip_rows[1] - strip_rows[0]) * W * frames_per_launch
        k_avg_ms = kms / max(klaunches, 1)
        achieved = ALGO_BYTES_PER_PIXEL * px_launch / (k_avg_ms * 1e-3) / 1e9 if klaunches else 0.0
        # HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE,
        # gfx950-corrected) -- only for the shape that was profiled
        traffic = None
        issue = None
        if args.mode == "ring" and world == 1 and (W, H, args.d_max) == (1920, 1080, 128) \
                and os.path.exists(TRAFFIC_FILE):
            prof = json.load(open(TRAFFIC_FILE))
            per_launch = prof.get("dominant_kernel_traffic_bytes_per_launch_corrected")
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from profile_summaries import kernel_source_sha16
            # only while the kernels are the ones that were profiled (else null: re-profile)
            if per_launch and prof.get("frames_per_launch") and DOMINANT_KERNEL in prof.get("dominant_kernel", "") \
                    and prof.get("kernel_source_sha16") == kernel_source_sha16():
                traffic = int(per_launch * frames_per_launch / prof["frames_per_launch"])
                sq = prof.get("dominant_kernel_sq_counters_per_launch") or {}
                if sq.get("SQ_ACTIVE_INST_VALU") and klaunches and frames_per_launch == prof["frames_per_launch"]:
                    # what actually bounds the kernel: its VALU instructions against the SIMDs'
                    # issue slots (SQ_ACTIVE_INST_VALU counts quad-cycles; 256 CUs x 4 SIMDs at
                    # 2.4 GHz; counters from the committed PMC passes, time measured live)
                    simd_quad_cycles = k_avg_ms * 1e-3 * 2.4e9 / 4 * 256 * 4
                    issue = {"bound": "valu_issue",
                             "valu_insts_per_launch": sq.get("SQ_INSTS_VALU"),
                             "valu_insts_per_pixel": round(sq.get("SQ_INSTS_VALU", 0) / px_launch, 2),
                             "salu_insts_per_launch": sq.get("SQ_INSTS_SALU"),
                             "valu_busy_frac": round(min(sq["SQ_ACTIVE_INST_VALU"] / simd_quad_cycles, 1.0), 3),
                             "waves_resident_per_simd": round(sq.get("SQ_WAVE_CYCLES", 0) / simd_quad_cycles, 2),
                             "cycles_per_wave64_int_valu_inst": 3.9,
                             "source": "profiles/r3_hbm_traffic.json (rocprofv3 --pmc SQ_*), "
                                       "profiles/r3_issue_microbench.txt"}
        n_slots = max(1, args.streams)
        out = {
            "metric": ("Mpixels/sec depth output, 1080p stereo d_max=128" if args.mode == "ring"
                       else "Mpixels/sec depth output, windowed-SAD mode (not in the reference)"),
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic (seeded value-noise stereo pairs + flips, "
                                   "stepth_amd/synth.py); inputs resident in HBM",
            "config": {
                "workload": (f"{n_frames} x {W}x{H} synthetic stereo pairs per step "
                             f"({args.fpg} per GPU), "
                             + (f"ring first-hit (reference-exact), 1x1 groups, max_steps=d_max={args.d_max}"
                                if args.mode == "ring" else
                                f"windowed SAD {args.win}x{args.win}, d_max={args.d_max}")
                             + f", precision [{args.precision};3]"),
                "frames_per_step": n_frames, "width": W, "height": H, "d_max": args.d_max,
                "mode": args.mode,
                "frame_sets_cycled": n_sets,
                "parallelism": ((f"whole frames per rank (frame f on rank f % {world}): no halo, no all_reduce, "
                                  f"the maps stay on their rank ({'RCCL' if args.backend == 'nccl' else args.backend} "
                                  "only for the barrier)") if world > 1 and args.shard == "frames" else
                                ((f"interleaved {args.interleave}-row strips" if args.interleave and args.mode == "ring"
                                  else "contiguous row strips, cut points following measured per-rank times"
                                  if balancer is not None else "contiguous row strips")
                                 + f" x{world} + all_reduce(max) + gather of rows "
                                 + ("(frame f to rank f % N: every rank a sink)" if dst == "frames" else f"to rank {dst}")
                                 + f" ({'RCCL' if args.backend == 'nccl' else args.backend}"
                                 + (", overlapped with the next step's kernels)" if not args.no_overlap and not args.interleave else ")"))
                                if world > 1 else "single GPU")
                               + (f"; steps alternate over {n_slots} engines/HIP streams" if n_slots > 1
                                  else "; one HIP stream"),
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                "kernel": DOMINANT_KERNEL if args.mode == "ring" else "sad_search_kernel",
                "kernel_avg_ms": round(k_avg_ms, 4), "kernel_launches": klaunches,
                "frames_per_launch": frames_per_launch,
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_PIXEL * px_launch,
                "gpu_items_examined_per_pixel": None if probes_pp is None else round(probes_pp, 1),
                "issue": issue,
                "note": "items = bound tests + cells the GPU search examined per pixel; the "
                        "reference's own probe count for the same frame is "
                        "cpu_baseline.reference_probes_per_pixel",
            },
        }
        if other is not None:
            out["frame_split" if other[0] == "frames" else "row_strips"] = {
                "value": round(px_step * args.steps / other[1] / 1e6, 3), "unit": "Mpixels/s",
                "ms_per_step": round(other[1] / args.steps * 1e3, 4), "kernel_avg_ms": round(other[2], 4),
                "workload": "the same steps with " + ("whole frames per rank (frame f on rank f % N; no halo, no "
                            "all_reduce, no gather: BASELINE config 4's independent pairs)" if other[0] == "frames"
                            else "row strips of every frame over the ranks (north_star's partition)"),
                "note": "timed like `value` (barrier + synchronize, max over ranks), a second loop after it"}
        O = None
        if not args.no_cpu_baseline:
            O = load_oracle()
        extra = args.mode == "ring" and world == 1 and not args.no_extra_legs
        if extra:
            # -- the same headline with two steps in flight on two HIP streams
            dt2, kms2, kl2, _d2, _m2, _b2, eng2, _w2, _ls = run_timed(2, args.steps, args.warmup)
            eng2.close()
            out["two_streams"] = {
                "value": round(px_step * args.steps / dt2 / 1e6, 1), "unit": "Mpixels/s",
                "ms_per_step": round(dt2 / args.steps * 1e3, 4),
                "note": "steps alternate over 2 engines on 2 HIP streams; their dominant kernels "
                        "overlap, so the per-launch time is not comparable with the headline's",
                "kernel_avg_ms_overlapped": round(kms2 / max(kl2, 1), 4)}
        if extra and O is not None:
            # -- the other BASELINE.json configurations and the reference's own image pair
            out["config2_1080p_d64"] = ring_leg(
                eng, O, frames_np, prec, 64, 5, "full", dev,
                f"{n_frames} x 1920x1080 synthetic pairs, ring first-hit, max_steps=d_max=64")
            out["asset_tiled_1080p"] = ring_leg(
                eng, O, asset_frames(1920, 1080, 8), prec, 128, 5, "full", dev,
                "8 x 1920x1080 = the reference's assets/main.jpg + additional.jpg mirror-tiled "
                "(natural image statistics), ring first-hit, max_steps=128", heavy_first=True)
            out["asset_tiled_1080p_image_order"] = ring_leg(
                eng, O, asset_frames(1920, 1080, 8), prec, 128, 5, "full", dev,
                "same frames, workgroups in image order (the C-ABI default; round 2's figure)", cpu_rows=False)
            out["asset_tiled_1080p_max255"] = ring_leg(
                eng, O, asset_frames(1920, 1080, 8), prec, 255, 5, "sample", dev,
                "same frames, max_steps=255 (the reference's own reach, src/depth_image.rs:119)",
                heavy_first=True)
            out["headline_heavy_first"] = ring_leg(
                eng, O, frames_np, prec, args.d_max, 5, "full", dev,
                "the headline frames with STEPTH_FLAG_HEAVY_FIRST: most tiles search far, the device keeps "
                "image order -- what the flag costs where it cannot help", heavy_first=True, cpu_rows=False)
            f4k = make_frames(3840, 2160, 128, 4, bases=1)
            out["config3_4k_d128"] = ring_leg(
                eng, O, f4k, prec, 128, 3, "sample", dev,
                "4 x 3840x2160 synthetic pairs (1 seed + flips), ring first-hit, max_steps=d_max=128 "
                "(north_star's target configuration)")
            # 8K: the 4K pair continued by mirroring (generating 8K value noise takes minutes)
            f8k = [(mirror_tile(f4k[0][0], 7680, 4320), mirror_tile(f4k[0][1], 7680, 4320))]
            out["config5_8k_d256_1gpu"] = ring_leg(
                eng, O, f8k, prec, 256, 3, "sample", dev,
                "1 x 7680x4320 pair (the 4K synthetic pair mirror-tiled 2x2), ring first-hit, "
                "max_steps=d_max=256, one frame per launch")
            # -- BASELINE.json's configurations read literally (an N x N SAD window over d_max
            # disparities): the engine's second mode, which the reference does not have
            out["sad_config2_1080p_d64_5x5"] = sad_leg(
                eng, O, frames_np, prec, 64, 5, 3, dev,
                f"{n_frames} x 1920x1080 synthetic pairs, windowed SAD 5x5, d_max=64")
            out["sad_config3_4k_d128_5x5"] = sad_leg(
                eng, O, f4k, prec, 128, 5, 3, dev, "4 x 3840x2160 synthetic pairs, windowed SAD 5x5, d_max=128")
            out["sad_config5_8k_d256_9x9"] = sad_leg(
                eng, O, f8k, prec, 256, 9, 2, dev, "1 x 7680x4320 pair, windowed SAD 9x9, d_max=256")
            del f4k, f8k
        if extra:
            frames = frame_sets[0]
            # -- the path the way the reference runs it: adaptive pixel groups (the engine's
            # stand-in for disage, src/depth_image.rs:101-110) -> match -> normalise -> paint,
            # all frames through the batched entry points
            mains_t, adds_t = [f[0] for f in frames], [f[1] for f in frames]
            outs_t = [torch.empty((H, W), dtype=torch.uint8, device=dev) for _ in frames]
            if hasattr(eng, "adaptive_depth_batch"):
                def flow():
                    eng.adaptive_depth_batch(mains_t, adds_t, prec, max_steps=args.d_max, outs=outs_t,
                                             asynchronous=True)
                flow()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    flow()
                torch.cuda.synchronize()
                adt = (time.perf_counter() - t1) / 5
                out["adaptive_flow"] = {
                    "value": round(px_step / adt / 1e6, 1), "unit": "Mpixels/s",
                    "ms_per_call": round(adt * 1e3, 4),
                    "workload": f"{n_frames} frames: stepth_adaptive_depth_batch = adaptive groups -> ring first-hit per "
                                "group (the search paints its rectangles itself) -> normalise in place; groups never "
                                "leave the device (grouping rule: this engine's own, parity with disage unpinned)"}
                if O is not None:
                    d0, _ = eng.adaptive_depth_batch(mains_t[:1], adds_t[:1], prec, max_steps=args.d_max)
                    g = O.adaptive_groups(frames_np[0][0], prec)
                    ref = O.depth_from_additional(frames_np[0][0], frames_np[0][1], prec, args.d_max, groups=g)
                    out["adaptive_flow"]["parity_frame0_bit_exact"] = bool(
                        np.array_equal(d0[0].cpu().numpy(), ref["depth"]))
                    out["adaptive_flow"]["groups_frame0"] = int(len(g))
            # -- what a user of the crate gets by default (src/depth_image.rs:101-135): adaptive
            # groups -> max = 255 -> Gaussian tail, on the reference's own pair mirror-tiled
            if hasattr(eng, "adaptive_depth_batch") and O is not None:
                af = asset_frames(1920, 1080, 8)
                am = [torch.from_numpy(l).to(dev) for l, _ in af]
                aa = [torch.from_numpy(r).to(dev) for _, r in af]
                ao = [torch.empty((1080, 1920), dtype=torch.uint8, device=dev) for _ in af]

                def dflow():
                    eng.adaptive_depth_batch(am, aa, prec, max_steps=255, outs=ao, gaussian_tail=True,
                                             asynchronous=True)
                dflow()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    dflow()
                torch.cuda.synchronize()
                ddt = (time.perf_counter() - t1) / 5
                d0, _ = eng.adaptive_depth_batch(am[:1], aa[:1], prec, max_steps=255, gaussian_tail=True)
                g = O.adaptive_groups(af[0][0], prec)
                t1 = time.perf_counter()
                ref = O.depth_from_additional(af[0][0], af[0][1], prec, 255, groups=g)
                cpu_s = time.perf_counter() - t1
                want = O.gaussian_same_size(ref["depth"])
                out["reference_default_flow"] = {
                    "value": round(8 * 1920 * 1080 / ddt / 1e6, 1), "unit": "Mpixels/s",
                    "ms_per_call": round(ddt * 1e3, 4),
                    "workload": "8 x 1920x1080 = the reference's pair mirror-tiled: stepth_adaptive_groups_batch "
                                "(min 16 splits) -> ring first-hit per group, max_steps 255 -> normalise -> paint -> "
                                "same-size Gaussian resample (src/depth_image.rs:101-135; grouping and resample "
                                "parity unpinned, DESIGN.md)",
                    "groups_frame0": int(len(g)),
                    "parity_frame0_bit_exact": bool(np.array_equal(d0[0].cpu().numpy(), want)),
                    "cpu_ms_frame0_match_only": round(cpu_s * 1e3, 1),
                    "cpu_sample": f"oracle PASS 1 + 2 on the oracle's own groups of frame 0, {O.usable_cpus()} cores"}
                del am, aa, ao
            # -- host-resident frames through the pipelined host batch (the
            # end-to-end, PCIe-inclusive rate; never `value`)
            n_host = 32
            hm = [frames_np[i % n_frames][0] for i in range(n_host)]
            ha = [frames_np[i % n_frames][1] for i in range(n_host)]
            ho = [np.zeros((H, W), np.uint8) for _ in range(n_host)]
            eng.reset_stream()
            eng.depth_from_additional_batch(hm[:8], ha[:8], prec, max_steps=args.d_max, outs=ho[:8])
            t1 = time.perf_counter()
            eng.depth_from_additional_batch(hm, ha, prec, max_steps=args.d_max, outs=ho)
            t_page = time.perf_counter() - t1
            pm = [eng.host_array((H, W, 3)) for _ in range(n_host)]
            pa = [eng.host_array((H, W, 3)) for _ in range(n_host)]
            po = [eng.host_array((H, W)) for _ in range(n_host)]
            for f in range(n_host):
                pm[f][...] = hm[f]
                pa[f][...] = ha[f]
            eng.depth_from_additional_batch(pm[:8], pa[:8], prec, max_steps=args.d_max, outs=po[:8])
            t1 = time.perf_counter()
            eng.depth_from_additional_batch(pm, pa, prec, max_steps=args.d_max, outs=po)
            t_pin = time.perf_counter() - t1
            dev0, _ = eng.depth_from_additional_batch(mains_t[:1], adds_t[:1], prec, max_steps=args.d_max)
            e2e_ok = all(np.array_equal(ho[f], po[f]) for f in range(n_host)) and \
                np.array_equal(ho[0], dev0[0].cpu().numpy())
            out["end_to_end"] = {
                "value_pageable": round(n_host * W * H / t_page / 1e6, 1),
                "value_page_locked": round(n_host * W * H / t_pin / 1e6, 1), "unit": "Mpixels/s",
                "workload": f"{n_host} host-resident 1920x1080 pairs through the pipelined host batch: "
                            "H2D of the main image and of the third of the additional one a search can "
                            "reach, kernels, D2H of the depth map (10.5 of a frame's 14.5 MB travel)",
                "matches_device_path": bool(e2e_ok)}
            del pm, pa, po, hm, ha, ho
            eng.use_torch_stream()
            # -- the engine's second mode on the same frames (north_star's literal windowed SAD;
            # no counterpart in the reference)
            sworker = EngineStripWorker(eng, prec, mode=MODE_SAD_WINDOW, d_max=args.d_max, win=args.win)
            sharded_depth(sworker, frames, H, W)
            torch.cuda.synchronize()
            eng.kernel_time_enable(True)
            t1 = time.perf_counter()
            for _ in range(3):
                sharded_depth(sworker, frames, H, W)
            torch.cuda.synchronize()
            sdt = (time.perf_counter() - t1) / 3
            sms, sl = eng.kernel_time_read()
            eng.kernel_time_enable(False)
            out["sad_mode"] = {
                "value": round(px_step / sdt / 1e6, 3), "unit": "Mpixels/s",
                "workload": f"windowed SAD {args.win}x{args.win}, d_max={args.d_max}, same frames",
                "kernel": "sad_search_kernel", "kernel_avg_ms": round(sms / max(sl, 1), 4),
                "frames_per_launch": min(n_frames, 64),
                "hbm_frac": round(ALGO_BYTES_PER_PIXEL * W * H * min(n_frames, 64)
                                  / (sms / max(sl, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)}
            # -- the reference's complete output: depth map + same-size Gaussian resample
            # (src/depth_image.rs:133-134; image 0.23.14's sampler restated, parity unpinned)
            eng.depth_from_additional_batch(mains_t, adds_t, prec, max_steps=args.d_max,
                                            gaussian_tail=True, outs=outs_t, asynchronous=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                eng.depth_from_additional_batch(mains_t, adds_t, prec, max_steps=args.d_max,
                                                gaussian_tail=True, outs=outs_t, asynchronous=True)
            torch.cuda.synchronize()
            gdt = (time.perf_counter() - t1) / 3
            out["with_gaussian_tail"] = {
                "value": round(px_step / gdt / 1e6, 1), "unit": "Mpixels/s",
                "workload": "same frames, depth map + same-size Gaussian resample, one stream"}
            # -- README's second half on the depth maps of the last step (Readme.md:18-24):
            # select_foreground = histogram + depth_split(2) + slice, three launches for all
            # frames; 1 B read twice + 1 B written per pixel
            planes = [depth[f] for f in range(n_frames)]
            dl = eng.plane_list(planes, (1,), "depth")
            ml = eng.plane_list([torch.empty_like(p) for p in planes], (1,), "mask")
            eng.select_foreground_batch(dl, ml, asynchronous=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                eng.select_foreground_batch(dl, ml, asynchronous=True)
            e1.record()
            torch.cuda.synchronize()
            fms = e0.elapsed_time(e1) / 10
            out["foreground_tail"] = {
                "value": round(px_step / (fms * 1e-3) / 1e6, 1), "unit": "Mpixels/s",
                "workload": f"select_foreground (depth_split(2) + slice) on the step's {n_frames} depth maps",
                "ms": round(fms, 4), "algorithmic_bytes_per_pixel": 3,
                "hbm_frac": round(3 * px_step / (fms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if O is not None:
            base, ref_depth = cpu_baseline(O, sets_np[last_set][0], args)
            out["cpu_baseline"] = base
            got = depth[0].cpu().numpy()
            if args.mode == "ring":
                out["parity_frame0_bit_exact"] = bool(np.array_equal(got, ref_depth))
            else:
                mx = int(_mx[0].item())
                exp = (ref_depth.astype(np.uint32) * 255 // max(mx, 1)).astype(np.uint8)
                out["parity_frame0_bit_exact"] = bool(np.array_equal(got[: exp.shape[0]], exp))
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
