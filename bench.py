#!/usr/bin/env python3
"""bench.py -- Mpixels/s of depth output for stepth's open_depth_from_additional hot path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched under
torch.distributed.run, one rank per GPU.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json metric: "Mpixels/sec depth output, 1080p stereo d_max=128"):
  a step = one pass of the hot path over a batch of `8 * N` synthetic 1920x1080
  stereo pairs (inputs resident in HBM), reference-exact mode (square-ring first hit,
  /root/reference/src/helpers.rs:9-54), one group per pixel, max_steps = d_max = 128,
  precision [36;3] (Readme.md:14).  All frames of a step go through ONE set of launches
  (stepth_depth_from_additional_batch): pack additional -> ring0 (centre cell + compaction)
  -> ring_search -> ring_finish (distance, max per frame) -> normalise -> depth rows.  With N > 1 every frame is sharded by row strips over the N ranks,
  with one all_reduce(MAX) and one gather of the depth rows to rank 0 per step (RCCL); per-GPU work is
  constant in N (weak scaling).  value = frames * 1920*1080 / step time, whole job.

  --mode sad runs the windowed-SAD mode (5x5, d_max=128) instead; it has no counterpart in the
  reference and is reported under its own metric name.

roofline: HBM-bound by contract (scan/compare, no MFMA).  achieved = algorithmic bytes per
  ring_search launch (7 B per output pixel: 3 B main RGB + 3 B additional RGB + 1 B depth,
  SURVEY.md 8d) / that kernel's mean launch time, measured live with HIP events on the
  engine's stream; peak = 8 TB/s.  The ring search is data dependent, so the mean number of
  probes per pixel is printed next to it.
cpu_baseline: the C restatement of the reference algorithm (oracle/, "port" -- NOT the Rust
  binary) on one frame of the same workload, all host cores with dynamic scheduling, plus the
  reference's own static 8-chunk partition (src/depth_image.rs:111-112) as `ref8_value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The steps alternate over two HIP streams (see main); with the runtime's default of 4 hardware
# queues the two streams sometimes end up on the same queue and serialise, with 8 they do not.
# Must be set before the HIP runtime starts, i.e. before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
ALGO_BYTES_PER_PIXEL = 7.0


def make_frames(width, height, d_max, count):
    """`count` distinct synthetic pairs from two seeded base pairs and their flips."""
    from stepth_amd import synth
    bases = [synth.stereo_pair(width, height, d_max, seed=0x5EED + i)[:2]
             for i in range(min(2, count))]
    flips = [lambda a: a, lambda a: a[:, ::-1], lambda a: a[::-1], lambda a: a[::-1, ::-1]]
    frames = []
    for i in range(count):
        l, r = bases[i % len(bases)]
        f = flips[(i // len(bases)) % 4]
        frames.append((np.ascontiguousarray(f(l)), np.ascontiguousarray(f(r))))
    return frames


def cpu_baseline(frame, args):
    """Times the oracle (test infrastructure; allowed here as the CPU baseline) on one frame."""
    from oracle import oracle as O
    try:  # a host-tuned build of the same C file, kept out of the tree
        lib_path = O.build(force=True, lib_path=os.path.join("/tmp", f"stepth_oracle_native_{os.getpid()}.so"),
                           march="native")
        O._lib = O.lib(lib_path)
    except Exception:
        O.lib()
    O.set_threads(O.usable_cpus())
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
               "probes_per_pixel": round(ref["probes"] / px, 1)}
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
    ap.add_argument("--interleave", type=int, default=0,
                    help="N > 1, ring mode: deal rows in interleaved blocks of this many rows "
                         "(0 = contiguous strips, a few % faster on the bench frames; blocks of 16 "
                         "balance the data-dependent work whatever the image content)")
    ap.add_argument("--streams", type=int, default=2,
                    help="steps alternate over this many engines/HIP streams (1 = strictly one "
                         "step after the other)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: finish every step's gather before the next step starts")
    ap.add_argument("--no-balance", action="store_true",
                    help="N > 1: keep equal row strips instead of work-weighted ones")
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
    from stepth_amd.sharding import (EngineStripWorker, GatherPipeline, StripBalancer, sharded_depth,
                                     strip_bounds)

    W, H = args.width, args.height
    n_frames = args.fpg * world
    frames_np = make_frames(W, H, args.d_max, n_frames)
    frames = [(torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)) for l, r in frames_np]

    # Steps alternate over `--streams` engines, each on its own HIP stream: while one step's
    # search kernel drains, the next step's pack / ring-0 kernels (HBM-bound) and the previous
    # step's finish / paint already run -- two batches in flight, like any streaming pipeline.
    n_slots = max(1, args.streams)
    torch.cuda.synchronize()  # the uploads above
    engines = [Engine(local_rank) for _ in range(n_slots)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_slots)] if n_slots > 1 else [None]
    for e_, s_ in zip(engines, streams):
        if s_ is None:
            e_.use_torch_stream()
        else:
            with torch.cuda.stream(s_):
                e_.use_torch_stream()
    eng = engines[0]
    prec = [args.precision] * 3

    def make_worker(e_):
        if args.mode == "ring":
            return EngineStripWorker(e_, prec, max_steps=args.d_max, tile=(1, 1), mode=MODE_RING_FIRST_HIT)
        return EngineStripWorker(e_, prec, mode=MODE_SAD_WINDOW, d_max=args.d_max, win=args.win)

    workers = [make_worker(e_) for e_ in engines]
    worker = workers[0]

    # N > 1, contiguous strips: the cut points follow the ranks' measured times from step to
    # step (the cost of a row is data dependent); the warm-up steps let them settle
    balancer = StripBalancer(H, world, align=2, lag=n_slots) if world > 1 and not args.interleave \
        and not args.no_balance else None

    # N > 1: the gather of a step's rows to rank 0 runs behind the following kernels; the
    # outstanding gathers are flushed inside the timed region
    pipelines = [GatherPipeline() if world > 1 and not args.interleave and not args.no_overlap
                 else None for _ in range(n_slots)]
    pipeline = pipelines[0]
    step_no = [0]

    def step():
        slot = step_no[0] % n_slots
        step_no[0] += 1

        def run():
            return sharded_depth(workers[slot], frames, H, W, dst=0 if world > 1 else None,
                                 interleave=args.interleave if args.mode == "ring" else 0,
                                 balancer=balancer, pipeline=pipelines[slot])
        if streams[slot] is None:
            return run()
        with torch.cuda.stream(streams[slot]):
            return run()

    def flush():
        """finishes the gathers still in flight, oldest first; -> depth maps of the last step"""
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

    depth = None
    for _ in range(args.warmup):
        depth, _mx = step()
    if pipeline is not None:
        flush()
    fence()
    for e_ in engines:
        e_.kernel_time_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        depth, _mx = step()
    if pipeline is not None:
        depth = flush()  # the last step's rows; part of the timed work
    fence()
    dt = time.perf_counter() - t0
    kms = klaunches = 0
    for e_ in engines:
        k_, l_ = e_.kernel_time_read()
        kms, klaunches = kms + k_, klaunches + l_
        e_.kernel_time_enable(False)
    eng.use_torch_stream()  # the extra legs below run on the default stream
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    # probes/pixel of this rank's strips (data-dependent work), outside the timed region
    probes_pp = None
    if args.mode == "ring":
        r0, r1 = (balancer.bounds if balancer is not None else strip_bounds(H, world))[rank]
        pat = None
        if args.interleave and world > 1:
            r0, r1, pat = 0, H, (args.interleave, world, rank)
        eng.stats_enable(True)
        tot_p = tot_g = 0
        for f in frames[: min(len(frames), 8)]:
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
        px_launch = (strip_rows[1] - strip_rows[0]) * W * frames_per_launch
        k_avg_ms = kms / max(klaunches, 1)
        achieved = ALGO_BYTES_PER_PIXEL * px_launch / (k_avg_ms * 1e-3) / 1e9 if klaunches else 0.0
        # HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE,
        # gfx950-corrected; see profiles/r1_ring_hbm_traffic.json) -- only for the profiled shape
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r1_ring_hbm_traffic.json")
        if args.mode == "ring" and world == 1 and (W, H, args.d_max) == (1920, 1080, 128) \
                and os.path.exists(tpath):
            prof = json.load(open(tpath))
            per_launch = prof.get("ring_search_kernel_traffic_bytes_per_launch_corrected")
            if per_launch and prof.get("frames_per_launch"):  # scale to this run's batch size
                traffic = int(per_launch * frames_per_launch / prof["frames_per_launch"])
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
                "parallelism": ((f"interleaved {args.interleave}-row strips" if args.interleave and args.mode == "ring"
                                 else "contiguous row strips, cut points following measured per-rank times"
                                 if balancer is not None else "contiguous row strips")
                                + f" x{world} + all_reduce(max) + gather of rows to rank 0 (RCCL"
                                + (", overlapped with the next step's kernels)" if pipeline is not None else ")")
                                if world > 1 else "single GPU")
                               + (f"; steps alternate over {n_slots} engines/HIP streams" if n_slots > 1 else ""),
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                "kernel": "ring_search_kernel" if args.mode == "ring" else "sad_search_kernel",
                "kernel_avg_ms": round(k_avg_ms, 4), "kernel_launches": klaunches,
                "frames_per_launch": frames_per_launch,
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_PIXEL * px_launch,
                "probes_per_pixel": None if probes_pp is None else round(probes_pp, 1),
            },
        }
        if args.mode == "ring" and world == 1:
            # the engine's second mode on the same frames (north_star's literal windowed SAD;
            # no counterpart in the reference): a short extra leg, outside the timed region
            sworker = EngineStripWorker(eng, prec, mode=MODE_SAD_WINDOW, d_max=args.d_max,
                                        win=args.win)
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
        if args.mode == "ring" and world == 1:
            # the reference's complete output: depth map + same-size Gaussian resample
            # (src/depth_image.rs:133-134; image 0.23.14's sampler restated, parity unpinned)
            mains_t, adds_t = [f[0] for f in frames], [f[1] for f in frames]
            outs_t = [torch.empty((H, W), dtype=torch.uint8, device=dev) for _ in frames]
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
        if args.mode == "ring" and world == 1:
            # README's second half on the depth maps of the last step (Readme.md:18-24):
            # select_foreground = histogram + depth_split(2) + slice, three launches for all
            # frames; 1 B read twice + 1 B written per pixel.  Outside the timed region.
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
        if not args.no_cpu_baseline:
            base, ref_depth = cpu_baseline(frames_np[0], args)
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
    for e_ in engines:
        e_.close()


if __name__ == "__main__":
    main()
