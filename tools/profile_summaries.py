"""Turns the rocprofv3 outputs of tools/run_profiles.sh into the files kept under profiles/.

    gpurun -- 'bash tools/run_profiles.sh'          (on the GPU box; outputs under gpurun_out/prof_r3)
    python tools/profile_summaries.py gpurun_out/prof_r3 profiles r3

Writes <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the headline command, stepth
kernels), <tag>_kernel_stats_all_legs.csv (the same for the other profiled legs),
<tag>_hbm_traffic.json (FETCH_SIZE / WRITE_SIZE per launch, gfx950-corrected, with the hash of the
kernel sources they were measured on -- bench.py only quotes `traffic` while that hash is
current) and <tag>_sq_counters.txt.

Per-launch means are taken over the launches of the HEADLINE STEP only: the profiled command also
runs bench.py's statistics leg (one-frame launches of pack / paint and the STATS variant of the
search), which would otherwise be averaged in -- a kernel's headline launches are those of its
largest grid, `<true>` (STATS) variants are dropped, and every kernel of the step must show
exactly steps + warmup launches (`calls`; tests/test_bench_contract.py asserts it).
"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ["tile_kernels.h", "tile_lane.h", "cell_pack.h", "ring_kernels.h", "engine.hip"]
DOMINANT = "ring_tile_kernel<false, false>"
STEPS, WARMUP = 6, 2
CMD = f"python bench.py --steps {STEPS} --warmup {WARMUP} --no-cpu-baseline --no-extra-legs"
LEGS = {  # directory under the profile root -> (what ran, the kernel the leg is about)
    "asset": ("python tools/leg_time.py asset 128 hf  (8 x 1080p natural frames, STEPTH_FLAG_HEAVY_FIRST: order + pooled probes, chosen on the device)", "ring_tile_auto_kernel"),
    "adaptive": ("python tools/leg_adaptive.py bench  (adaptive flow, 8 x 1080p synthetic)", "ring_list_kernel<false>"),
    "adaptive_asset": ("python tools/leg_adaptive.py asset  (adaptive flow, 8 x 1080p natural)", "ring_list_kernel<false>"),
    "sad": ("python tools/leg_sad.py  (windowed SAD 5x5, d_max 128, 4 x 3840x2160)", "sad_search_kernel<5, 16, 16>"),
}


def kernel_source_sha16():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "stepth_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*\)$", "", name)


def find(root, d, suffix):
    hits = glob.glob(os.path.join(root, d, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit(f"missing {root}/{d}/*{suffix}")
    return hits[0]


def counters(root, d):
    """{(kernel, grid): {counter: [values per launch]}}"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(find(root, d, "counter_collection.csv"))):
        if "stepth::" in r["Kernel_Name"]:
            agg[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def headline_groups(agg):
    """per kernel the launches of its largest grid; STATS variants dropped"""
    best = {}
    for (k, grid) in agg:
        if "<true" in k:
            continue
        if k not in best or grid > best[k]:
            best[k] = grid
    return {k: agg[(k, g)] for k, g in best.items()}, best


def mean(v):
    return sum(v) / len(v)


def traffic_of(root):
    fetch, grids = headline_groups(counters(root, "prof_fetch"))
    write, _ = headline_groups(counters(root, "prof_write"))
    out = {}
    for k in fetch:
        fv, wv = fetch[k]["FETCH_SIZE"], write.get(k, {}).get("WRITE_SIZE", [0.0])
        out[k] = {"calls": len(fv), "grid_threads": grids[k],
                  "FETCH_SIZE_KB_per_launch": round(mean(fv), 1), "WRITE_SIZE_KB_per_launch": round(mean(wv), 1),
                  "corrected_bytes_per_launch": int((2 * mean(fv) + mean(wv)) * 1024)}
    return out


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    # 1. kernel-trace stats of the headline command: stepth kernels only
    rows = list(csv.reader(open(find(src, "prof_kt", "kernel_stats.csv"))))
    keep = [rows[0]] + [r for r in rows[1:] if "stepth::" in r[0]]
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- {CMD} ; stepth kernels only (all launches of the command: the\n"
                "# statistics leg's one-frame launches included -- per-step means are in the traffic JSON)\n")
        csv.writer(f, quoting=csv.QUOTE_ALL).writerows(keep)
    # headline-step launch durations from the trace itself (largest grid per kernel)
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(find(src, "prof_kt", "kernel_trace.csv"))):
        if "stepth::" in r["Kernel_Name"]:
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            dur[(short(r["Kernel_Name"]), grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)

    # 2. HBM traffic per launch of the headline step (FETCH_SIZE / WRITE_SIZE are in KB)
    kern = traffic_of(src)
    for k, v in kern.items():
        us = dur.get((k, v["grid_threads"]), [])
        v["avg_us_kernel_trace"] = round(mean(us), 2) if us else None
    dom = next(k for k in kern if DOMINANT in k)
    step_kernels = sorted(kern)
    step = sum(v["corrected_bytes_per_launch"] for v in kern.values())
    sq, _ = headline_groups({**counters(src, "prof_sq")})
    sq2, _ = headline_groups({**counters(src, "prof_sq2")})
    for k, v in sq2.items():
        sq.setdefault(k, {}).update(v)
    dom_sq = {c: round(mean(v)) for c, v in sorted(sq.get(dom, {}).items())}
    legs = {}
    for d, (what, kernel) in LEGS.items():
        root = os.path.join(src, d)
        if not os.path.isdir(root):
            continue
        tk = traffic_of(root)
        k = next((x for x in tk if kernel in x), None)
        if k:
            legs[d] = {"command": what, "kernel": k, **tk[k]}
    algo = 7 * 8 * 1920 * 1080
    json.dump({
        "_about": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, as MI355X_MICROARCH.md "
                  f"prescribes) on `{CMD}`; means over the launches of the HEADLINE STEP (the largest grid of each "
                  "kernel: one batch of 8 frames 1920x1080; the one-frame launches of bench.py's statistics leg and "
                  "the STATS variant of the search are left out; `calls` = steps + warmup for every kernel).  "
                  "gfx950 correction per the guide's HBM section: FETCH_SIZE reports 1/2 of the bytes of coalesced "
                  "dword streams (checked in round 1 on paint_1x1_vec4_kernel: 2025 KB read -> FETCH_SIZE 1043 KB, "
                  "WRITE_SIZE exact), so corrected traffic = 2 x FETCH_SIZE + WRITE_SIZE.  These counters count "
                  "requests that leave the L2: Infinity-Cache hits are NOT excluded -- the headline cycles ONE frame "
                  "set (100 MB of inputs and tables, inside the 256 MB Infinity Cache), so part of this traffic is "
                  "fabric traffic that never reaches the HBM stacks.",
        "frames_per_launch": 8, "algorithmic_bytes_per_launch": algo,
        "steps_plus_warmup": STEPS + WARMUP,
        "kernel_source_sha16": kernel_source_sha16(), "kernel_sources": KERNEL_SOURCES,
        "dominant_kernel": dom,
        "dominant_kernel_traffic_bytes_per_launch_corrected": kern[dom]["corrected_bytes_per_launch"],
        "dominant_kernel_sq_counters_per_launch": dom_sq,
        "headline_step_kernels": step_kernels,
        "headline_step_traffic_bytes_corrected": step,
        "headline_step_traffic_over_algorithmic": round(step / algo, 2),
        "kernels": kern,
        "other_legs_dominant_kernel": legs,
    }, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)

    # 3. SQ counters: the headline step, then the other legs (tools/sq_summary.py's summaries)
    with open(os.path.join(dst, f"{tag}_sq_counters.txt"), "w") as f:
        f.write(f"# rocprofv3 --pmc (SQ counters, two passes); per launch of the headline step = one batch of 8 frames 1920x1080\n"
                f"# command: {CMD}\n"
                "# units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves "
                "(MI355X_MICROARCH.md); means per launch\n")
        for k in sorted(sq):
            for c in sorted(sq[k]):
                f.write(f"{k:58s} {c:24s} {mean(sq[k][c]):16.0f}\n")
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import sq_summary
        for d, (what, _kernel) in LEGS.items():
            root = os.path.join(src, d)
            if os.path.isdir(root):
                f.write(f"\n# ---- leg: {what}\n# (per kernel and grid: launches and mean time from the kernel trace, mean counters per launch;\n"
                        "#  valu_busy = SQ_ACTIVE_INST_VALU / SIMD quad-cycles of the launch, waves_per_simd = SQ_WAVE_CYCLES / the same)\n")
                f.write("\n".join(sq_summary.main(root)) + "\n")
    # 4. kernel stats of the other legs
    with open(os.path.join(dst, f"{tag}_kernel_stats_all_legs.csv"), "w", newline="") as f:
        wr = csv.writer(f, quoting=csv.QUOTE_ALL)
        for d, (what, _kernel) in LEGS.items():
            root = os.path.join(src, d)
            if not os.path.isdir(root):
                continue
            rows = list(csv.reader(open(find(root, "prof_kt", "kernel_stats.csv"))))
            f.write(f"# rocprofv3 --kernel-trace --stats -- {what}\n")
            wr.writerows([rows[0]] + [r for r in rows[1:] if "stepth::" in r[0]])
    print("wrote", tag, "summaries; dominant kernel traffic/launch", kern[dom]["corrected_bytes_per_launch"],
          "step", step, f"= {step / algo:.2f} x algorithmic; calls", {k: v["calls"] for k, v in kern.items()})


if __name__ == "__main__":
    main()
