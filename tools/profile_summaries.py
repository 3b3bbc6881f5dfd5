"""Turns the rocprofv3 outputs of tools/run_profiles.sh into the files kept under profiles/.

    gpurun -- 'bash tools/run_profiles.sh'          (on the GPU box; outputs under gpurun_out/prof_r2)
    python tools/profile_summaries.py gpurun_out/prof_r2 profiles r2

Writes <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, stepth kernels),
<tag>_hbm_traffic.json (FETCH_SIZE / WRITE_SIZE per launch, gfx950-corrected, with the hash of
the kernel sources they were measured on -- bench.py only quotes `traffic` while that hash is
current) and <tag>_sq_counters.txt.
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
DOMINANT = "ring_tile_kernel<false>"


def kernel_source_sha16():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "stepth_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]

    def find(d, suffix):
        hits = glob.glob(os.path.join(src, d, "**", "*" + suffix), recursive=True)
        if not hits:
            raise SystemExit(f"missing {d}/*{suffix}")
        return hits[0]

    def short(name):
        name = re.sub(r"^void ", "", name)
        return re.sub(r"\(.*\)$", "", name)

    cmd = "python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs"
    # 1. kernel-trace stats: stepth kernels only
    rows = list(csv.reader(open(find("prof_kt", "kernel_stats.csv"))))
    keep = [rows[0]] + [r for r in rows[1:] if "stepth::" in r[0]]
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- {cmd} ; stepth kernels only\n")
        csv.writer(f, quoting=csv.QUOTE_ALL).writerows(keep)

    def counters(d):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
            if "stepth::" in r["Kernel_Name"]:
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        return agg

    # 2. HBM traffic per launch (FETCH_SIZE / WRITE_SIZE are in KB)
    fetch, write = counters("prof_fetch"), counters("prof_write")
    kern = {}
    for k in fetch:
        fv = fetch[k]["FETCH_SIZE"]
        wv = write.get(k, {}).get("WRITE_SIZE", [0.0])
        kern[k] = {"FETCH_SIZE_KB_per_launch": round(sum(fv) / len(fv), 1),
                   "WRITE_SIZE_KB_per_launch": round(sum(wv) / len(wv), 1),
                   "corrected_bytes_per_launch": int((2 * sum(fv) / len(fv) + sum(wv) / len(wv)) * 1024)}
    dom = next(k for k in kern if DOMINANT in k)
    step = sum(v["corrected_bytes_per_launch"] for v in kern.values())
    # SQ counters (two more passes): the dominant kernel's also go into the JSON for bench.py
    sq = counters("prof_sq")
    for k, v in counters("prof_sq2").items():
        sq[k].update(v)
    dom_sq = {c: round(sum(v) / len(v)) for c, v in sorted(sq.get(dom, {}).items())}
    json.dump({
        "_about": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, as MI355X_MICROARCH.md "
                  f"prescribes) on `{cmd}`; means over launches; one launch = one batch of 8 frames 1920x1080. "
                  "gfx950 correction per the guide's HBM section: FETCH_SIZE reports 1/2 of the bytes of coalesced "
                  "dword streams (checked in round 1 on paint_1x1_vec4_kernel: 2025 KB read -> FETCH_SIZE 1043 KB, "
                  "WRITE_SIZE exact), so corrected traffic = 2 x FETCH_SIZE + WRITE_SIZE; Infinity-Cache hits are "
                  "not excluded by these counters.",
        "frames_per_launch": 8, "algorithmic_bytes_per_launch": 7 * 8 * 1920 * 1080,
        "kernel_source_sha16": kernel_source_sha16(), "kernel_sources": KERNEL_SOURCES,
        "dominant_kernel": dom,
        "dominant_kernel_traffic_bytes_per_launch_corrected": kern[dom]["corrected_bytes_per_launch"],
        "dominant_kernel_sq_counters_per_launch": dom_sq,
        "headline_step_kernels": sorted(kern),
        "headline_step_traffic_bytes_corrected": step,
        "kernels": kern,
    }, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)

    # 3. SQ counters
    with open(os.path.join(dst, f"{tag}_sq_counters.txt"), "w") as f:
        f.write(f"# rocprofv3 --pmc (SQ counters, two passes); per launch = one batch of 8 frames 1920x1080\n"
                f"# command: {cmd}\n"
                "# units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves "
                "(MI355X_MICROARCH.md); means per launch\n")
        for k in sorted(sq):
            for c in sorted(sq[k]):
                v = sq[k][c]
                f.write(f"{k:58s} {c:24s} {sum(v) / len(v):16.0f}\n")
    print("wrote", tag, "summaries; dominant kernel traffic/launch",
          kern[dom]["corrected_bytes_per_launch"], "step", step)


if __name__ == "__main__":
    main()
