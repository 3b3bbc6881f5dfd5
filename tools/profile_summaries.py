"""Turns the rocprofv3 outputs of the profile passes into the files kept under profiles/.

Passes (run on the GPU box, outputs under gpurun_out/):
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline
  rocprofv3 --pmc FETCH_SIZE  --output-format csv -d gpurun_out/prof_fetch -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE  --output-format csv -d gpurun_out/prof_write -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
            --output-format csv -d gpurun_out/prof_sq -o sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
(counters in their own runs, as MI355X_MICROARCH.md prescribes).  Then, here:
  python tools/profile_summaries.py gpurun_out profiles r1
"""
import collections
import csv
import glob
import json
import os
import re
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def find(d, suffix):
    hits = glob.glob(os.path.join(src, d, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit(f"missing {d}/*{suffix}")
    return hits[0]


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*\)$", "", name)


# 1. kernel-trace stats: stepth kernels only
rows = list(csv.reader(open(find("prof_kt", "kernel_stats.csv"))))
keep = [rows[0]] + [r for r in rows[1:] if "stepth::" in r[0]]
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline ; stepth kernels only\n")
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
               "WRITE_SIZE_KB_per_launch": round(sum(wv) / len(wv), 1)}
search = next(k for k in kern if "ring_search_kernel<false, false>" in k)
corrected = int((2 * kern[search]["FETCH_SIZE_KB_per_launch"] + kern[search]["WRITE_SIZE_KB_per_launch"]) * 1024)
json.dump({
    "_about": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, as MI355X_MICROARCH.md "
              "prescribes) on `python bench.py --steps 2 --warmup 1 --no-cpu-baseline`; means over launches. "
              "One ring_search_kernel<false,false> launch = one batch of 8 frames 1920x1080 (other kernels' "
              "means mix launch shapes of the bench's extra legs). gfx950 correction per the guide's HBM "
              "section: FETCH_SIZE reports 1/2 of the bytes of coalesced dword streams (checked this round on "
              "paint_1x1_vec4_kernel: 2025 KB read -> FETCH_SIZE 1043 KB, WRITE_SIZE exact), so corrected "
              "traffic = 2 x FETCH_SIZE + WRITE_SIZE; Infinity-Cache hits are not excluded by these counters.",
    "frames_per_launch": 8, "kernels": kern,
    "ring_search_kernel_traffic_bytes_per_launch_corrected": corrected,
}, open(os.path.join(dst, f"{tag}_ring_hbm_traffic.json"), "w"), indent=1)

# 3. SQ counters
sq = counters("prof_sq")
with open(os.path.join(dst, f"{tag}_sq_counters.txt"), "w") as f:
    f.write("# rocprofv3 --pmc (SQ counters); ring kernels: per launch = one batch of 8 frames 1920x1080\n"
            "# command: rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY "
            "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline\n"
            "# units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves (MI355X_MICROARCH.md); "
            "means per launch\n")
    for k in sorted(sq):
        for c in sorted(sq[k]):
            v = sq[k][c]
            f.write(f"{k:58s} {c:24s} {sum(v) / len(v):16.0f}\n")
print("wrote", tag, "summaries; search traffic/launch", corrected)
