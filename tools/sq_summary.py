"""Summary of tools/run_sq_quick.sh's passes: per stepth kernel and grid size the mean SQ
counters per launch, the launch time from the kernel trace, and the derived busy fractions.
usage: python tools/sq_summary.py gpurun_out/<dir>"""
import collections
import csv
import glob
import sys


def main(out):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ("prof_sq", "prof_sq2"):
        for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "stepth::" in r["Kernel_Name"]:
                    key = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]))
                    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for f in glob.glob(f"{out}/prof_kt/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "stepth::" in r["Kernel_Name"]:
                grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
                key = (r["Kernel_Name"].split("(")[0].replace("void ", ""), grid)
                dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lines = []
    for k in sorted(agg):
        us = dur.get(k, [0])
        mean_us = sum(us) / len(us)
        lines.append(f"== {k[0]} grid {k[1]}: {len(us)} launches, mean {mean_us:.1f} us")
        c = {n: sum(v) / len(v) for n, v in agg[k].items()}
        for n in sorted(c):
            lines.append(f"   {n:24s} {c[n]:16.0f}")
        if mean_us and "SQ_ACTIVE_INST_VALU" in c:
            quad = mean_us * 1e-6 * 2.4e9 / 4 * 1024  # SIMD quad-cycles of the launch (256 CUs x 4 SIMDs)
            lines.append(f"   valu_busy {c['SQ_ACTIVE_INST_VALU'] / quad:.3f}  scalar_busy "
                         f"{c.get('SQ_ACTIVE_INST_SCA', 0) / (quad / 4):.3f}  waves_per_simd "
                         f"{c.get('SQ_WAVE_CYCLES', 0) / quad:.2f}  lanes_per_valu "
                         f"{c.get('SQ_THREAD_CYCLES_VALU', 0) / max(c['SQ_ACTIVE_INST_VALU'], 1):.1f}")
    open(f"{out}/summary.txt", "w").write("\n".join(lines) + "\n")
    return lines


if __name__ == "__main__":
    print("\n".join(main(sys.argv[1])))
