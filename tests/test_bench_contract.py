"""CPU-side guards of what bench.py reports (no GPU, no bench run):

* `roofline.traffic` is quoted from profiles/r3_hbm_traffic.json; that file names the kernels it
  was measured on and the hash of their sources.  If the kernels of the headline step are
  renamed / replaced, or the dominant kernel changes, this fails -- the profile cannot go stale
  silently; and bench.py itself drops `traffic` (null) as soon as the source hash differs.
* the JSON line's vocabulary stays the contract's (metric name, workload config, no model keys).
"""
import importlib.util
import json
import os
import re

from conftest import ROOT

CSRC = os.path.join(ROOT, "stepth_amd", "csrc")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    src = open(os.path.join(ROOT, "bench.py")).read()
    return spec, src


def _launched_in(function_names):
    """kernels launched (hipLaunchKernelGGL) inside the given functions of engine.hip"""
    text = open(os.path.join(CSRC, "engine.hip")).read()
    found = set()
    for fn in function_names:
        m = re.search(r"\b" + fn + r"\s*\([^;{]*\)\s*\{", text)
        assert m, f"{fn} not found in engine.hip"
        depth, i = 0, m.end() - 1
        while True:
            depth += text[i] == "{"
            depth -= text[i] == "}"
            i += 1
            if depth == 0:
                break
        body = text[m.start():i]
        found |= {re.sub(r"[<(].*", "", k) for k in re.findall(r"hipLaunchKernelGGL\(\(?([A-Za-z0-9_]+)", body)}
    return found


def test_traffic_profile_names_the_kernels_the_headline_step_launches():
    _spec, bench_src = _bench()
    dominant = re.search(r'DOMINANT_KERNEL = "([a-z0-9_]+)"', bench_src).group(1)
    prof = json.load(open(os.path.join(ROOT, "profiles", "r3_hbm_traffic.json")))
    assert dominant in prof["dominant_kernel"]
    profiled = {re.sub(r"^stepth::", "", re.sub(r"<.*", "", k)) for k in prof["headline_step_kernels"]}
    # the one-group-per-pixel batch: pack (+ superblock fold), the dense search, paint; the
    # sharded flow bench.py drives normalises the strip's rows in a launch of its own
    launched = _launched_in(["pack_additional_tiles", "launch_tile"])
    # (the heavy-first order's three kernels and the two-variant search run only under STEPTH_FLAG_HEAVY_FIRST: not in the headline)
    launched -= {"tile_cost_kernel", "tile_scan_kernel", "tile_scatter_kernel", "ring_tile_auto_kernel"}
    assert launched == {"tile_pack_kernel", "tile_super_kernel", "ring_tile_kernel"}, launched
    assert launched <= profiled, f"profile lacks {launched - profiled}: re-run tools/run_profiles.sh"
    # per-launch means are over the headline step's launches only: every kernel of the step shows
    # exactly steps + warmup calls (the statistics leg's one-frame launches are left out), and the
    # step's traffic is the sum of exactly those kernels
    assert {v["calls"] for v in prof["kernels"].values()} == {prof["steps_plus_warmup"]}, prof["kernels"]
    assert not any("<true" in k for k in prof["kernels"]), "STATS variants do not belong to the headline step"
    assert prof["headline_step_traffic_bytes_corrected"] == sum(v["corrected_bytes_per_launch"] for v in prof["kernels"].values())
    assert abs(prof["headline_step_traffic_over_algorithmic"] -
               prof["headline_step_traffic_bytes_corrected"] / prof["algorithmic_bytes_per_launch"]) < 0.01
    for leg in ("asset", "adaptive", "sad"):  # the PMC passes of the other dominant kernels exist
        assert leg in prof["other_legs_dominant_kernel"], leg
    assert profiled - launched <= {"paint_1x1_vec4_kernel", "paint_tiles_kernel", "normalise_rows_kernel"}, profiled
    for k in profiled:  # every profiled kernel still exists in the sources
        assert any(re.search(r"\b" + k + r"\b", open(os.path.join(CSRC, f)).read())
                   for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))), k
    # the dominant kernel's traffic must stay below the whole step's and above nothing
    assert 0 < prof["dominant_kernel_traffic_bytes_per_launch_corrected"] <= prof["headline_step_traffic_bytes_corrected"]
    assert prof["algorithmic_bytes_per_launch"] == 7 * 8 * 1920 * 1080


def test_bench_quotes_traffic_only_for_current_kernel_sources():
    _spec, bench_src = _bench()
    assert "kernel_source_sha16" in bench_src, "bench.py must compare the profile's source hash"
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import profile_summaries
    prof = json.load(open(os.path.join(ROOT, "profiles", "r3_hbm_traffic.json")))
    assert prof["kernel_sources"] == profile_summaries.KERNEL_SOURCES
    assert re.fullmatch(r"[0-9a-f]{16}", prof["kernel_source_sha16"])


def test_json_vocabulary():
    _spec, bench_src = _bench()
    assert '"Mpixels/sec depth output, 1080p stereo d_max=128"' in bench_src  # BASELINE.json metric
    for key in ('"roofline"', '"cpu_baseline"', '"workload"', '"frames_per_step"', '"two_streams"',
                '"config2_1080p_d64"', '"config3_4k_d128"', '"config5_8k_d256_1gpu"',
                '"asset_tiled_1080p"', '"end_to_end"', '"adaptive_flow"'):
        assert key in bench_src, key
    assert '"model"' not in bench_src and '"seq_len"' not in bench_src
