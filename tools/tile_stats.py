"""Wave-level counters and kernel time of the dense search on the bench frames and on the
reference's asset pair (tuning aid; STEPTH_AMD_LIB selects the build)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from stepth_amd import Engine
from stepth_amd.engine import PlaneList

def leg(eng, frames, ms, label, reps=5):
    dev = torch.device("cuda", 0)
    mains = [torch.from_numpy(l).to(dev) for l, _ in frames]
    adds = [torch.from_numpy(r).to(dev) for _, r in frames]
    h, w = frames[0][0].shape[:2]
    outs = [torch.empty((h, w), dtype=torch.uint8, device=dev) for _ in frames]
    pm, pa, po = PlaneList(mains, (3, 4), "main"), PlaneList(adds, (3,), "additional"), PlaneList(outs, (1,), "depth")
    call = lambda: eng.depth_from_additional_batch(pm, pa, [36] * 3, max_steps=ms, outs=po, asynchronous=True)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    eng.kernel_time_enable(True)
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    kms, kl = eng.kernel_time_read()
    eng.kernel_time_enable(False)
    eng.stats_enable(True)
    call()
    st = eng.stats()
    raw = eng.stats_raw(32)
    eng.stats_enable(False)
    px = len(frames) * w * h
    waves = px / 64
    print(json.dumps({"leg": label, "kernel_ms": round(kms / kl, 4), "waves": int(waves),
                      "step0_waves": st["near_steps"], "far_waves": st["survey_steps"],
                      "superblock_steps": st["superblock_steps"], "blocks_probed_pp": round(st["blocks_probed"] / px, 3),
                      "items_pp": round(st["probes"] / px, 1), "raw8_15": raw[8:16]}))

if __name__ == "__main__":
    eng = Engine(0)
    eng.use_torch_stream()
    leg(eng, bench.make_frames(1920, 1080, 128, 8), 128, "bench 8x1080p d128")
    leg(eng, bench.asset_frames(1920, 1080, 8), 128, "asset 8x1080p d128")
    leg(eng, bench.asset_frames(1920, 1080, 8), 255, "asset 8x1080p d255")
