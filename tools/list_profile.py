"""tuning build only (make CXXFLAGS+=-DSTEPTH_LIST_PROFILE OUT=../libstepth_amd_lp.so, STEPTH_AMD_LIB=...):
how long the waves of ring_list_kernel run and what the long ones do.  list_profile.py bench|asset [max_steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from stepth_amd import Engine
which, ms = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 128
frames = bench.make_frames(1920, 1080, 128, 8) if which == "bench" else bench.asset_frames(1920, 1080, 8)
dev = torch.device("cuda", 0)
eng = Engine(0)
mains = [torch.from_numpy(l).to(dev) for l, _ in frames]
adds = [torch.from_numpy(r).to(dev) for _, r in frames]
eng.adaptive_depth_batch(mains, adds, [36] * 3, max_steps=ms)
eng.stats_enable(True)
eng.adaptive_depth_batch(mains, adds, [36] * 3, max_steps=ms)
raw = eng.stats_raw(32)
eng.stats_enable(False)
waves = sum(raw[8:16])
print(f"{which} d{ms}: waves {waves}, mean {raw[6] / max(waves, 1):.0f} ticks, of which in goes {raw[7] / max(raw[6], 1):.2f}, own-window tests {raw[3] / max(raw[6], 1):.2f}; visits {raw[21] / max(raw[6], 1):.2f}, survey {raw[22] / max(raw[6], 1):.2f}, queueing {raw[20] / max(raw[6], 1):.2f}, before the epilogue {raw[23] / max(raw[6], 1):.2f}; goes {sum(raw[24:32])}")
for b in range(8):
    n = raw[8 + b]
    if n:
        visits = f"{raw[16 + b] / n:6.1f}" if b < 4 else "     -"  # (the words of buckets 4..7 hold the section times)
        print(f"  {2 ** (b + 13):8d} ticks .. : {n:6d} waves, visits/wave {visits}, goes/wave {raw[24 + b] / n:6.1f}")
