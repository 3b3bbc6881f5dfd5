"""kernel time of one leg of the dense search: leg_time.py bench|asset [max_steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from stepth_amd import Engine
from stepth_amd.engine import PlaneList
which, ms = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 128
hf = len(sys.argv) > 3 and sys.argv[3] == "hf"
frames = bench.make_frames(1920, 1080, 128, 8) if which == "bench" else bench.asset_frames(1920, 1080, 8)
dev = torch.device("cuda", 0)
eng = Engine(0)
eng.use_torch_stream()
mains = [torch.from_numpy(l).to(dev) for l, _ in frames]
adds = [torch.from_numpy(r).to(dev) for _, r in frames]
outs = [torch.empty(l.shape[:2], dtype=torch.uint8, device=dev) for l, _ in frames]
pm, pa, po = PlaneList(mains, (3, 4), "main"), PlaneList(adds, (3,), "additional"), PlaneList(outs, (1,), "depth")
call = lambda: eng.depth_from_additional_batch(pm, pa, [36] * 3, max_steps=ms, outs=po, asynchronous=True, heavy_first=hf)
for _ in range(3):
    call()
torch.cuda.synchronize()
import time
eng.kernel_time_enable(True)
t0 = time.perf_counter()
for _ in range(10):
    call()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
kms, kl = eng.kernel_time_read()
import zlib
print(f"call_ms {dt * 1e3:.4f} kernel_ms {kms / kl:.4f} crc {zlib.crc32(outs[3].cpu().numpy().tobytes()):08x}")
