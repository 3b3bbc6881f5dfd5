"""the windowed-SAD mode on BASELINE config 3 read literally (4 x 3840x2160, 5x5, d_max 128), a few calls"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from stepth_amd import MODE_SAD_WINDOW, Engine
from stepth_amd.engine import PlaneList
frames = bench.make_frames(3840, 2160, 128, 4, bases=1)
dev = torch.device("cuda", 0)
eng = Engine(0)
eng.use_torch_stream()
mains = [torch.from_numpy(l).to(dev) for l, _ in frames]
adds = [torch.from_numpy(r).to(dev) for _, r in frames]
outs = [torch.empty(l.shape[:2], dtype=torch.uint8, device=dev) for l, _ in frames]
pm, pa, po = PlaneList(mains, (3, 4), "main"), PlaneList(adds, (3,), "additional"), PlaneList(outs, (1,), "depth")
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 36  # 0: no pixel is ever decided by the threshold (every wave runs all its chunks)
call = lambda: eng.depth_from_additional_batch(pm, pa, [prec] * 3, mode=MODE_SAD_WINDOW, d_max=128, win=5, outs=po, asynchronous=True)
for _ in range(2):
    call()
torch.cuda.synchronize()
eng.kernel_time_enable(True)
t0 = time.perf_counter()
for _ in range(6):
    call()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 6
kms, kl = eng.kernel_time_read()
print(f"sad 4x4K 5x5 d128 prec {prec}: call_ms {dt * 1e3:.4f} search_kernel_ms {kms / kl:.4f}")
