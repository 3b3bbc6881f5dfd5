"""the adaptive flow (grouping -> list search -> paint) on the bench or asset frames, a few calls"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from stepth_amd import Engine
which, ms = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 128
frames = bench.make_frames(1920, 1080, 128, 8) if which == "bench" else bench.asset_frames(1920, 1080, 8)
dev = torch.device("cuda", 0)
eng = Engine(0)
eng.use_torch_stream()
mains = [torch.from_numpy(l).to(dev) for l, _ in frames]
adds = [torch.from_numpy(r).to(dev) for _, r in frames]
outs = [torch.empty(l.shape[:2], dtype=torch.uint8, device=dev) for l, _ in frames]
call = lambda: eng.adaptive_depth_batch(mains, adds, [36] * 3, max_steps=ms, outs=outs, asynchronous=True)
for _ in range(3):
    call()
torch.cuda.synchronize()
eng.kernel_time_enable(True)
t0 = time.perf_counter()
for _ in range(8):
    call()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 8
kms, kl = eng.kernel_time_read()
print(f"adaptive {which} d{ms}: call_ms {dt * 1e3:.4f} list_kernel_ms {kms / max(kl, 1):.4f} crc {zlib.crc32(outs[3].cpu().numpy().tobytes()):08x}")
