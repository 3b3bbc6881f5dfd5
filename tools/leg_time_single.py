"""one frame per call: leg_time_single.py W H max_steps [hf]"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from stepth_amd import Engine
W, H, ms = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
hf = len(sys.argv) > 4 and sys.argv[4] == "hf"
l, r = bench.asset_frames(W, H, 1)[0]
dev = torch.device("cuda", 0)
eng = Engine(0)
eng.use_torch_stream()
m, a = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
out = torch.empty((H, W), dtype=torch.uint8, device=dev)
call = lambda: eng.depth_from_additional_batch([m], [a], [36] * 3, max_steps=ms, outs=[out], asynchronous=True, heavy_first=hf)
for _ in range(3):
    call()
torch.cuda.synchronize()
eng.kernel_time_enable(True)
t0 = time.perf_counter()
for _ in range(10):
    call()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
kms, kl = eng.kernel_time_read()
print(f"{W}x{H} d{ms} hf={hf}: call_ms {dt * 1e3:.4f} kernel_ms {kms / kl:.4f} crc {zlib.crc32(out.cpu().numpy().tobytes()):08x}")
