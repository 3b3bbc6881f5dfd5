"""PCIe-inclusive throughput of the path for callers that hold images in host memory
(the reference's DepthImage does, src/depth_image.rs:8-9): one synchronous call per frame vs
the pipelined host batch (stepth_depth_from_additional_batch on STEPTH_MEM_HOST images),
pageable and page-locked, beside the device-resident batch.  1920x1080, d_max 128."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from stepth_amd import Engine, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W, H, D = 1920, 1080, 128
eng = Engine(0)
base = [synth.stereo_pair(W, H, D, seed=0x5EED + i)[:2] for i in range(4)]
mains = [np.ascontiguousarray(base[i % 4][0] if i < 4 else base[i % 4][0][::-1]) for i in range(N)]
adds = [np.ascontiguousarray(base[i % 4][1] if i < 4 else base[i % 4][1][::-1]) for i in range(N)]
px = N * W * H
prec = [36] * 3


def timed(fn, reps=3):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) / reps


dm = [torch.from_numpy(m).cuda() for m in mains]
da = [torch.from_numpy(a).cuda() for a in adds]
do = [torch.empty((H, W), dtype=torch.uint8, device="cuda") for _ in range(N)]
t = timed(lambda: eng.depth_from_additional_batch(dm, da, prec, max_steps=D, outs=do))
print(f"device-resident batch of {N}:      {t * 1e3:8.2f} ms  {px / t / 1e6:8.1f} Mpx/s")
ref = [o.cpu().numpy() for o in do]

outs = [np.zeros((H, W), np.uint8) for _ in range(N)]
t = timed(lambda: [eng.depth_from_additional(mains[f], adds[f], prec, max_steps=D, out=outs[f]) for f in range(N)], 1)
print(f"host, one call per frame:          {t * 1e3:8.2f} ms  {px / t / 1e6:8.1f} Mpx/s")
assert all(np.array_equal(outs[f], ref[f]) for f in range(N))

outs = [np.zeros((H, W), np.uint8) for _ in range(N)]
t = timed(lambda: eng.depth_from_additional_batch(mains, adds, prec, max_steps=D, outs=outs))
print(f"host batch, pageable memory:       {t * 1e3:8.2f} ms  {px / t / 1e6:8.1f} Mpx/s")
assert all(np.array_equal(outs[f], ref[f]) for f in range(N))

pm = [eng.host_array((H, W, 3)) for _ in range(N)]
pa = [eng.host_array((H, W, 3)) for _ in range(N)]
po = [eng.host_array((H, W)) for _ in range(N)]
for f in range(N):
    pm[f][...] = mains[f]
    pa[f][...] = adds[f]
t = timed(lambda: eng.depth_from_additional_batch(pm, pa, prec, max_steps=D, outs=po))
print(f"host batch, page-locked memory:    {t * 1e3:8.2f} ms  {px / t / 1e6:8.1f} Mpx/s")
assert all(np.array_equal(po[f], ref[f]) for f in range(N))
