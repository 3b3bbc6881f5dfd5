"""The path the way the reference runs it: adaptive groups (the engine's stand-in for disage,
(16, ceil(log2(W*H))) splits) -> ring first-hit per group -> normalise -> paint, 1920x1080,
d_max 128, inputs resident in HBM; beside the oracle on the same groups."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from stepth_amd import Engine, synth

W, H, D = 1920, 1080, 128
eng = Engine(0)
eng.use_torch_stream()
frames = [synth.stereo_pair(W, H, D, seed=0x5EED + i)[:2] for i in range(4)]
dev = [(torch.from_numpy(np.ascontiguousarray(l)).cuda(), torch.from_numpy(np.ascontiguousarray(r)).cuda()) for l, r in frames]
prec = [36] * 3


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


groups = [eng.adaptive_groups(l, prec) for l, _ in dev]
print("groups per frame:", [len(g) for g in groups])
tg = timed(lambda: [eng.adaptive_groups(l, prec) for l, _ in dev]) / len(dev)
td = timed(lambda: [eng.depth_from_additional(l, r, prec, max_steps=D, groups=g) for (l, r), g in zip(dev, groups)]) / len(dev)
t1 = timed(lambda: [eng.depth_from_additional(l, r, prec, max_steps=D) for l, r in dev]) / len(dev)
print(f"adaptive grouping   {tg * 1e3:7.3f} ms/frame")
print(f"depth from groups   {td * 1e3:7.3f} ms/frame  -> grouping + depth {W * H / (tg + td) / 1e6:8.1f} Mpx/s")
print(f"depth, 1x1 groups   {t1 * 1e3:7.3f} ms/frame  (frame-by-frame calls) {W * H / t1 / 1e6:8.1f} Mpx/s")
if "--cpu" in sys.argv:
    from oracle import oracle as O
    O.set_threads(O.usable_cpus())
    l, r = frames[0]
    t = time.perf_counter()
    g = O.adaptive_groups(l, prec)
    ref = O.depth_from_additional(l, r, prec, D, groups=g)
    dt = time.perf_counter() - t
    got, _ = eng.depth_from_additional(dev[0][0], dev[0][1], prec, max_steps=D, groups=groups[0])
    print(f"oracle, {O.usable_cpus()} cores: {dt * 1e3:.1f} ms/frame {W * H / dt / 1e6:.2f} Mpx/s; GPU == oracle: {np.array_equal(got.cpu().numpy(), ref['depth'])}")
