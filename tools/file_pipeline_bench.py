"""End-to-end throughput from image FILES (JPEG, 1920x1080): the reference's per-pair flow
(decode, then compute: DepthImage.open + open_depth_from_additional) against the file pipeline
(stepth_amd/pipeline.py: decode threads || batched engine calls).  Decode is CPU work
(Pillow / libjpeg-turbo); everything after the decoded bytes runs in the HIP engine."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image

from stepth_amd import DepthImage, Engine, synth
from stepth_amd.pipeline import depth_from_files

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W, H = 1920, 1080
tmp = tempfile.mkdtemp(prefix="stepth_files_")
base = [synth.stereo_pair(W, H, 128, seed=0x5EED + i)[:2] for i in range(2)]
pairs = []
for i in range(N):
    l, r = base[i % 2]
    if (i // 2) % 2:
        l, r = l[::-1], r[::-1]
    pm, pa = os.path.join(tmp, f"m{i}.jpg"), os.path.join(tmp, f"a{i}.jpg")
    Image.fromarray(np.ascontiguousarray(l)).save(pm, quality=90)
    Image.fromarray(np.ascontiguousarray(r)).save(pa, quality=90)
    pairs.append((pm, pa))
eng = Engine(0)
prec = [255 // 7] * 3
px = N * W * H

t = time.perf_counter()
for pm, pa in pairs[:8]:
    img = DepthImage.open(pm)
    img.open_depth_from_additional(pa, prec, engine=eng)
dt = (time.perf_counter() - t) / 8
print(f"one pair at a time (decode, then compute): {dt * 1e3:7.1f} ms/pair  {W * H / dt / 1e6:7.1f} Mpx/s")
for workers in (1, 4, 16):
    list(depth_from_files(pairs[:8], prec, engine=eng, workers=workers))
    t = time.perf_counter()
    n = sum(1 for _ in depth_from_files(pairs, prec, engine=eng, workers=workers, chunk=8))
    dt = time.perf_counter() - t
    print(f"file pipeline, {workers:2d} decode threads: {dt / n * 1e3:7.1f} ms/pair  {px / dt / 1e6:7.1f} Mpx/s")
t = time.perf_counter()
for pm, pa in pairs[:8]:
    Image.open(pm).convert("RGB").load()
    Image.open(pa).convert("RGB").load()
print(f"decode alone, one thread: {(time.perf_counter() - t) / 8 * 1e3:.1f} ms/pair")
