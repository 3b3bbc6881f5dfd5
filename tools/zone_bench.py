"""Timing of the depth-map consumer kernels (select_foreground_batch, invert, slice, apply_mask)
on 32x1080p / 8K depth maps; the command behind profiles/r1_zone_kernel_stats.csv."""
import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
from stepth_amd import Engine
eng=Engine(0); eng.use_torch_stream()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
g=torch.Generator(device='cuda').manual_seed(1)
def depth_like(h,w,n):
    # depth-map-like: 70% zeros + smooth blobs
    out=[]
    for i in range(n):
        d=torch.randint(0,256,(h,w),dtype=torch.uint8,device='cuda',generator=g)
        m=torch.rand((h//8+1,w//8+1),device='cuda',generator=g).repeat_interleave(8,0).repeat_interleave(8,1)[:h,:w]
        d=torch.where(m<0.7, torch.zeros_like(d), (m*255).to(torch.uint8)); out.append(d.contiguous())
    return out
for name,(h,w,n) in {"1080p x32":(1080,1920,32),"8K x1":(4320,7680,1),"8K x4":(4320,7680,4)}.items():
    for kind in ("blocky","noise"):
        ds=depth_like(h,w,n) if kind=="blocky" else [torch.randint(0,256,(h,w),dtype=torch.uint8,device='cuda',generator=g) for _ in range(n)]
        ms=[torch.empty_like(d) for d in ds]
        px=h*w*n
        t=timeit(lambda: eng.select_foreground_batch(ds,ms,asynchronous=True))
        print(f"{name} {kind}: select_foreground_batch {t*1e3:.1f} us  {px/t/1e6:.1f} Gpx/s  alg 3B/px -> {3*px/t/1e6:.0f} GB/s")
        t=timeit(lambda: [eng.invert_depth(d,asynchronous=True) for d in ds[:1]])
        print(f"   invert 1 frame {t*1e3:.1f} us -> {2*h*w/t/1e6:.0f} GB/s")
        t=timeit(lambda: eng.slice(ds[0],10,200,out=ms[0],asynchronous=True))
        print(f"   slice 1 frame {t*1e3:.1f} us -> {2*h*w/t/1e6:.0f} GB/s")
        rgba=torch.randint(0,256,(h,w,4),dtype=torch.uint8,device='cuda',generator=g)
        t=timeit(lambda: eng.apply_mask(rgba,ms[0],asynchronous=True))
        print(f"   apply_mask 1 frame {t*1e3:.1f} us")
