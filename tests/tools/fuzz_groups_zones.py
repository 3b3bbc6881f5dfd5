import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np
from stepth_amd import Engine, SplitError
from oracle import oracle as O
eng=Engine(0); rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 3)
budget=float(sys.argv[2] if len(sys.argv)>2 else 60)
t0=time.time(); n=bad=0
while time.time()-t0 < budget:
    n+=1
    w,h=int(rng.integers(1,260)),int(rng.integers(1,160))
    kind=rng.integers(0,3)
    img = rng.integers(0,256,(h,w,3),dtype=np.uint8) if kind==0 else (rng.integers(0,256,(h,w,3))//int(rng.choice([16,32,64]))*int(rng.choice([16,32,60]))).clip(0,255).astype(np.uint8) if kind==1 else np.repeat(np.repeat(rng.integers(0,256,((h+7)//8,(w+7)//8,3),dtype=np.uint8),8,0),8,1)[:h,:w].copy()
    prec=[int(v) for v in rng.integers(0,90,3)]; mn=int(rng.integers(0,20)); mx=int(rng.choice([0,0,0,3,10,17,25]))
    a=O.adaptive_groups(img,prec,mn,mx); b=eng.adaptive_groups(img,prec,mn,mx)
    ok=len(a)==len(b) and np.array_equal(np.sort(a,order=["y","x"]),np.sort(b,order=["y","x"]))
    d=img[...,0].copy(); zones=int(rng.integers(0,12))
    ref=O.depth_split(d,zones)
    try: got=(0,eng.depth_split(d,zones))
    except SplitError as e: got=(e.status,[])
    ok = ok and got==(ref["status"],ref["zones"])
    if not ok: bad+=1; print("MISMATCH",n,w,h,prec,mn,mx,zones,flush=True)
print(f"groups/zones fuzz: {n} cases, {bad} mismatches"); sys.exit(1 if bad else 0)
