import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np, torch
from stepth_amd import Engine, MODE_SAD_WINDOW
from oracle import oracle as O
eng=Engine(0); rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 5)
t0=time.time(); n=bad=0
while time.time()-t0 < float(sys.argv[2] if len(sys.argv)>2 else 90):
    n+=1
    w,h=int(rng.integers(1,150)),int(rng.integers(1,70)); nf=int(rng.integers(1,5))
    win=int(rng.choice([1,3,5,7,9,11])); dmax=int(rng.integers(1,70)); prec=[int(v) for v in rng.integers(0,30,3)]
    L=[rng.integers(0,256,(h,w,3),dtype=np.uint8) for _ in range(nf)]
    R=[np.roll(l,int(rng.integers(0,min(dmax,w))),axis=1).copy() for l in L]
    r0=int(rng.integers(0,h)); r1=int(rng.integers(r0+1,h+1))
    want=[O.sad_rows(l,r,dmax,win,prec)[r0:r1] for l,r in zip(L,R)]
    dev=rng.random()<0.5
    if dev:
        # device images at any byte alignment (views into larger buffers), the main one RGB or RGBA
        rgba=rng.random()<0.3; offs=[int(v) for v in rng.integers(0,4,2)]
        def view(a,off):
            b=torch.zeros(a.size+off+int(rng.integers(0,3)),dtype=torch.uint8,device="cuda")
            v=b[off:off+a.size].view(*a.shape); v.copy_(torch.from_numpy(a)); return v
        Lm=[np.concatenate([l,rng.integers(0,256,(h,w,1),dtype=np.uint8)],axis=2) if rgba else l for l in L]
        outs,mx=eng.depth_from_additional_batch([view(l,offs[0]) for l in Lm],[view(r,offs[1]) for r in R],prec,mode=MODE_SAD_WINDOW,d_max=dmax,win=win,raw_only=True,rows=(r0,r1))
        got=[o.cpu().numpy()[r0:r1] for o in outs]
    else:
        outs,mx=eng.depth_from_additional_batch(L,R,prec,mode=MODE_SAD_WINDOW,d_max=dmax,win=win,raw_only=True,rows=(r0,r1))
        got=[o[r0:r1] for o in outs]
    ok=all(np.array_equal(g,wv) for g,wv in zip(got,want)) and list(mx)==[int(wv.max()) for wv in want]
    if not ok: bad+=1; print("MISMATCH",n,w,h,nf,win,dmax,prec,(r0,r1),dev,flush=True)
print(f"sad fuzz: {n} cases, {bad} mismatches"); sys.exit(1 if bad else 0)
