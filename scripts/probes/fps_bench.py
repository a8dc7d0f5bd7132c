import os, sys, torch
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hotrack_amd import pointnet2_utils as ops
def t(fn,n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
for B,N,M in [(64,1024,256),(64,256,128),(1,1024,256),(8,8192,2048),(64,2048,512),(64,512,128)]:
    x=torch.rand(B,N,3,device='cuda')
    r=[]
    for thr in ("0","64","256","1024"):
        os.environ["PN2_FPS_THREADS"]=thr
        us=t(lambda: ops.furthest_point_sample(x,M), n=10 if N>2048 else 30)
        r.append(f"T={thr}: {us:8.1f} us ({us/M:.3f}/it)")
    print(B,N,M," | ".join(r))
