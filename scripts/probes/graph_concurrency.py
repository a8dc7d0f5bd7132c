"""Does a captured HIP graph run independent branches concurrently on MI355X / ROCm 7.2?
Two FPS launches (64 workgroups each: a quarter of the chip) on two streams inside one graph."""
import os, sys, time, torch
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hotrack_amd import pointnet2_utils as ops
x1=torch.rand(64,1024,3,device='cuda'); x2=torch.rand(64,1024,3,device='cuda')
def timeit(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
s1,s2=torch.cuda.Stream(),torch.cuda.Stream()
def serial():
    ops.furthest_point_sample(x1,256); ops.furthest_point_sample(x2,256)
def forked():
    cur=torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): ops.furthest_point_sample(x1,256)
    with torch.cuda.stream(s2): ops.furthest_point_sample(x2,256)
    cur.wait_stream(s1); cur.wait_stream(s2)
serial(); forked(); torch.cuda.synchronize()
print("eager serial  %.1f us"%timeit(serial)); print("eager forked  %.1f us"%timeit(forked))
for name,fn in (("graph serial",serial),("graph forked",forked)):
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    print("%s  %.1f us"%(name,timeit(g.replay)))
