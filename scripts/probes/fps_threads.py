import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, torch
sys.path.insert(0, %r)
from hotrack_amd import pointnet2_utils as ops
def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it * 1e3
for B, N, M in ((1, 1024, 256), (64, 1024, 256), (64, 256, 128), (8, 8192, 2048), (64, 512, 128)):
    x = torch.rand(B, N, 3, device="cuda")
    print("T=%%s B=%%d N=%%d M=%%d: %%.1f us" %% (sys.argv[1], B, N, M, t(lambda: ops.furthest_point_sample(x, M))))
''' % ROOT
for T in ("0", "64", "128", "256", "512", "1024"):
    env = dict(os.environ, PN2_FPS_THREADS=T)
    print(subprocess.run([sys.executable, "-c", code, T], env=env, capture_output=True, text=True).stdout.strip())
