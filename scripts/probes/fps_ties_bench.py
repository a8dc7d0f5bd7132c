import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hotrack_amd import ext, pointnet2_utils as ops, pointnet2_hip as nat
import ctypes
lib = nat._lib
def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it * 1e3
for B in (1, 64):
    xyz = torch.rand(B, 1024, 3, device="cuda")
    i1 = torch.empty(B, 256, dtype=torch.int32, device="cuda"); flag = torch.empty(B, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    plain = t(lambda: lib.pn2_furthest_point_sampling(B, 1024, 256, xyz.data_ptr(), None, i1.data_ptr(), st))
    for mc in (0, 1, 64, 128, 256):
        ties = t(lambda: lib.pn2x_furthest_point_sampling_ties(B, 1024, 256, xyz.data_ptr(), i1.data_ptr(), mc, flag.data_ptr(), st))
        print(f"B={B} plain {plain:.1f} us  ties(m_check={mc}) {ties:.1f} us")
