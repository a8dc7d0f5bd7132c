"""Per-phase cycle stamps of workgroup 0 of the 32-32-64 set-abstraction kernel (sa1 of the inference path: B = 64, S = 256,
K = 32, coordinates only).  Needs the SA_TRACE variant:
    python -c "from hotrack_amd import _build; print(_build.build_variant('satrace', ['-DSA_TRACE=1']))"
    PN2_LIB_PATH=hotrack_amd/libpn2_hip.satrace.so python scripts/probes/sa1_trace.py"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hotrack_amd import ext, pointnet2_hip as nat
lib = nat._lib
lib.pn2x_debug_set_sa_trace.argtypes = [ctypes.c_void_p]
B, N, S, K, C1, C2, C3 = 64, 1024, 256, 32, 32, 32, 64
g = torch.Generator(device='cuda').manual_seed(0)
xyz = torch.rand(B, N, 3, device='cuda', generator=g); cxyz = torch.rand(B, S, 3, device='cuda', generator=g)
idx = torch.randint(0, N, (B, S, K), device='cuda', dtype=torch.int32, generator=g)
w2 = torch.randn(C2, C1, device='cuda', generator=g) * 0.05; b2 = torch.randn(C2, device='cuda', generator=g)
w3 = torch.randn(C3, C2, device='cuda', generator=g) * 0.05; b3 = torch.randn(C3, device='cuda', generator=g)
wx = torch.randn(C1, 3, device='cuda', generator=g); b1 = torch.randn(C1, device='cuda', generator=g)
run = lambda: ext.sa_mlp_max(idx, w2, b2, w3, b3, xyz=xyz, cxyz=cxyz, wx=wx, b1=b1)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print("launch us", e0.elapsed_time(e1) * 1e3 / 20)
tr = torch.zeros(2 * 8 * 8, dtype=torch.int64, device='cuda')
lib.pn2x_debug_set_sa_trace(tr.data_ptr())
run()
torch.cuda.synchronize()
lib.pn2x_debug_set_sa_trace(None)
t = tr.cpu().view(2, 8, 8)
names = ["start", "mfma2 done|idx loaded", "H2 written|half0 done", "after B1", "mfma3 done|a1f loaded", "epilogue|half1 done", "after B2", "xyz loaded|rows issued"]
for role, rn in ((0, "compute wave0"), (1, "loader wave4")):
    print(rn)
    for it in range(8):
        row = t[role, it]
        if row[0] == 0: continue
        base = t[0, 0, 0]
        print("  it", it, " ".join(f"{names[i]}={int(row[i] - base)}" for i in range(8) if row[i] != 0))
