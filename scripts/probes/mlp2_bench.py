"""Kernel-only timing of pn2x_mlp2_rows against the two library GEMMs it replaces (GPU box)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hotrack_amd import ext, gemm_tuning
R, C = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 128
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(R, C, device="cuda", generator=g)
w2, w3 = (torch.randn(C, C, device="cuda", generator=g) / C ** 0.5 for _ in range(2))
b2, b3 = (torch.randn(C, device="cuda", generator=g) * 0.1 for _ in range(2))
out = torch.empty(R, C, device="cuda")


def t(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


fl = 2.0 * R * 2 * C * C
us = t(lambda: ext.mlp2_rows(x, w2, b2, w3, b3, out=out))
print(f"mlp2_rows     {us:7.2f} us  {fl / us * 1e-6:6.1f} TFLOP/s  frac {fl / us * 1e-6 / 157.3:.3f}")
gemm_tuning.enable()
with gemm_tuning.scope():
    us2 = t(lambda: torch._addmm_activation(b3, torch._addmm_activation(b2, x, w2.t()), w3.t()))
print(f"2 lib GEMMs   {us2:7.2f} us  {fl / us2 * 1e-6:6.1f} TFLOP/s")
