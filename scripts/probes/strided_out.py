"""Can torch._addmm_activation write into a row-strided column block (ld > n) without a copy?"""
import torch
torch.manual_seed(0)
M, K, N, LD = 65536, 128, 384, 388
x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
buf = torch.zeros(M, LD, device="cuda")
ref = torch._addmm_activation(b, x, W.t())
def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it * 1e3
try:
    torch._addmm_activation(b, x, W.t(), out=buf[:, :N])
    print("strided out ok, max diff", (buf[:, :N] - ref).abs().max().item(), "tail untouched", buf[:, N:].abs().max().item())
    print("contig  %.1f us" % t(lambda: torch._addmm_activation(b, x, W.t())))
    print("strided %.1f us" % t(lambda: torch._addmm_activation(b, x, W.t(), out=buf[:, :N])))
except Exception as ex:
    print("strided out failed:", ex)
# K = 388 GEMM reading the strided buffer as A (lda = 388)
W2 = torch.randn(512, LD, device="cuda") * 0.05
print("gemm K=388 %.1f us" % t(lambda: torch.mm(buf, W2.t())))
src = buf[:, :N].contiguous()
print("gemm K=384 %.1f us" % t(lambda: torch.mm(src, W2[:, :N].t().contiguous().t() if False else W2[:, :N].contiguous().t())))
xs = buf[:, :N]
print("gemm K=384 strided A %.1f us" % t(lambda: torch.mm(xs, W2[:, :N].contiguous().t())))
