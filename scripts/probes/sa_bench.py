"""Kernel-only timing + fp64 check of the fused set-abstraction kernel at the production shapes (GPU box).

    python scripts/probes/sa_bench.py [--iters 300] [--json out.json]

Shapes = the four launches of one bench.py step (B=64): sa1, sa2, the q1 pair (a1f + xyz) and the q2 pair (+ cadd), plus the
single K=64 / K=16 launches of the q shape.  Timing: HIP events around `iters` back-to-back launches on the current stream.
The check compares with an fp64 torch evaluation of the same formula (max |diff| printed; > 2e-4 fails)."""
import argparse, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hotrack_amd import ext

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--json", default=None)
ap.add_argument("--ld", type=int, default=512, help="row stride (floats) of the q-shape a1f / cadd operands")
ap.add_argument("--local-idx", action="store_true", help="neighbour indices = consecutive points (cache-friendly gather)")
args = ap.parse_args()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
PEAK = 157.3


def problem(B, N, S, K, C1, C2, C3, a1f, cadd, ld=None):
    ld = ld or C1
    idx = torch.randint(0, N, (B, S, K), device=dev, dtype=torch.int32, generator=g)
    if args.local_idx:
        idx = (torch.arange(S * K, device=dev, dtype=torch.int32) % N).view(1, S, K).expand(B, S, K).contiguous()
    p = dict(idx=idx,
             w2=torch.randn(C2, C1, device=dev, generator=g) * (1.0 / C1 ** 0.5), b2=torch.randn(C2, device=dev, generator=g) * 0.1,
             w3=torch.randn(C3, C2, device=dev, generator=g) * (1.0 / C2 ** 0.5), b3=torch.randn(C3, device=dev, generator=g) * 0.1,
             xyz=torch.rand(B, N, 3, device=dev, generator=g), cxyz=torch.rand(B, S, 3, device=dev, generator=g),
             wx=torch.randn(C1, 3, device=dev, generator=g), b1=torch.randn(C1, device=dev, generator=g) * 0.1,
             a1f=None, cadd=None, out=torch.zeros(B, S, C3, device=dev))
    if a1f:
        p["a1f"] = torch.randn(B, N, ld, device=dev, generator=g)[:, :, :C1]
    if cadd:
        p["cadd"] = torch.randn(B, S, ld, device=dev, generator=g)[:, :, :C1]
    return p


def reference(p):
    B, S, K = p["idx"].shape
    ii = p["idx"].long()
    bi = torch.arange(B, device=dev).view(B, 1, 1)
    d = p["xyz"].double()[bi, ii] - p["cxyz"].double()[:, :, None, :]
    h = d @ p["wx"].double().t() + p["b1"].double()
    if p["a1f"] is not None:
        h = h + p["a1f"].double()[bi, ii]
    if p["cadd"] is not None:
        h = h + p["cadd"].double()[:, :, None, :]
    h = torch.relu(h)
    h = torch.relu(h @ p["w2"].double().t() + p["b2"].double())
    h = torch.relu(h @ p["w3"].double().t() + p["b3"].double())
    return h.max(dim=2).values.float()


def run_single(p):
    ext.sa_mlp_max(p["idx"], p["w2"], p["b2"], p["w3"], p["b3"], a1f=p["a1f"], xyz=p["xyz"], cxyz=p["cxyz"], wx=p["wx"], b1=p["b1"],
                   cadd=p["cadd"], out=p["out"])


def time_us(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / args.iters


def flops(ps):
    t = 0
    for p in ps:
        B, S, K = p["idx"].shape
        C2, C1 = p["w2"].shape
        C3 = p["w3"].shape[0]
        t += 2.0 * B * S * K * (C1 * C2 + C2 * C3)
    return t


B = args.batch
rows = []
cases = [
    ("sa1 32-32-64 K32", [problem(B, 1024, 256, 32, 32, 32, 64, False, False)]),
    ("sa2 64-64-128 K32", [problem(B, 256, 128, 32, 64, 64, 128, True, False)]),
    ("q K64 128-128-192", [problem(B, 1024, 21, 64, 128, 128, 192, True, False, ld=args.ld)]),
    ("q K16 128-128-192", [problem(B, 1024, 21, 16, 128, 128, 192, True, False, ld=args.ld)]),
    ("q1 pair 16+64", [problem(B, 1024, 21, 16, 128, 128, 192, True, False, ld=args.ld), problem(B, 1024, 21, 64, 128, 128, 192, True, False, ld=args.ld)]),
    ("q2 pair 16+64 +cadd", [problem(B, 1024, 21, 16, 128, 128, 192, True, True, ld=args.ld), problem(B, 1024, 21, 64, 128, 128, 192, True, True, ld=args.ld)]),
]
ok = True
for name, ps in cases:
    fn = (lambda ps=ps: run_single(ps[0])) if len(ps) == 1 else (lambda ps=ps: ext.sa_mlp_max_pair(ps[0], ps[1]))
    fn()
    torch.cuda.synchronize()
    err = max(float((p["out"] - reference(p)).abs().max()) for p in ps)
    us = time_us(fn)
    tf = flops(ps) / us * 1e-6
    good = err < 2e-4
    ok &= good
    rows.append(dict(case=name, us=round(us, 2), tflops=round(tf, 1), frac=round(tf / PEAK, 4), max_err=err, ok=good))
    print(f"{name:24s} {us:8.2f} us  {tf:6.1f} TFLOP/s  frac {tf / PEAK:.3f}  max|err| {err:.2e} {'ok' if good else 'FAIL'}")
if args.json:
    json.dump(rows, open(args.json, "w"), indent=1)
sys.exit(0 if ok else 1)
