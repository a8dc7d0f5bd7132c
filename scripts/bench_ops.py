"""Per-operator micro-benchmark on one MI355X: time, algorithmic GB/s (SURVEY.md 8(d) formulas)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotrack_amd import pointnet2_utils as ops  # noqa: E402


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def timeit_graph(fn, reps=10, iters=20):
    """GPU time per call with the host out of the loop: `reps` calls captured into one HIP graph, replayed `iters` times."""
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (iters * reps) * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    B = a.B
    g = torch.Generator(device="cuda").manual_seed(0)
    res = []

    def rec(name, us, nbytes, **kw):
        r = dict(op=name, us=round(us, 2), alg_MB=round(nbytes / 1e6, 3), GBps=round(nbytes / us / 1e3, 1), **kw)
        res.append(r)
        print(json.dumps(r), flush=True)

    for N, M in [(1024, 256), (256, 128), (8192, 2048), (5120, 1024)]:
        b = B if N <= 1024 else max(1, B // 8)
        xyz = torch.rand(b, N, 3, device="cuda", generator=g)
        for thr in ["0", "64", "256", "1024"]:
            os.environ["PN2_FPS_THREADS"] = thr
            us = timeit(lambda: ops.furthest_point_sample(xyz, M), iters=20)
            rec("fps", us, b * (12 * N + 4 * M), B=b, N=N, M=M, threads=thr, us_per_iter=round(us / M, 3))
        os.environ["PN2_FPS_THREADS"] = "0"

    for N, S, r, K in [(1024, 256, 0.1, 32), (256, 128, 0.2, 32), (8192, 2048, 0.2, 64), (8192, 2048, 0.1, 64)]:
        b = B if N <= 1024 else max(1, B // 8)
        xyz = torch.rand(b, N, 3, device="cuda", generator=g)
        idx = ops.furthest_point_sample(xyz, S)
        new = torch.gather(xyz, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        us = timeit(lambda: ops.ball_query(r, K, xyz, new))
        rec("ball_query", us, b * (12 * N + 12 * S + 4 * S * K), B=b, N=N, S=S, r=r, K=K)

    for n, m in [(256, 128), (1024, 256)]:
        u = torch.rand(B, n, 3, device="cuda", generator=g)
        k = torch.rand(B, m, 3, device="cuda", generator=g)
        us = timeit(lambda: ops.three_nn(u, k))
        rec("three_nn", us, B * (12 * n + 12 * m + 24 * n), B=B, n=n, m=m)

    for n, m, k in [(21, 1024, 16), (21, 1024, 64), (21, 1024, 4)]:
        u = torch.rand(B, n, 3, device="cuda", generator=g)
        kn = torch.rand(B, m, 3, device="cuda", generator=g)
        us = timeit(lambda: ops.knn(k, u, kn))
        rec("knn", us, B * (12 * n + 12 * m + 8 * n * k), B=B, n=n, m=m, k=k)

    for C, N, P, S in [(3, 1024, 256, 32), (64, 256, 128, 32), (384, 1024, 21, 16), (384, 1024, 21, 64), (67, 8192, 2048, 64)]:
        b = B if N <= 1024 else max(1, B // 8)
        f = torch.randn(b, C, N, device="cuda", generator=g)
        idx = torch.randint(0, N, (b, P, S), device="cuda", dtype=torch.int32, generator=g)
        us = timeit_graph(lambda: ops.grouping_operation(f, idx))
        nb = b * (4 * P * S + 4 * C * min(N, P * S) + 4 * C * P * S)
        rec("group_fwd", us, nb, B=b, C=C, N=N, P=P, S=S)
        go = torch.randn(b, C, P, S, device="cuda", generator=g)
        f2 = f.clone().requires_grad_(True)
        out = ops.grouping_operation(f2, idx)
        us = timeit(lambda: torch.autograd.grad(out, f2, go, retain_graph=True))
        gp = torch.zeros_like(f)
        from hotrack_amd import pointnet2_hip as native
        us_k = timeit_graph(lambda: native.group_points_grad_wrapper(b, C, N, P, S, go, idx, gp))  # kernel only (accumulates)
        rec("group_bwd", us_k, nb, B=b, C=C, N=N, P=P, S=S, us_autograd=round(us, 2))

    for C, M, n in [(256, 128, 256), (128, 256, 1024)]:
        f = torch.randn(B, C, M, device="cuda", generator=g)
        idx = torch.randint(0, M, (B, n, 3), device="cuda", dtype=torch.int32, generator=g)
        w = torch.rand(B, n, 3, device="cuda", generator=g)
        us = timeit_graph(lambda: ops.three_interpolate(f, idx, w))
        nb = B * (4 * C * M + 24 * n + 4 * C * n)
        rec("interp_fwd", us, nb, B=B, C=C, M=M, n=n)
        f2 = f.clone().requires_grad_(True)
        out = ops.three_interpolate(f2, idx, w)
        go = torch.randn_like(out)
        us = timeit(lambda: torch.autograd.grad(out, f2, go, retain_graph=True))
        gp = torch.zeros_like(f)
        from hotrack_amd import pointnet2_hip as native
        us_k = timeit_graph(lambda: native.three_interpolate_grad_wrapper(B, C, n, M, go, idx, w, gp))  # kernel only (accumulates)
        rec("interp_bwd", us_k, nb, B=B, C=C, M=M, n=n, us_autograd=round(us, 2))

    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
