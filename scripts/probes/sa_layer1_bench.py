"""Kernel-level timing of train_ops.sa_layer1, bn_stats over its output, and the xyz weight gradient (rows_outer3) against
torch.mm, on the SA shapes of a 32 x 1024 training step.  Prints one JSON object."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network")]
import torch  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / iters


def main():
    from hotrack_amd import train_ops
    from hotrack_amd.train_ops import sa_layer1
    lib = train_ops._lib
    out = {}
    for name, B, N, S, K, C1, D in [("sa1", 32, 1024, 256, 32, 32, 0), ("sa2", 32, 256, 128, 32, 64, 64), ("q64", 32, 1024, 21, 64, 128, 384), ("q16", 32, 1024, 21, 16, 128, 384)]:
        xyz = torch.rand(B, N, 3, device="cuda")
        cxyz = torch.rand(B, S, 3, device="cuda")
        a1f = torch.randn(B, N, C1, device="cuda") if D else None
        idx = torch.randint(0, N, (B, S, K), device="cuda", dtype=torch.int32)
        wx = torch.randn(C1, 3, device="cuda")
        y = sa_layer1(a1f, None, xyz, cxyz, [idx], [wx])[0]
        sums = torch.zeros(lib.pn2x_bn_sums_doubles(C1), dtype=torch.float64, device="cuda")
        R = B * S * K

        def plain():
            sa_layer1(a1f, None, xyz, cxyz, [idx], [wx])

        def stats_after():
            lib.pn2x_bn_stats(R, C1, y.data_ptr(), C1, sums.data_ptr(), torch.cuda.current_stream().cuda_stream)

        from hotrack_amd.train_ops import Workspace
        ws = Workspace("cuda")

        def with_stats():
            ws.used = 0
            sa_layer1(a1f, None, xyz, cxyz, [idx], [wx], ws=ws, aux={})

        dy = torch.randn(R, C1, device="cuda")
        rel = torch.randn(R, 3, device="cuda")
        dwx = torch.empty(C1, 3, device="cuda")
        scratch = torch.empty(int(lib.pn2x_rows_outer3_scratch_floats(R, C1)), device="cuda")

        def outer3():
            lib.pn2x_rows_outer3(R, C1, dy.data_ptr(), C1, rel.data_ptr(), dwx.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                 torch.cuda.current_stream().cuda_stream)

        def mm():
            torch.mm(dy.t(), rel)

        outer3()
        err = float((dwx - torch.mm(dy.t(), rel)).abs().max())
        out[name] = {"rows": R, "c1": C1, "sa_layer1_us": round(timed(plain), 1), "sa_layer1_stats_us": round(timed(with_stats), 1), "bn_stats_us": round(timed(stats_after), 1),
                     "rows_outer3_us": round(timed(outer3), 1),
                     "torch_mm_us": round(timed(mm), 1), "outer3_max_err": err}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
