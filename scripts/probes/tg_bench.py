"""Kernel-level timing of the fused BatchNorm GEMM stacks (hotrack_amd.train_stack) against the round-2 path (library GEMM +
streaming BatchNorm kernels) on the layer shapes of a 32 x 1024 training step.  Prints one JSON object.
    python scripts/probes/tg_bench.py [--iters 20]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

SHAPES = [  # name, rows, widths, max_over
    ("sa1", 32 * 256 * 32, [32, 32, 64], 32),
    ("sa2", 32 * 128 * 32, [64, 64, 128], 32),
    ("sa3", 32 * 128, [128, 128, 512], 128),
    ("fp3", 32 * 128, [256, 256], 0),
    ("fp2", 32 * 256, [256, 128], 0),
    ("fp1+conv1", 32 * 1024, [128, 128, 384], 0),
    ("q K=16", 32 * 21 * 16, [128, 128, 192], 16),
    ("q K=64", 32 * 21 * 64, [128, 128, 192], 64),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from hotrack_amd import gemm_tuning, train_stack
    from hotrack_amd.train_ops import Workspace, bn_relu, bn_relu_max
    res = {}
    for name, R, widths, K in SHAPES:
        convs = [torch.nn.Conv1d(x, y, 1).cuda() for x, y in zip(widths[:-1], widths[1:])]
        bns = [torch.nn.BatchNorm1d(c).cuda().train() for c in widths]
        y1 = torch.randn(R, widths[0], device="cuda")
        ws = Workspace("cuda")

        def fused(y):
            layers = [train_stack.Layer(None, bns[0])] + [train_stack.Layer(c.weight.view(c.weight.shape[0], -1), bn, c.bias)
                                                         for c, bn in zip(convs, bns[1:])]
            return train_stack.mlp_stack(y, layers, ws, max_over=K)

        def unfused(y):
            x = bn_relu(y, bns[0], ws, None) if len(widths) > 1 or not K else bn_relu_max(y, K, bns[0], ws, None)
            for i, (c, bn) in enumerate(zip(convs, bns[1:])):
                yy = F.linear(x, c.weight.view(c.weight.shape[0], -1))
                last = i == len(convs) - 1
                x = bn_relu_max(yy, K, bn, ws, c.bias) if (K and last) else bn_relu(yy, bn, ws, c.bias)
            return x

        out = {}
        go_cache = {}

        def step(fn, backward):
            ws.reset()
            y = y1.detach().requires_grad_(True)
            o = fn(y)
            if backward:
                key = tuple(o.shape)
                if key not in go_cache:
                    go_cache[key] = torch.ones_like(o)
                (gy,) = torch.autograd.grad(o, y, go_cache[key])
                return gy
            return o

        for label, fn in (("fused", fused), ("round2", unfused)):
            with gemm_tuning.scope():
                times = {}
                for backward in (False, True):  # HIP-graph replays: kernel time without host launch gaps
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        for _ in range(3):
                            step(fn, backward)
                    torch.cuda.current_stream().wait_stream(side)
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        keep = step(fn, backward)
                    for _ in range(3):
                        g.replay()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.iters):
                        g.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    times[backward] = e0.elapsed_time(e1) * 1e3 / a.iters
                    del g, keep
            out[label] = {"fwd_us": round(times[False], 1), "bwd_us": round(times[True] - times[False], 1)}
        flops_f = 2.0 * R * sum(x * y for x, y in zip(widths[:-1], widths[1:]))
        act_bytes = 4.0 * R * sum(widths)
        out["fwd_gflop"] = round(flops_f / 1e9, 3)
        out["activation_MB_one_pass"] = round(act_bytes / 1e6, 1)
        res[name] = out
    print(json.dumps({"note": "HIP-graph replays timed with HIP events (forward alone; backward = forward+backward minus forward); a workspace fill launch is included in each", "shapes": res}))


if __name__ == "__main__":
    main()
