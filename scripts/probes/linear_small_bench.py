"""pn2x_linear_small vs the BLAS library (with the shipped solution table) on the dense-layer shapes of a B = 1 frame.
Graph-replay timing, 20 calls per replay.  Prints one JSON object."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

SHAPES = [(256, 64, 64), (128, 131, 128), (128, 128, 128), (128, 128, 512), (1, 512, 256), (128, 128, 256), (128, 256, 256), (256, 320, 256),
          (256, 256, 128), (1024, 128, 384), (1024, 384, 512), (21, 768, 384), (21, 384, 256), (21, 384, 1024), (21, 1024, 384), (21, 384, 128),
          (168, 384, 256), (1344, 384, 256)]


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
    from hotrack_amd import ext, gemm_tuning
    gemm_tuning.enable()
    out = {}
    with torch.no_grad(), gemm_tuning.scope():
        for M, K, N in SHAPES:
            x = torch.randn(M, K, device="cuda")
            w = torch.randn(N, K, device="cuda")
            b = torch.randn(N, device="cuda")
            old = ext.LINEAR_SMALL_MAX_ROWS
            ext.LINEAR_SMALL_MAX_ROWS = 1 << 30
            ys = ext.linear(x, w, b, relu=True)
            ts = timed(lambda: ext.linear(x, w, b, relu=True))
            ext.LINEAR_SMALL_MAX_ROWS = old
            yl = torch._addmm_activation(b, x, w.t())
            tl = timed(lambda: torch._addmm_activation(b, x, w.t()))
            err = float((ys - yl).abs().max() / (yl.abs().max() + 1e-9))
            out[f"{M}x{K}->{N}"] = {"small_us": round(ts, 2), "library_us": round(tl, 2), "rel_err": err}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
