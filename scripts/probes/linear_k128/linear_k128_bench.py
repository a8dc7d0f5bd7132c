"""pn2x_linear_k128 (csrc/linear_k128.hip) against the library's recorded solution for the backbone's conv1 layer and its smaller
siblings: microseconds per launch (HIP-graph replay of 20 back-to-back launches), TFLOP/s.  python scripts/probes/linear_k128_bench.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network")]
import torch  # noqa: E402


def timed(fn, iters=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / (5 * iters)


def main():
    from hotrack_amd import ext, gemm_tuning
    out = {}
    for rows, n in [(65536, 384), (32768, 384), (16384, 384), (8192, 384), (65536, 128), (65536, 256), (1024, 384)]:
        x = torch.randn(rows, 128, device="cuda")
        w = torch.randn(n, 128, device="cuda") * 0.1
        b = torch.randn(n, device="cuda")
        old = ext.LINEAR_K128_MIN_ROWS
        ext.LINEAR_K128_MIN_ROWS = 1
        mine = timed(lambda: ext.linear(x, w, b, relu=True))
        ext.LINEAR_K128_MIN_ROWS = 0
        with gemm_tuning.scope():
            lib = timed(lambda: ext.linear(x, w, b, relu=True))
        ext.LINEAR_K128_MIN_ROWS = old
        fl = 2.0 * rows * 128 * n
        out[f"{rows}x128->{n}"] = {"k128_us": round(mine, 1), "k128_tflops": round(fl / mine / 1e6, 1), "k128_mfma_frac": round(fl / mine / 1e6 / 157.3, 3),
                                   "library_us": round(lib, 1), "library_tflops": round(fl / lib / 1e6, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
