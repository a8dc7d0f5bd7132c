"""The grouped weight-gradient launch (csrc/train_wgrad.hip) alone, on the problem list of one training step at 32 x 1024
(BASELINE configs[2] per GPU): microseconds per launch pair (HIP-graph replay of 20 back-to-back calls), TFLOP/s of the live
work and of the tile work (edge tiles count whole), against the library's per-problem GEMMs.
    python scripts/probes/wgrad_bench.py [--lib-gemms]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network")]
import torch  # noqa: E402

STEP = [(32768, 128, 384)] * 4 + [(32768, 128, 131), (8192, 256, 320), (4096, 256, 640), (4096, 128, 131), (8192, 64, 64),
                                  (672, 384, 1920), (672, 384, 1920), (672, 1024, 384), (672, 384, 1024), (672, 1024, 384),
                                  (672, 384, 1024), (672, 256, 384)]


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
    from hotrack_amd import train_stack as ts
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(0)
    probs = [(torch.randn((r, n), device=dev, generator=gen), torch.randn((r, k), device=dev, generator=gen), torch.empty((n, k), device=dev))
             for r, n, k in STEP]
    st = torch.cuda.current_stream().cuda_stream

    def run(sub):
        def f():
            stream = torch.cuda.current_stream().cuda_stream
            ts.wgrad_multi([ts.WgradItem(g, x, dw, dw.data_ptr(), dw.stride(0), g.shape[1], x.shape[1], stream) for g, x, dw in sub])
        return f
    out = {}
    live = sum(2.0 * r * n * k for r, n, k in STEP)
    tile = sum(2.0 * (-(-r // 32) * 32) * (-(-n // 128) * 128) * (-(-k // 128) * 128) for r, n, k in STEP)
    us = timed(run(probs))
    out["all"] = {"us": round(us, 1), "live_gflop": round(live / 1e9, 2), "tflops_live": round(live / us / 1e6, 1),
                  "tflops_tiles": round(tile / us / 1e6, 1), "mfma_frac_live": round(live / us / 1e6 / 157.3, 3)}
    big = probs[:4]
    us = timed(run(big))
    fl = sum(2.0 * g.shape[0] * g.shape[1] * x.shape[1] for g, x, _ in big)
    out["four_32768x128x384"] = {"us": round(us, 1), "tflops": round(fl / us / 1e6, 1), "mfma_frac": round(fl / us / 1e6 / 157.3, 3)}
    if "--lib-gemms" in sys.argv:
        from hotrack_amd import gemm_tuning
        with gemm_tuning.scope():
            us = timed(lambda: [torch.mm(g.t(), x) for g, x, _ in probs])
        out["library_gemms"] = {"us": round(us, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
