"""BASELINE.json configs[4] ("dense stress"): N=8192 points, npoint=2048, nsample=64, C=64, 64 clouds per GPU.
Per-operator time, algorithmic GB/s (SURVEY.md 8(d) byte formulas) as a fraction of the HBM roofline, and the fused
set-abstraction kernel's TFLOP/s against the fp32 MFMA peak.  One GPU; rank-local under torchrun."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotrack_amd import ext  # noqa: E402
from hotrack_amd import pointnet2_utils as ops  # noqa: E402

HBM, MFMA = 8000.0, 157.3


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    B, N, S, K, C = a.B, 8192, 2048, 64, 64
    g = torch.Generator(device="cuda").manual_seed(0)
    xyz = torch.rand(B, N, 3, device="cuda", generator=g)
    feat = torch.randn(B, C, N, device="cuda", generator=g)
    res = []

    def rec(name, sec, nbytes=None, flops=None, **kw):
        r = dict(op=name, ms=round(sec * 1e3, 3), **kw)
        if nbytes is not None:
            r.update(alg_GB=round(nbytes / 1e9, 3), GBps=round(nbytes / sec / 1e9, 1), hbm_frac=round(nbytes / sec / 1e9 / HBM, 4))
        if flops is not None:
            r.update(TFLOPs=round(flops / sec / 1e12, 2), mfma_frac=round(flops / sec / 1e12 / MFMA, 4))
        res.append(r)
        print(json.dumps(r), flush=True)

    t = timeit(lambda: ops.furthest_point_sample(xyz, S), iters=3, warm=1)
    rec("fps 8192->2048", t, B * (12 * N + 4 * S), us_per_iteration=round(t * 1e6 / S, 3))
    fps = ops.furthest_point_sample(xyz, S)
    new_xyz = ext.gather_rows(xyz, fps)
    for r_ in (0.2, 0.1):
        t = timeit(lambda: ops.ball_query(r_, K, xyz, new_xyz))
        rec(f"ball_query r={r_}", t, B * (12 * N + 12 * S + 4 * S * K))
    idx = ops.ball_query(0.2, K, xyz, new_xyz)
    xyz_cm = xyz.transpose(1, 2).contiguous()
    t = timeit(lambda: ops.grouping_operation(xyz_cm, idx))
    rec("group xyz (C=3)", t, B * (4 * S * K + 4 * 3 * N + 4 * 3 * S * K))
    t = timeit(lambda: ops.grouping_operation(feat, idx))
    rec("group feat (C=64), unfused", t, B * (4 * S * K + 4 * C * N + 4 * C * S * K))
    W1 = torch.randn(64, C + 3, device="cuda", generator=g) * 0.2
    b1 = torch.randn(64, device="cuda", generator=g)
    W2 = torch.randn(64, 64, device="cuda", generator=g) * 0.2
    b2 = torch.randn(64, device="cuda", generator=g)
    W3 = torch.randn(128, 64, device="cuda", generator=g) * 0.2
    b3 = torch.randn(128, device="cuda", generator=g)
    w1f_t, wx = W1[:, :C].t().contiguous(), W1[:, C:].contiguous()

    def fused():
        a1f = torch.matmul(feat.transpose(1, 2), w1f_t)
        return ext.sa_mlp_max(idx, W2, b2, W3, b3, a1f=a1f, xyz=xyz, cxyz=new_xyz, wx=wx, b1=b1)
    t = timeit(fused)
    # flops ACTUALLY executed: layer 1 is split -- its per-point half is a GEMM over the N points (not over the S*K grouped slots),
    # the xyz half 3 MACs per slot and channel in-kernel -- then layers 2-3 over all slots
    flops = 2.0 * B * (N * C * 64 + S * K * (3 * 64 + 64 * 64 + 64 * 128))
    comp = B * (4 * C * N + 12 * N + 12 * S + 4 * S * K + 4 * 128 * S)  # compulsory traffic of a fused SA layer
    rec("fused SA layer [67->64->64->128] + max (a1f GEMM + sa_mlp_max), executed flops", t, comp, flops,
        equivalent_TFLOPs_of_the_grouped_formulation=round(2.0 * B * S * K * ((C + 3) * 64 + 64 * 64 + 64 * 128) / t / 1e12, 2))
    a1f = torch.matmul(feat.transpose(1, 2), w1f_t)
    t = timeit(lambda: ext.sa_mlp_max(idx, W2, b2, W3, b3, a1f=a1f, xyz=xyz, cxyz=new_xyz, wx=wx, b1=b1))
    rec("sa_mlp_max kernel alone", t, None, 2.0 * B * S * K * (64 * 64 + 64 * 128))
    # whole set-abstraction level of the stress shape: sample -> gather -> query -> fused MLP + max (what one SA layer costs)
    def whole():
        i = ops.furthest_point_sample(xyz, S)
        c = ext.gather_rows(xyz, i)
        j = ops.ball_query(0.2, K, xyz, c)
        a1 = torch.matmul(feat.transpose(1, 2), w1f_t)
        return ext.sa_mlp_max(j, W2, b2, W3, b3, a1f=a1, xyz=xyz, cxyz=c, wx=wx, b1=b1)
    t = timeit(whole, iters=3, warm=1)
    rec("whole SA level (fps + gather + ball_query + a1f GEMM + sa_mlp_max), %d clouds" % B, t, None, None, clouds_per_s=round(B / t, 1))
    assert all(r.get("mfma_frac", 0) <= 1 and r.get("hbm_frac", 0) <= 1 for r in res), "a roofline fraction above 1 is a labelling bug"
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
