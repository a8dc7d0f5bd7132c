"""Offline GEMM algorithm selection for the inference path's library GEMMs (PyTorch TunableOp over hipBLASLt /
rocBLAS) on the GPU it runs on.  Runs the fused HandTrackNet forward eagerly at the given batch sizes with tuning
on, writes the chosen solution per (op, shape) to a CSV, and reports graph-replay time before / after.

usage: python scripts/tune_gemms.py [--out hotrack_amd/tunableop_gfx950.csv] [--batches 1 8 64]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402
import torch.cuda.tunable as tunable  # noqa: E402

from netinit import deterministic_init, make_cfg, synthetic_frames  # noqa: E402
from hotrack_amd import fused, pointnet2_utils  # noqa: E402
from models import pointnet_utils  # noqa: E402
from models.hand_network import HandTrackNet  # noqa: E402

FLAGS = {"track_flag": False, "test_flag": True, "save_flag": False, "IKNet_flag": False}


def graph_ms(model, d, iters=100):
    with torch.no_grad():
        for _ in range(3):
            model(d, dict(FLAGS))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            model(d, dict(FLAGS))
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            g.replay()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tunableop_gfx950.csv"))
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 2, 4, 8, 16, 32, 64])
    ap.add_argument("--max-ms", type=int, default=30)
    a = ap.parse_args()
    os.environ["PN2_TUNED_GEMMS"] = "0"  # start from the library defaults
    pointnet_utils.set_operator_backend(pointnet2_utils)
    pointnet_utils.set_fused_backend(fused)
    model = HandTrackNet(make_cfg("cuda"))
    deterministic_init(model)
    model = model.cuda().eval()
    data = {}
    for B in a.batches:
        d = synthetic_frames(5, B, 1024)
        data[B] = {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in d.items()}
    res = {"before_ms": {B: round(graph_ms(model, data[B]), 4) for B in a.batches}}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_filename(a.out)
    tunable.set_max_tuning_duration(a.max_ms)
    tunable.set_max_tuning_iterations(50)
    t0 = time.perf_counter()
    with torch.no_grad():
        for B in a.batches:
            model(data[B], dict(FLAGS))
            torch.cuda.synchronize()
    res["tuning_s"] = round(time.perf_counter() - t0, 1)
    getattr(tunable, "write_file", lambda: None)()  # older builds flush on exit only
    tunable.tuning_enable(False)
    res["entries"] = len(tunable.get_results())
    res["after_ms"] = {B: round(graph_ms(model, data[B]), 4) for B in a.batches}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
