"""Offline GEMM algorithm selection for the TRAINING step's library GEMMs (PyTorch TunableOp over hipBLASLt / rocBLAS).
The point-major training path issues every 1x1 convolution as a GEMM over all B*S*K positions; its weight-gradient GEMMs
are (C_out x R) . (R x C_in) with R up to 262144 and C <= 512 -- shapes the libraries' default heuristic handles badly
(no split along R).  Runs a few eager training steps with tuning on and MERGES the chosen solutions into the shipped table.

usage (GPU box): python scripts/tune_gemms_train.py [--batches 32] [--out gpurun_out/tunableop_gfx950.csv]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network")]
import torch  # noqa: E402
import torch.cuda.tunable as tunable  # noqa: E402


def step_ms(tr, batch, iters=20):
    for _ in range(3):
        tr.update(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        tr.update(batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tunableop_gfx950.csv"))
    ap.add_argument("--batches", type=int, nargs="+", default=[32])
    ap.add_argument("--max-ms", type=int, default=30)
    a = ap.parse_args()
    os.environ["PN2_TUNED_GEMMS"] = "0"
    os.environ.setdefault("HOTRACK_DATA_ROOT", "/tmp/hotrack_bench_data")
    from configs.config import get_config
    from datasets.synthetic import make_frame
    from hotrack_amd import gemm_tuning
    from parse_args import add_args
    from trainer import Trainer
    args = add_args(argparse.ArgumentParser()).parse_args(["--config", "handtracknet_train_SimGrasp.yml"])
    args.num_points, args.batch_size = 1024, a.batches[0]
    cfg = get_config(args, save=False)
    cfg["graph_step"] = False
    torch.manual_seed(0)
    tr = Trainer(cfg)
    tr.step_epoch()
    data = {}
    for B in a.batches:
        b = torch.utils.data.default_collate([make_frame(i, 1024, 0.02) for i in range(B)])
        data[B] = {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in b.items()}
    res = {"before_ms": {B: round(step_ms(tr, data[B]), 3) for B in a.batches}}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    tmp = a.out + ".train_only"
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_filename(tmp)
    tunable.set_max_tuning_duration(a.max_ms)
    tunable.set_max_tuning_iterations(30)
    t0 = time.perf_counter()
    for B in a.batches:
        tr.update(data[B])
        torch.cuda.synchronize()
    res["tuning_s"] = round(time.perf_counter() - t0, 1)
    getattr(tunable, "write_file", lambda: None)()
    tunable.tuning_enable(False)
    res["entries"] = len(tunable.get_results())
    res["after_ms"] = {B: round(step_ms(tr, data[B]), 3) for B in a.batches}
    # merge: shipped inference table + the new rows (same validators: same image / GPU)
    old = open(gemm_tuning.RESULTS).read().splitlines() if os.path.exists(gemm_tuning.RESULTS) else []
    new = open(tmp).read().splitlines()
    head = [l for l in new if l.startswith("Validator")]
    old_head = [l for l in old if l.startswith("Validator")]
    res["validators_match"] = head == old_head
    rows = {}
    for l in ([x for x in old if not x.startswith("Validator")] if head == old_head else []) + [x for x in new if not x.startswith("Validator")]:
        key = ",".join(l.split(",")[:2])
        rows[key] = l
    with open(a.out, "w") as f:
        f.write("\n".join(head + list(rows.values())) + "\n")
    res["merged_rows"] = len(rows)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
