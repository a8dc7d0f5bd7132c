"""BASELINE.json configs[2]: HandTrackNet training step (handtracknet_train_SimGrasp.yml hyper-parameters), 32 clouds per
GPU x 1024 points, Adam, DDP over RCCL when launched with torchrun.  Prints one JSON line (rank 0)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--graph", action="store_true", help="whole-step HIP graph (single GPU)")
    a = ap.parse_args()
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local % torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("PN2_DIST_BACKEND", "nccl"))  # gloo: several ranks on one GPU (self-test)
    os.environ.setdefault("HOTRACK_DATA_ROOT", "/tmp/hotrack_bench_data")
    from configs.config import get_config
    from datasets.synthetic import make_frame
    from parse_args import add_args
    from trainer import Trainer
    args = add_args(argparse.ArgumentParser()).parse_args(["--config", "handtracknet_train_SimGrasp.yml"])
    args.num_points, args.batch_size = 1024, a.batch
    cfg = get_config(args, save=False)
    cfg["graph_step"] = a.graph  # data parallel: forward+backward graph | eager flat all-reduce | Adam graph
    torch.manual_seed(0)
    tr = Trainer(cfg)
    tr.step_epoch()
    batches = [torch.utils.data.default_collate([make_frame(1000 * rank + 64 * j + i, 1024, 0.02) for i in range(a.batch)]) for j in range(4)]
    batches = [{k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in b.items()} for b in batches]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    for i in range(a.warmup):
        loss = tr.update(batches[i % 4])
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = tr.update(batches[i % 4])
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    if rank == 0:
        print(json.dumps({"metric": "HandTrackNet training frames/sec (N=1024)", "value": round(a.batch * world * a.steps / dt, 1),
                          "unit": "frames/s", "n_gpus": world, "ms_per_step": round(dt / a.steps * 1e3, 2), "per_gpu_batch": a.batch,
                          "scaling": "weak", "graph_step": bool(getattr(tr, "graph_step", False)), "dp_mode": tr.dp_mode, "loss": float(loss["total_loss"]), "dtype": "f32", "data": "synthetic"}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
