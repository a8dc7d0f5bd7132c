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


def run_training_leg(steps, warmup, batch, graph, rank, world, measure_allreduce=True, cfg_extra=None):
    """configs[2] per GPU on an already initialised process group (or a single process): returns the result dict (every
    rank; times are max over ranks).  Used by main() below and by bench.py's `train` leg when WORLD_SIZE > 1."""
    os.environ.setdefault("HOTRACK_DATA_ROOT", "/tmp/hotrack_bench_data")
    from configs.config import get_config
    from datasets.synthetic import make_frame
    from parse_args import add_args
    from trainer import Trainer
    args = add_args(argparse.ArgumentParser()).parse_args(["--config", "handtracknet_train_SimGrasp.yml"])
    args.num_points, args.batch_size = 1024, batch
    cfg = get_config(args, save=False)
    cfg["graph_step"] = graph  # data parallel: forward+backward graph | all-reduce of the flat gradient buffer | Adam graph
    cfg.update(cfg_extra or {})
    torch.manual_seed(0)
    tr = Trainer(cfg)
    tr.step_epoch()
    batches = [torch.utils.data.default_collate([make_frame(1000 * rank + 64 * j + i, 1024, 0.02) for i in range(batch)]) for j in range(4)]
    batches = [{k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in b.items()} for b in batches]
    on_dev = world > 1 and dist.get_backend() == "nccl"

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def reduce_max(x):
        if world == 1:
            return x
        t = torch.tensor([x], device="cuda" if on_dev else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)

    # the loop names its next batch (what network/train.py does with one batch of DataLoader lookahead): the trainer then runs
    # that batch's geometry stage on a second stream beside this batch's dense step
    for i in range(warmup):
        loss = tr.update(batches[i % 4], next_data=batches[(i + 1) % 4])
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = tr.update(batches[(warmup + i) % 4], next_data=batches[(warmup + i + 1) % 4])
    sync()
    local = time.perf_counter() - t0
    dt = reduce_max(local)
    # live dense work of a step: forward 1.344 GFLOP per frame (0.67 GMAC of grouped MLPs + heads, the discarded
    # attention excluded) x 3 (forward, data gradient, weight gradient) -- the figure DESIGN.md 5b prices the step with
    flops = 3 * 1.344e9 * batch
    launches = None
    if getattr(tr, "_graph", None) is not None:
        from hotrack_amd.graph_utils import kernel_nodes
        parts = [kernel_nodes(g) for g in (tr._graph, getattr(tr, "_graph_rest", None), tr._opt_graph, getattr(tr, "_geo_graph", None))
                 if g is not None]
        launches = sum(parts) if parts and all(p is not None for p in parts) else None
    res = {"metric": "HandTrackNet training frames/sec (N=1024)", "value": round(batch * world * steps / dt, 1),
           "launches": launches, "tflops": round(flops / (dt / steps) / 1e12, 2),
           "mfma_frac": round(flops / (dt / steps) / 1e12 / 157.3, 4),
           "unit": "frames/s", "n_gpus": world, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3), "per_gpu_batch": batch,
           "scaling": "weak", "graph_step": bool(getattr(tr, "graph_step", False)), "dp_mode": tr.dp_mode,
           "geometry_prefetch": getattr(tr, "_geo_graph", None) is not None,
           "loss": float(loss["total_loss"]), "dtype": "f32", "data": "synthetic"}
    if tr.dp_mode is not None:
        res["backend"] = dist.get_backend()  # "nccl" = RCCL on ROCm
        res["world_size_seen_by_backend"] = dist.get_world_size()
        res["bwd_segments"] = 2 if getattr(tr, "_graph_rest", None) is not None else 1
        res["dp_overlap"] = bool(tr.dp_overlap) and res["bwd_segments"] > 1
        if measure_allreduce and tr.dp_mode == "flat" and tr._flat is not None:
            # the gradient exchange alone: the same flat buffers the step all-reduces (one per backward segment), K times back to
            # back; the LAST segment's is the one a step cannot hide (the others travel beside the next segment's backward)
            sync()
            t0 = time.perf_counter()
            for _ in range(20):
                tr._allreduce_flat()
            sync()
            res["allreduce_us"] = round(reduce_max(time.perf_counter() - t0) / 20 * 1e6, 1)
            res["segment_bytes"] = [int(f.numel() * f.element_size()) for f in tr._flat]
            # bytes the per-step multi-tensor copy still moves into the flat buffers (gradients whose producers do not write there)
            res["moved_bytes"] = [int(4 * tr._segs[s_].get("moved", 0)) for s_ in sorted(tr._segs)]
            res["bytes"] = int(sum(res["segment_bytes"]))
            last = sorted(tr._segs)[-1]
            sync()
            t0 = time.perf_counter()
            for _ in range(20):
                tr._finish_exchange([tr._exchange(last)])
            sync()
            res["exposed_allreduce_us"] = round(reduce_max(time.perf_counter() - t0) / 20 * 1e6, 1)
            # ring all-reduce moves 2 (N-1)/N of the buffer per rank
            res["allreduce_busbw_GBs"] = round(2 * (world - 1) / world * res["bytes"] / (res["allreduce_us"] * 1e-6) / 1e9, 2)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--graph", action="store_true", help="whole-step HIP graph (single GPU)")
    ap.add_argument("--dp-selftest", action="store_true",
                    help="ONE rank with a one-rank process group and dp=flat: the segmented backward + exchange path end to end "
                         "on one GPU (what it costs beside the single-graph step; the all-reduce is RCCL's one-rank path)")
    ap.add_argument("--segments", type=int, default=None, help="dp=flat: backward segments (default 1; 2 = everything after the backbone | the backbone)")
    ap.add_argument("--no-overlap", action="store_true", help="dp=flat: exchanges in stream order (A/B)")
    a = ap.parse_args()
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local % torch.cuda.device_count())
    if os.environ.get("PN2_BENCH_FAIL_TRAIN_LEG", "") == str(rank):  # harness self-test: this rank's training leg dies
        raise RuntimeError("injected failure of the training leg on rank %d (PN2_BENCH_FAIL_TRAIN_LEG)" % rank)
    extra = {}
    if a.segments is not None:
        extra["bwd_segments"] = a.segments
    if a.no_overlap:
        extra["dp_overlap"] = False
    if world > 1 or a.dp_selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(os.environ.get("PN2_DIST_BACKEND", "nccl"), rank=rank, world_size=world)  # gloo: several ranks on one GPU (self-test)
        if world == 1:
            extra["dp_force"] = "flat"
    res = run_training_leg(a.steps, a.warmup, a.batch, a.graph, rank, world, cfg_extra=extra)
    if rank == 0:
        print(json.dumps(res))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
