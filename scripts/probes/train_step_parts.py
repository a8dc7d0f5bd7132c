"""Where a graph-captured training step's time goes OUTSIDE its kernels, without a profiler (rocprofv3's kernel tracing keeps
the host inside hipGraphLaunch for the length of the graph, so its timelines show hand-over gaps the free-running loop may not
have): the full update() loop against (a) the dense graph replayed back to back, (b) batch hand-over + dense graph, no
geometry prefetch, (c) the geometry graph alone.  usage: python scripts/probes/train_step_parts.py [--dp]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network")]
os.environ.setdefault("HOTRACK_DATA_ROOT", "/tmp/hotrack_bench_data")
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dp", action="store_true", help="one-rank process group, dp = flat")
    ap.add_argument("--segments", type=int, default=1)
    ap.add_argument("--iters", type=int, default=40)
    a = ap.parse_args()
    from configs.config import get_config
    from datasets.synthetic import make_frame
    from parse_args import add_args
    from trainer import Trainer
    args = add_args(argparse.ArgumentParser()).parse_args(["--config", "handtracknet_train_SimGrasp.yml"])
    args.num_points, args.batch_size = 1024, 32
    cfg = get_config(args, save=False)
    cfg["graph_step"] = True
    if a.dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1)
        cfg["dp_force"], cfg["bwd_segments"] = "flat", a.segments
    torch.manual_seed(0)
    tr = Trainer(cfg)
    tr.step_epoch()
    batches = [torch.utils.data.default_collate([make_frame(64 * j + i, 1024, 0.02) for i in range(32)]) for j in range(4)]
    batches = [{k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in b.items()} for b in batches]
    n = a.iters

    def timed(fn, label):
        for i in range(6):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{label:58s} {dt / n * 1e3:7.3f} ms / iteration   (host enqueue {t_host / n * 1e3:6.3f} ms)")
        return dt / n

    full = timed(lambda i: tr.update(batches[i % 4], next_data=batches[(i + 1) % 4]), "update(data, next_data)  [prefetch]")
    timed(lambda i: tr.update(batches[i % 4]), "update(data)             [geometry in line]")
    timed(lambda i: tr.update(batches[i % 4], next_data=batches[(i + 1) % 4]), "update(data, next_data)  [prefetch] again")
    torch.cuda.synchronize()
    dense = timed(lambda i: tr._graph.replay(), "dense graph alone, back to back")
    if tr._opt_graph is not None:
        timed(lambda i: (tr._graph.replay(), tr._opt_graph.replay()), "dense graph + optimiser graph, back to back")
        timed(lambda i: (tr._graph.replay(), tr._allreduce_flat(), tr._opt_graph.replay()), "dense graph + exchange + optimiser graph")
    timed(lambda i: (tr._copy_leaves(tr._static, batches[i % 4]), tr._graph.replay()), "batch hand-over + dense graph")
    if tr._geo_graph is not None:
        timed(lambda i: tr._geo_graph.replay(), "geometry graph alone, back to back")
    print(f"full step - dense graph = {(full - dense) * 1e3:.0f} us")


if __name__ == "__main__":
    main()
