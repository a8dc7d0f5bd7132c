"""Host-side cost of Trainer.update() with and without the geometry prefetch: per-call wall time of the enqueue (no sync) and
the steady-state step time.  usage: python scripts/probes/train_host_timing.py"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HOTRACK_DATA_ROOT", "/tmp/hotrack_bench_data")
import torch  # noqa: E402


def main():
    from configs.config import get_config
    from datasets.synthetic import make_frame
    from parse_args import add_args
    from trainer import Trainer
    args = add_args(argparse.ArgumentParser()).parse_args(["--config", "handtracknet_train_SimGrasp.yml"])
    args.num_points, args.batch_size = 1024, 32
    cfg = get_config(args, save=False)
    cfg["graph_step"] = True
    torch.manual_seed(0)
    tr = Trainer(cfg)
    tr.step_epoch()
    batches = [torch.utils.data.default_collate([make_frame(64 * j + i, 1024, 0.02) for i in range(32)]) for j in range(4)]
    batches = [{k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in b.items()} for b in batches]
    main_stream = torch.cuda.Stream() if os.environ.get("PROBE_SIDE_STREAM", "0") == "1" else torch.cuda.current_stream()
    torch.cuda.set_stream(main_stream)
    for mode in ("prefetch", "inline", "prefetch"):
        for i in range(6):
            tr.update(batches[i % 4], next_data=batches[(i + 1) % 4] if mode == "prefetch" else None)
        torch.cuda.synchronize()
        host = []
        t0 = time.perf_counter()
        for i in range(30):
            h0 = time.perf_counter()
            tr.update(batches[i % 4], next_data=batches[(i + 1) % 4] if mode == "prefetch" else None)
            host.append(time.perf_counter() - h0)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        host.sort()
        print(f"{mode:9s} step {dt / 30 * 1e3:.3f} ms | host enqueue per call: median {host[15] * 1e3:.3f} ms, max {host[-1] * 1e3:.3f} ms, "
              f"all 30 enqueued after {t_enq * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
