"""python network/test.py --config handtracknet_test_SimGrasp.yml [--num_points 1024]

Same entry point as the reference's network/test.py: per-sequence tracking (batch 1, frame t seeded by
frame t-1), prints data / network frames-per-second like the reference (test.py:65-98) -- here with an
explicit device synchronisation so the network time is real.  Enables the fused inference backend."""
import argparse
import logging
import os
import sys
import time

# (bench.py raises GPU_MAX_HW_QUEUES to 8 because its serving loop keeps four batches in flight on four HIP streams; this entry
# point tracks one sequence at batch 1 on ONE stream, where the runtime's default queue mapping changes nothing.  A service
# that runs several of these loops in one process sets GPU_MAX_HW_QUEUES >= its stream count in its environment.)
import torch

base_dir = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, base_dir)
sys.path.insert(0, os.path.join(base_dir, ".."))

from configs.config import get_config  # noqa: E402
from datasets.synthetic import get_dataloader  # noqa: E402
from parse_args import add_args  # noqa: E402
from trainer import Trainer  # noqa: E402


def main(args):
    cfg = get_config(args, save=False)
    logger = logging.getLogger("TestModel")
    logger.setLevel(logging.INFO)
    if torch.cuda.is_available():
        from hotrack_amd import fused
        from models import pointnet_utils
        pointnet_utils.set_fused_backend(fused)
    loader = get_dataloader(cfg, args.mode_name, length=args.synthetic_frames)
    trainer = Trainer(cfg, logger, len(loader))
    trainer.resume(len(loader))
    t_data = t_net = 0.0
    frames = 0
    acc = {}
    zero = time.time()
    for i, data in enumerate(loader):
        n = len(data) if isinstance(data, list) else 1
        frames += n
        start = time.time()
        t_data += start - zero
        loss, _ = trainer.test(data, save_flag=args.save)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        t_net += time.time() - start
        print(f"Trajectory {i}: {n} frames, network {n / (time.time() - start):8.2f} FPS")
        for k, v in loss.items():
            acc[k] = acc.get(k, 0.0) + float(v)
        zero = time.time()
    print(f"Overall, {frames:8} frames")
    print(f"Data Preprocessing: {t_data:8.2f}s {frames / max(t_data, 1e-9):8.2f}FPS")
    print(f"Network Forwarding: {t_net:8.2f}s {frames / max(t_net, 1e-9):8.2f}FPS")
    for k, v in acc.items():
        print("Test {} is {}".format(k, v / max(len(loader), 1)))


if __name__ == "__main__":
    p = add_args(argparse.ArgumentParser())
    p.add_argument("--mode_name", default="test")
    main(p.parse_args())
