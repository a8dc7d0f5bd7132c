"""python network/train.py --config handtracknet_train_SimGrasp.yml [--num_points 1024] ...
   torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 network/train.py --config ...

Same entry point / flags as the reference's network/train.py; data = seeded synthetic frames when the
SimGrasp directory is absent.  Under torchrun each rank owns one GPU and `batch_size` clouds per step
(weak scaling); gradients are all-reduced over RCCL."""
import argparse
import logging
import os
import sys

import torch
import torch.distributed as dist

base_dir = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, base_dir)
sys.path.insert(0, os.path.join(base_dir, ".."))

from configs.config import get_config  # noqa: E402
from datasets.synthetic import get_dataloader  # noqa: E402
from parse_args import add_args  # noqa: E402
from trainer import Trainer  # noqa: E402


def main(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count())
        # RCCL (backend "nccl") on the GPUs; PN2_DIST_BACKEND=gloo lets several ranks share one GPU (plumbing self-test only:
        # scripts/scale_selftest.sh -- RCCL refuses two ranks on one device)
        dist.init_process_group(os.environ.get("PN2_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo"))
    cfg = get_config(args)
    log_dir = os.path.join(cfg["experiment_dir"], "log")
    os.makedirs(log_dir, exist_ok=True)
    logger = logging.getLogger("TrainModel")
    logger.setLevel(logging.INFO)
    fh = logging.FileHandler(os.path.join(log_dir, "log.txt"))
    fh.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
    logger.addHandler(fh)

    train_loader = get_dataloader(cfg, "train", shuffle=True, num_workers=args.num_workers, distributed=world > 1,
                                  length=args.synthetic_frames)
    test_loader = get_dataloader(cfg, "test")
    trainer = Trainer(cfg, logger, len(train_loader))
    start = trainer.resume(len(train_loader))
    for epoch in range(start, cfg["total_epoch"]):
        trainer.step_epoch()
        if world > 1:
            train_loader.sampler.set_epoch(epoch)
        acc, n = {}, 0
        it = iter(train_loader)
        data = next(it, None)
        while data is not None:
            # one batch of lookahead: with graph_step the trainer runs the NEXT batch's geometry stage (sampling, neighbour
            # searches) on a second stream beside this batch's dense step
            nxt = None if (args.max_iters and n + 1 >= args.max_iters) else next(it, None)
            loss = trainer.update(data, next_data=nxt)
            for k, v in loss.items():
                acc[k] = acc.get(k, 0.0) + float(v)
            n += 1
            data = nxt
        for k, v in acc.items():
            trainer.log_string("Train {} is {}".format(k, v / max(n, 1)))
        trainer.log_string("world_size %d, %d iterations" % (world, n))
        if (epoch + 1) % cfg["freq"]["save"] == 0:
            trainer.save()
        acc, n = {}, 0
        for data in test_loader:
            loss, _ = trainer.test(data)
            for k, v in loss.items():
                acc[k] = acc.get(k, 0.0) + float(v)
            n += 1
            if args.max_iters and n >= args.max_iters:
                break
        for k, v in acc.items():
            trainer.log_string("Test {} is {}".format(k, v / max(n, 1)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(add_args(argparse.ArgumentParser()).parse_args())
