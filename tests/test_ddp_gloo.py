"""CPU, world_size 2, gloo: the data-parallel training path (Trainer + DistributedDataParallel).
Checks what RCCL will do on the GPUs: replicas stay identical after optimiser steps, gradients are the
mean over ranks, and the never-used attention parameters keep grad=None (so Adam's weight decay never
touches them -- the single-GPU reference behaviour)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp, q, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HOTRACK_DATA_ROOT=tmp)
    sys.path[:0] = [ROOT, os.path.join(ROOT, "network"), os.path.join(ROOT, "tests")]
    torch.cuda.is_available = lambda: False
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import argparse
    from configs.config import get_config
    from datasets.synthetic import make_frame
    from models import pointnet_utils
    from oracle import torch_ops
    from parse_args import add_args
    from trainer import Trainer
    pointnet_utils.set_operator_backend(torch_ops)
    a = add_args(argparse.ArgumentParser()).parse_args(["--config", "handtracknet_train_SimGrasp.yml"])
    a.num_points, a.batch_size = 256, 2
    cfg = get_config(a, save=False)
    cfg["dp"] = mode
    torch.manual_seed(rank if mode == "flat" else 0)  # "flat" must broadcast rank 0's weights itself (DDP's constructor does)
    tr = Trainer(cfg)
    assert (tr.ddp is not None) == (mode == "ddp") and tr.dp_mode == mode
    for m in tr.model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    tr.step_epoch()

    def batch(seed0):
        frames = [make_frame(seed0 + i, 256, 0.02) for i in range(2)]
        return torch.utils.data.default_collate(frames)

    w0 = tr.model.transt.s12.attn.in_proj_weight.detach().clone()  # never used -> must never change
    losses = []
    for it in range(2):
        losses.append(float(tr.update(batch(100 * rank + 10 * it))["total_loss"]))  # different shard per rank
    named = dict(tr.model.named_parameters())
    none_names = sorted(n for n, p in named.items() if p.grad is None)
    flat = torch.cat([p.detach().flatten() for p in tr.model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    # single-process reference gradient for the first step of THIS rank, to check the averaging
    q.put({"rank": rank, "replicas_equal": bool(torch.equal(gathered[0], gathered[1])),
           "n_none": len(none_names), "numel_none": sum(named[n].numel() for n in none_names),
           "unused_unchanged": bool(torch.equal(w0, tr.model.transt.s12.attn.in_proj_weight.detach())),
           "losses": losses, "finite": bool(torch.isfinite(flat).all())})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", ["ddp", "flat"])
def test_ddp_two_ranks_gloo(tmp_path, mode):
    """mode "ddp": torch DistributedDataParallel; "flat": Trainer's one-all-reduce-per-step mode (the one that runs as HIP
    graphs on the GPUs)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert r["replicas_equal"] and r["finite"] and r["unused_unchanged"]
        assert r["n_none"] == 30 and r["numel_none"] == 3746944  # 3.75 M parameters never get a gradient
    assert res[0]["losses"] != res[1]["losses"]  # the ranks really saw different shards
