"""Worker of tests/test_gpu_train_full.py::test_flat_exchange_*: ONE rank with a one-rank RCCL process group -- the graph-captured
data-parallel step (dp = flat) against the single-process graph-captured step on the same batches.  Prints one JSON line."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network")]
import torch  # noqa: E402


def main():
    segments, port = int(sys.argv[1]), sys.argv[2]
    from configs.config import get_config
    from datasets.synthetic import make_frame
    from parse_args import add_args
    from trainer import Trainer
    import torch.distributed as dist

    def build(extra):
        a = add_args(argparse.ArgumentParser()).parse_args(["--config", "handtracknet_train_SimGrasp.yml"])
        a.num_points, a.batch_size = 1024, 8
        cfg = get_config(a, save=False)
        cfg["graph_step"] = True
        cfg.update(extra)
        torch.manual_seed(0)
        tr = Trainer(cfg)
        tr.step_epoch()
        for m in tr.model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        return tr

    batches = [torch.utils.data.default_collate([make_frame(9000 + 16 * j + i, 1024, 0.02) for i in range(8)]) for j in range(2)]
    batches = [{k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in b.items()} for b in batches]
    solo = build({})
    l_solo = [float(solo.update(batches[i % 2], next_data=batches[(i + 1) % 2])["total_loss"]) for i in range(4)]
    p_solo = torch.cat([p.detach().flatten() for p in solo.model.parameters()]).double()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    dist.init_process_group("nccl", rank=0, world_size=1)
    dp = build({"dp_force": "flat", "bwd_segments": segments})
    assert dp.dp_mode == "flat"
    l_dp = [float(dp.update(batches[i % 2], next_data=batches[(i + 1) % 2])["total_loss"]) for i in range(4)]
    p_dp = torch.cat([p.detach().flatten() for p in dp.model.parameters()]).double()
    flats = dp._flat
    in_flat = all(any(f.data_ptr() <= p.grad.data_ptr() < f.data_ptr() + 4 * f.numel() for f in flats)
                  for p in dp.model.parameters() if p.grad is not None)
    print(json.dumps({"graph_step": bool(dp.graph_step), "l_solo": l_solo, "l_dp": l_dp, "param_diff": float((p_solo - p_dp).abs().max()),
                      "flat_floats": [int(f.numel()) for f in flats], "moved_floats": [int(dp._segs[s].get("moved", 0)) for s in sorted(dp._segs)],
                      "grads_in_flat": bool(in_flat), "graphs": [dp._graph is not None, dp._graph_rest is not None, dp._opt_graph is not None],
                      "n_none": sum(p.grad is None for p in dp.model.parameters())}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
