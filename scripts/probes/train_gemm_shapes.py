"""Which GEMM shapes carry the training step?  torch.profiler with input shapes over a few eager steps (tuned table on)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network")]
import torch
os.environ.setdefault("HOTRACK_DATA_ROOT", "/tmp/hotrack_bench_data")
from configs.config import get_config
from datasets.synthetic import make_frame
from parse_args import add_args
from trainer import Trainer
args = add_args(argparse.ArgumentParser()).parse_args(["--config", "handtracknet_train_SimGrasp.yml"])
args.num_points, args.batch_size = 1024, 32
cfg = get_config(args, save=False)
tr = Trainer(cfg); tr.step_epoch()
b = torch.utils.data.default_collate([make_frame(i, 1024, 0.02) for i in range(32)])
b = {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in b.items()}
for _ in range(5): tr.update(b)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(5): tr.update(b)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::mm", "aten::addmm", "aten::bmm", "aten::linear", "aten::_addmm_activation") and e.device_time_total > 0:
        rows.append((e.device_time_total / 5, e.count / 5, e.key, str(e.input_shapes)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows if r[2] != "aten::linear")
print("total GEMM device us/step (mm+addmm+bmm):", round(tot, 1))
for r in rows[:40]:
    print("%8.1f us/step  x%4.1f  %-12s %s" % r)
