"""Which stage of a training step breaks HIP-graph capture?  forward / +loss / +backward / +optimizer, each in a fresh process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import argparse, os, sys
ROOT = %r
sys.path[:0] = [ROOT, os.path.join(ROOT, "network")]
import torch
os.environ.setdefault("HOTRACK_DATA_ROOT", "/tmp/hotrack_bench_data")
from configs.config import get_config
from datasets.synthetic import make_frame
from parse_args import add_args
from trainer import Trainer
stage = int(sys.argv[1])
args = add_args(argparse.ArgumentParser()).parse_args(["--config", "handtracknet_train_SimGrasp.yml"])
args.num_points, args.batch_size = 1024, 8
cfg = get_config(args, save=False); cfg["graph_step"] = True
torch.manual_seed(0)
tr = Trainer(cfg); tr.step_epoch(); tr.model.train()
b = torch.utils.data.default_collate([make_frame(i, 1024, 0.02) for i in range(8)])
b = {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in b.items()}
flags = tr.init_flag_dict()
def run():
    ret = tr.model(b, flags)
    if stage >= 1:
        ld, _ = tr.model.compute_loss(b, ret, flags); ld = tr.summarize_losses(ld)
    if stage >= 2: ld["total_loss"].backward()
    if stage >= 3: tr.optimizer.step()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        tr.optimizer.zero_grad(set_to_none=True); run()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
tr.optimizer.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g): run()
    g.replay(); torch.cuda.synchronize(); print("stage", stage, "captured + replayed OK")
except Exception as e:
    import traceback; tb = traceback.format_exc().splitlines()
    print("stage", stage, "FAILED:", str(e).splitlines()[0]); print("\n".join(l for l in tb if "/root/repo" in l or "File" in l)[-1500:])
''' % ROOT
for st in range(4):
    r = subprocess.run([sys.executable, "-c", code, str(st)], capture_output=True, text=True)
    print(r.stdout.strip()[-1800:] or r.stderr.strip()[-800:])
