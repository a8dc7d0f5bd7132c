import argparse, os, sys, torch
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0]=[ROOT, os.path.join(ROOT,"network")]
os.environ.setdefault("HOTRACK_DATA_ROOT","/tmp/hotrack_bench_data")
from configs.config import get_config
from datasets.synthetic import make_frame
from parse_args import add_args
from trainer import Trainer
args=add_args(argparse.ArgumentParser()).parse_args(["--config","handtracknet_train_SimGrasp.yml"])
args.num_points,args.batch_size=1024,32
cfg=get_config(args,save=False)
tr=Trainer(cfg); tr.step_epoch()
b=torch.utils.data.default_collate([make_frame(i,1024,0.02) for i in range(32)])
b={k:(v.cuda() if torch.is_tensor(v) else {kk:vv.cuda() for kk,vv in v.items()}) for k,v in b.items()}
for _ in range(5): tr.update(b)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3): tr.update(b)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=35, max_name_column_width=70))
