"""Which torch ops launch the small element-wise kernels of the training step (eager, torch.profiler with stacks)?"""
import argparse, os, sys, collections, torch
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
N=3
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    for _ in range(N): tr.update(b)
    torch.cuda.synchronize()
# aggregate device time + launches by (op, first repo frame of the python stack)
agg=collections.defaultdict(lambda:[0,0.0])
for e in prof.events():
    if e.device_time_total<=0 or not e.name.startswith("aten::") and "Backward" not in e.name: continue
    if e.cpu_children and any(c.device_time_total>0 and c.name.startswith("aten::") for c in e.cpu_children): continue  # count leaves only
    st=[s for s in (e.stack or []) if "/network/" in s or "/hotrack_amd/" in s]
    where=st[0].split("/")[-1][:48] if st else "(autograd)"
    k=(e.name, where, str(e.input_shapes)[:60])
    agg[k][0]+=1; agg[k][1]+=e.device_time_total
rows=sorted(agg.items(), key=lambda kv:-kv[1][1])
tot=sum(v[1] for v in agg.values())/N
print("device us per step over listed ops:", round(tot))
small=[(k,v) for k,v in rows if v[1]/v[0] < 8.0]
print("small (<8us avg) ops: launches/step", sum(v[0] for k,v in small)/N, "us/step", round(sum(v[1] for k,v in small)/N))
for (name,where,shp),(n,t) in small[:70]:
    print(f"{name[:34]:34s} {where:50s} {shp:60s} n/step={n/N:5.1f} us/step={t/N:7.1f}")
