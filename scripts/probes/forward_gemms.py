"""Device time of every library GEMM of one B=64 forward, by shape (torch.profiler, eager launches, tuned table on)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network")]
import torch
from torch.profiler import profile, ProfilerActivity
from netinit import deterministic_init, make_cfg, synthetic_frames
from hotrack_amd import fused, pointnet2_utils
from models import pointnet_utils
from models.hand_network import HandTrackNet
pointnet_utils.set_operator_backend(pointnet2_utils); pointnet_utils.set_fused_backend(fused)
FLAGS = {"track_flag": False, "test_flag": True, "save_flag": False, "IKNet_flag": False}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = HandTrackNet(make_cfg("cuda")); deterministic_init(model); model = model.cuda().eval()
d = synthetic_frames(5, B, 1024)
d = {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in d.items()}
with torch.no_grad():
    for _ in range(5): model(d, dict(FLAGS))
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(5): model(d, dict(FLAGS))
        torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0])
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::addmm", "aten::mm", "aten::_addmm_activation", "aten::linear") and e.device_time_total > 0 and e.key != "aten::linear":
        k = (e.key, str(e.input_shapes))
        rows[k][0] += e.count; rows[k][1] += e.device_time_total
tot = 0
for (k, shp), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:26s} {shp:70s} calls/fwd={n / 5:4.1f} us/call={t / n:8.1f} us/fwd={t / 5:8.1f}")
    tot += t / 5
print("GEMM device time per forward:", round(tot, 1), "us")
