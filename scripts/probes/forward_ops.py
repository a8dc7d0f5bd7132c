import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network"), os.path.join(ROOT, "tests")]
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
    for _ in range(3): model(d, dict(FLAGS))
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        model(d, dict(FLAGS)); torch.cuda.synchronize()
from torch.autograd import DeviceType
print("device kernels in one forward:", sum(1 for e in prof.events() if e.device_type == DeviceType.CUDA))
#print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=40, max_shapes_column_width=60))
