import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network"), os.path.join(ROOT, "tests")]
import torch
from netinit import deterministic_init, make_cfg, synthetic_frames
from hotrack_amd import fused, pointnet2_utils
from models import pointnet_utils
from models.hand_network import HandTrackNet
pointnet_utils.set_operator_backend(pointnet2_utils); pointnet_utils.set_fused_backend(fused)
FLAGS = {"track_flag": False, "test_flag": True, "save_flag": False, "IKNet_flag": False}
model = HandTrackNet(make_cfg("cuda")); deterministic_init(model); model = model.cuda().eval()
for B in (1, 64):
    d = synthetic_frames(5, B, 1024)
    d = {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in d.items()}
    res = {}
    with torch.no_grad():
        model(d, dict(FLAGS))
        graphs = {}
        for mode in (True, False):
            model._fast.two_level_fps = mode
            for _ in range(3): model(d, dict(FLAGS))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g): model(d, dict(FLAGS))
            graphs[mode] = g
        for rep in range(3):
            for mode in (True, False):
                g = graphs[mode]
                for _ in range(5): g.replay()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(300): g.replay()
                torch.cuda.synchronize(); res.setdefault(mode, []).append((time.perf_counter() - t0) / 300 * 1e3)
    print(B, {("two_level" if k else "plain"): [round(x, 4) for x in v] for k, v in res.items()})
