"""B=1 tracking latency (the deployment shape of network/test.py): one 1024-point frame through HandTrackNet,
eager vs HIP-graph replay, fused inference path.  SURVEY.md 8(f) rank 2."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402

from netinit import deterministic_init, make_cfg, synthetic_frames  # noqa: E402
from hotrack_amd import fused, pointnet2_utils  # noqa: E402
from models import pointnet_utils  # noqa: E402
from models.hand_network import HandTrackNet  # noqa: E402

pointnet_utils.set_operator_backend(pointnet2_utils)
pointnet_utils.set_fused_backend(fused)
FLAGS = {"track_flag": False, "test_flag": True, "save_flag": False, "IKNet_flag": False}
res = {}
for B in (1, 8):
    model = HandTrackNet(make_cfg("cuda"))
    deterministic_init(model)
    model = model.cuda().eval()
    d = synthetic_frames(5, B, 1024)
    d = {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in d.items()}
    with torch.no_grad():
        for _ in range(5):
            model(d, dict(FLAGS))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            model(d, dict(FLAGS))
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 50
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = model(d, dict(FLAGS))
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 200
    res[f"B={B}"] = {"eager_ms": round(eager * 1e3, 3), "graph_ms": round(graph * 1e3, 3), "graph_frames_per_s": round(B / graph, 1)}
print(json.dumps(res))
