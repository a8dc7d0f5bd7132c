"""hand_frame launch at B = 1 / 64 in a captured graph of 20 back-to-back launches (us per launch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network"), os.path.join(ROOT, "tests")]
import torch
from netinit import synthetic_frames
from hotrack_amd import ext
for B in (1, 64):
    d = synthetic_frames(3, B, 1024)
    pts, kp, palm = d["hand_points"].cuda(), d["jittered_hand_kp"].cuda(), d["gt_hand_pose"]["palm_template"].cuda()
    idx = torch.tensor([0, 1, 5, 9, 13, 17], dtype=torch.int32, device="cuda")
    f = lambda: ext.hand_frame(palm, kp, idx, pts, 0.2)
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20): out = f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    a.record()
    for _ in range(20): g.replay()
    b.record(); torch.cuda.synchronize()
    print("B=%d hand_frame %.2f us per launch" % (B, a.elapsed_time(b) / 400 * 1e3))
