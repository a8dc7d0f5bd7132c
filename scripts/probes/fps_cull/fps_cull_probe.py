"""Timing probes of fps_cull.hip (PN2_FPC_PROBE bit mask, read once per process): run as
    for p in 0 1 2 3 4 8; do PN2_FPC_PROBE=$p python scripts/probes/fps_cull_probe.py; done"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hotrack_amd import pointnet2_utils as ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
xyz = torch.rand(64, 8192, 3, device="cuda", generator=g)
for _ in range(2):
    ops.furthest_point_sample(xyz, 2048)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(3):
    ops.furthest_point_sample(xyz, 2048)
e.record()
torch.cuda.synchronize()
print("PN2_FPC_PROBE=%s PN2_FPS_CULL=%s: %.3f ms (%.3f us per pick)" % (os.environ.get("PN2_FPC_PROBE", "0"), os.environ.get("PN2_FPS_CULL", "1"),
      s.elapsed_time(e) / 3, s.elapsed_time(e) / 3 / 2047 * 1e3))
