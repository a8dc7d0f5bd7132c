"""When does hipGraphLaunch return?  Graphs of N element-wise kernels (~`us` microseconds each), optionally with a device-to-
device memcpy node in the middle: host time of each replay() while the previous replay is still running, against the graph's
device time.  (Round 6: the training step's dense graph kept the host inside hipGraphLaunch for ~2.3 of its 2.65 ms.)"""
import os
import sys
import time

import torch

dev = torch.device("cuda")
x = torch.ones(int(os.environ.get("PROBE_ELEMS", 1 << 24)), device=dev)  # 64 MB: ~25 us per in-place multiply
a = torch.ones(1024, device=dev)
b = torch.zeros(1024, device=dev)


def build(n, memcpy_at=-1, big_args=False):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            x.mul_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for i in range(n):
                if i == memcpy_at:
                    b.copy_(a)
                x.mul_(1.0)
    return g, s


for n, mc in ((20, -1), (60, -1), (100, -1), (150, -1), (200, -1), (400, -1), (200, 100), (200, 190)):
    g, s = build(n, mc)
    with torch.cuda.stream(s):
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        t_first = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_dev = time.perf_counter() - t0
        hosts = []
        t0 = time.perf_counter()
        for _ in range(6):
            h0 = time.perf_counter()
            g.replay()
            hosts.append(time.perf_counter() - h0)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / 6
    print(f"nodes {n:4d} memcpy_at {mc:4d}: device {t_dev * 1e3:7.3f} ms/graph | replay() host: idle queue {t_first * 1e3:6.3f} ms, "
          f"back to back {[round(h * 1e3, 3) for h in hosts]} | steady {per * 1e3:.3f} ms/graph")
