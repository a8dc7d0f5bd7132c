import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from hotrack_amd import pointnet2_hip as native
g = torch.Generator(device="cuda"); g.manual_seed(0)
def timeit_graph(fn, iters=30):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            fn()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(iters): gr.replay()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (b, C, N, P, S) in [(8, 67, 8192, 2048, 64), (8, 3, 8192, 2048, 64), (1, 67, 8192, 2048, 64), (8, 33, 8192, 1024, 64), (64, 67, 8192, 2048, 64)]:
    idx = torch.randint(0, N, (b, P, S), device="cuda", dtype=torch.int32, generator=g)
    go = torch.randn(b, C, P, S, device="cuda", generator=g)
    gp = torch.zeros(b, C, N, device="cuda")
    us = timeit_graph(lambda: native.group_points_grad_wrapper(b, C, N, P, S, go, idx, gp))
    nb = b * (4 * P * S + 4 * C * min(N, P * S) + 4 * C * P * S)
    print(os.environ.get("PN2_SCM_CC", "auto"), (b, C, N, P, S), round(us, 1), "us", round(nb / us / 1e3, 1), "GB/s")
