"""FPS 1024 -> 256 per launch (us) for B = 1 / 64 under PN2_FPS_THREADS (child processes: the variable is read per call, the
LDS-size attribute per instantiation)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path[:0] = [ROOT, os.path.join(ROOT, "network")]
    import torch
    from hotrack_amd import pointnet2_utils as ops
    for B in (1, 64):
        x = torch.rand(B, 1024, 3, device="cuda")
        ref = None
        f = lambda: ops.furthest_point_sample(x, 256)
        out = f(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            f()
            with torch.cuda.graph(g, stream=s):
                for _ in range(10): out = f()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); torch.cuda.synchronize(); a.record()
        for _ in range(20): g.replay()
        b.record(); torch.cuda.synchronize()
        print("threads=%s B=%d %.2f us  checksum %d" % (os.environ.get("PN2_FPS_THREADS", "default"), B, a.elapsed_time(b) / 200 * 1e3, int(out.sum())))
else:
    for t in ("", "64", "128", "256", "512"):
        env = dict(os.environ)
        if t: env["PN2_FPS_THREADS"] = t
        print(subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout, end="")
