"""Per-pick time of the streamed-coordinate sampling kernel (16385 .. 65536 points) against the HBM-temp kernel it replaces
(PN2_FPS_NO_STREAM=1 in a child process: the switch is read once per process).  GPU box."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run():
    import torch
    from hotrack_amd import pointnet2_hip as native
    res = {}
    for B, N, M in [(1, 20000, 256), (8, 32768, 256), (1, 65536, 256), (8, 65536, 256), (64, 32768, 128)]:
        x = torch.rand(B, N, 3, device="cuda")
        out = torch.empty(B, M, dtype=torch.int32, device="cuda")
        temp = torch.full((B, N), 1e10, device="cuda")

        def go():
            temp.fill_(1e10)
            native.furthest_point_sampling_wrapper(B, N, M, x, temp, out)
        go()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            go()
        e.record()
        torch.cuda.synchronize()
        res["B=%d N=%d M=%d" % (B, N, M)] = round(s.elapsed_time(e) / 3 * 1e3 / M, 3)
    return res


if __name__ == "__main__":
    if len(sys.argv) > 1:
        print(json.dumps(run()))
    else:
        a = json.loads(subprocess.check_output([sys.executable, __file__, "child"], env=dict(os.environ)).decode().strip().split("\n")[-1])
        b = json.loads(subprocess.check_output([sys.executable, __file__, "child"], env=dict(os.environ, PN2_FPS_NO_STREAM="1")).decode().strip().split("\n")[-1])
        print(json.dumps({"us_per_pick": {k: {"streamed": a[k], "hbm_temp": b[k]} for k in a}}, indent=1))
