"""Secondary legs of bench.py, each run as a CHILD process after the headline regions so that nothing they do -- an exception,
a hang (bounded by a timeout), a crash -- can cost the headline line:

    train    BASELINE configs[2] per GPU: 32 x 1024 training step as HIP graphs (scripts/bench_train.py; under WORLD_SIZE > 1
             the children of all ranks form their own process group on a fresh port and the step carries the RCCL all-reduce)
    stress   BASELINE configs[4] per GPU: 64 clouds x 8192 points, npoint 2048, nsample 64, C = 64: FPS, ball query (r = 0.1 /
             0.2), fused SA kernel, the whole level serial and with two batches in flight
    latency  B = 1 tracking-shaped forward as a HIP graph (ms per frame)

`python scripts/bench_legs.py <leg>` prints ONE JSON object; bench.py calls run_child()."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_child(cmd, timeout_s, env=None, expect_json=True):
    """Run a leg; returns its last JSON line as a dict, or {"error": ...}.  The child is killed by its exact pid on timeout.
    expect_json=False (a rank whose child prints nothing): {} on exit code 0."""
    t0 = time.perf_counter()
    try:
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env)
    except OSError as exc:
        return {"error": f"spawn failed: {exc}"}
    try:
        out, err = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        p.kill()
        out, err = p.communicate()
        return {"error": f"timeout after {timeout_s:.0f} s", "stderr_tail": (err or "")[-300:]}
    lines = [l for l in (out or "").splitlines() if l.startswith("{")]
    if p.returncode == 0 and not expect_json:
        return {}
    if p.returncode != 0 or not lines:
        return {"error": f"exit code {p.returncode}", "stderr_tail": (err or out or "")[-400:]}
    try:
        res = json.loads(lines[-1])
    except ValueError as exc:
        return {"error": f"unparsable output: {exc}"}
    res["leg_wall_s"] = round(time.perf_counter() - t0, 1)
    return res


def leg_cmd(name, *extra):
    return [sys.executable, os.path.abspath(__file__), name, *map(str, extra)]


# ---------------------------------------------------------------------------------------------------------------------------
def _timeit(fn, iters=10, warm=2):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def stress():
    """configs[4] per GPU (reference composition: pointnet_utils.py:389-403 over sampling_gpu.cu:94-209, ball_query_gpu.cu:9-45)."""
    import torch
    sys.path.insert(0, ROOT)
    from hotrack_amd import ext
    from hotrack_amd import pointnet2_utils as ops
    B, N, S, K, C = 64, 8192, 2048, 64, 64
    g = torch.Generator(device="cuda").manual_seed(0)
    W1 = torch.randn(64, C + 3, device="cuda", generator=g) * 0.2
    b1 = torch.randn(64, device="cuda", generator=g)
    W2 = torch.randn(64, 64, device="cuda", generator=g) * 0.2
    b2 = torch.randn(64, device="cuda", generator=g)
    W3 = torch.randn(128, 64, device="cuda", generator=g) * 0.2
    b3 = torch.randn(128, device="cuda", generator=g)
    w1f_t, wx = W1[:, :C].t().contiguous(), W1[:, C:].contiguous()
    batches = [(torch.rand(B, N, 3, device="cuda", generator=g), torch.randn(B, C, N, device="cuda", generator=g)) for _ in range(2)]
    xyz, feat = batches[0]
    res = {"workload": "configs[4] per GPU: %d clouds x %d points, npoint %d, nsample %d, C %d" % (B, N, S, K, C)}
    res["fps_ms"] = round(_timeit(lambda: ops.furthest_point_sample(xyz, S), iters=3, warm=1), 4)
    fps = ops.furthest_point_sample(xyz, S)
    new_xyz = ext.gather_rows(xyz, fps)
    res["ball_ms_r01"] = round(_timeit(lambda: ops.ball_query(0.1, K, xyz, new_xyz)), 4)
    res["ball_ms_r02"] = round(_timeit(lambda: ops.ball_query(0.2, K, xyz, new_xyz)), 4)
    idx = ops.ball_query(0.2, K, xyz, new_xyz)
    a1f = torch.matmul(feat.transpose(1, 2), w1f_t)
    t = _timeit(lambda: ext.sa_mlp_max(idx, W2, b2, W3, b3, a1f=a1f, xyz=xyz, cxyz=new_xyz, wx=wx, b1=b1))
    res["sa_ms"] = round(t, 4)
    res["sa_mfma_frac"] = round(2.0 * B * S * K * (64 * 64 + 64 * 128) / (t * 1e-3) / 1e12 / 157.3, 4)

    def level(x, f):  # sample -> gather -> query -> per-point layer-1 GEMM -> fused MLP + max: one set-abstraction level
        i = ops.furthest_point_sample(x, S)
        c = ext.gather_rows(x, i)
        j = ops.ball_query(0.2, K, x, c)
        a1 = torch.matmul(f.transpose(1, 2), w1f_t)
        return ext.sa_mlp_max(j, W2, b2, W3, b3, a1f=a1, xyz=x, cxyz=c, wx=wx, b1=b1)

    res["level_ms"] = round(_timeit(lambda: level(xyz, feat), iters=4, warm=1), 4)
    # two batches in flight, one HIP stream each: FPS of one batch (64 workgroups, latency chain) beside the ball query + SA
    # kernel of the other.  Time per level = wall time of 2 n levels / 2 n.
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [None, None]

    def pair(n):
        for it in range(n):
            for s in (0, 1):
                with torch.cuda.stream(streams[s]):
                    outs[s] = level(*batches[s])
    pair(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pair(4)
    torch.cuda.synchronize()
    res["level_ms_two_streams"] = round((time.perf_counter() - t0) / 8 * 1e3, 4)
    ref = level(*batches[1])
    torch.cuda.synchronize()
    ok = bool(torch.equal(ref, outs[1]))
    # Round 5: the level as a producer / consumer pair of streams.  The geometry of batch t + 1 (sampling,
    # gather, ball query: one workgroup per cloud for 1.36 ms, i.e. 64 of 256 CUs) runs on a HIGH-PRIORITY stream of its own
    # beside the dense half of batch t (layer-1 GEMM + the persistent SA kernel), and the SA grid is sized to leave those CUs
    # free (pn2x_sa_set_compute_units): a workgroup of that kernel fills a CU, so with all 256 taken the sampling of the next
    # batch could only start when the SA kernel ended.  The dense stream waits for the geometry stream's event (the direction
    # that costs nothing on this runtime); the geometry stream never waits for the dense one (its inputs are the raw clouds).
    lo, hi = torch.cuda.Stream.priority_range()
    sg, sd = torch.cuda.Stream(priority=hi), torch.cuda.Stream(priority=lo)

    def split(n, keep, ball_on_g=False):
        last = None
        for it in range(n):
            x, f = batches[it % 2]
            with torch.cuda.stream(sg):
                i = ops.furthest_point_sample(x, S)
                c = ext.gather_rows(x, i)
                if ball_on_g:
                    j = ops.ball_query(0.2, K, x, c)
                ev = torch.cuda.Event()
                ev.record(sg)
            with torch.cuda.stream(sd):
                sd.wait_event(ev)
                if not ball_on_g:  # the ball query wants the whole chip for 0.2 ms: beside the SA grid it is squeezed onto the CUs left free
                    j = ops.ball_query(0.2, K, x, c)
                a1 = torch.matmul(f.transpose(1, 2), w1f_t)
                last = ext.sa_mlp_max(j, W2, b2, W3, b3, a1f=a1, xyz=x, cxyz=c, wx=wx, b1=b1)
            keep.append((i, c, j, a1, last))  # (alive until the final synchronize: no block is reused across the two streams)
        return last

    best = None
    sweep = {}
    for ball_on_g in (False, True):
        for cus in (0, 224, 208, 192, 176):
            ext.sa_set_compute_units(cus)
            keep = []
            split(2, keep, ball_on_g)
            torch.cuda.synchronize()
            keep = []
            t0 = time.perf_counter()
            out = split(8, keep, ball_on_g)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 8 * 1e3
            sweep["%d%s" % (cus or 256, "+ball_beside_sampling" if ball_on_g else "")] = round(ms, 4)
            ok = ok and bool(torch.equal(ref, out))  # (8 levels: the last one is batch 1's)
            if best is None or ms < best[0]:
                best = (ms, cus or 256)
    ext.sa_set_compute_units(0)
    res["level_ms_pipelined"] = round(best[0], 4)
    res["pipelined_sa_cus"] = best[1]
    res["pipelined_by_sa_cus"] = sweep
    res["pipelined_equals_serial"] = ok
    res["clouds_per_s_pipelined"] = round(B / (res["level_ms_pipelined"] * 1e-3), 1)
    return res


def latency():
    """B = 1 frame (reference loop: track_network.py:159-217, test.py:65-98 with batch_size 1) as one replayed HIP graph."""
    import torch
    sys.path[:0] = [ROOT, os.path.join(ROOT, "network")]
    from netinit import deterministic_init, make_cfg, synthetic_frames
    from hotrack_amd import fused, pointnet2_utils
    from models import pointnet_utils
    from models.hand_network import HandTrackNet
    pointnet_utils.set_operator_backend(pointnet2_utils)
    pointnet_utils.set_fused_backend(fused)
    flags = {"track_flag": False, "test_flag": True, "save_flag": False, "IKNet_flag": False}
    model = HandTrackNet(make_cfg("cuda"))
    deterministic_init(model)
    model = model.cuda().eval()
    d = synthetic_frames(5, 1, 1024)
    d = {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in d.items()}
    with torch.no_grad():
        for _ in range(5):
            ref = model(d, dict(flags))["pred_kp"].clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            model(d, dict(flags))
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 30
        g = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(g):
            out = model(d, dict(flags))
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 300
    from hotrack_amd.graph_utils import kernel_nodes
    return {"workload": "HandTrackNet forward, B=1, N=1024, HIP-graph replay", "graph_ms": round(graph * 1e3, 4), "eager_ms": round(eager * 1e3, 4),
            "launches": kernel_nodes(g), "frames_per_s": round(1.0 / graph, 1),
            "max_abs_diff_vs_eager": float((out["pred_kp"] - ref).abs().max())}


def train(steps=20, warmup=5, batch=32):
    """configs[2] per GPU through scripts/bench_train.py (single process here; bench.py launches that script directly per rank
    when WORLD_SIZE > 1)."""
    os.environ["HOTRACK_KEEP_GRAPH"] = "1"
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from bench_train import run_training_leg
    import torch
    torch.cuda.set_device(0)
    return run_training_leg(int(steps), int(warmup), int(batch), True, 0, 1)


if __name__ == "__main__":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    fn = {"stress": stress, "latency": latency, "train": train}[sys.argv[1]]
    print(json.dumps(fn(*sys.argv[2:])), flush=True)
