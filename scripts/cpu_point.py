"""One point of bench.py's `cpu_baseline`: the reference's CPU path (port: oracle/cpu_reference.py driving the same network,
attention NOT elided, B = 1, eval forward -- SURVEY.md 8(d)) at a given number of torch intra-op threads, per-frame median /
p10 / p90 over a bounded sample.

In-process (bench.py, 1 and 16 threads: what the reference's test.py:26 sets, and torch's practical scaling limit on per-frame
tensors) through `measure()`; as a CHILD process for the pinned physical-core points, because thread placement has to be fixed
before the OpenMP runtime starts:

    python scripts/cpu_point.py --threads 64 --budget 6        # prints one JSON object

The child binds itself to the first `threads` PHYSICAL cores (one hardware thread per core, taken from
/sys/devices/system/cpu/*/topology) and runs with OMP_PROC_BIND=close OMP_PLACES=cores.  Test infrastructure: this is the only
bench leg that executes oracle/ code, never inside a timed GPU region."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = {"track_flag": False, "test_flag": True, "save_flag": False, "IKNet_flag": False}


def physical_cores(allowed=None):
    """One logical CPU per physical core (the lowest sibling), in core order, restricted to the process's affinity mask."""
    allowed = set(allowed if allowed is not None else (os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else range(os.cpu_count() or 1)))
    seen, firsts = set(), []
    for cpu in sorted(allowed):
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{cpu}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(cpu)
        if sib not in seen:
            seen.add(sib)
            firsts.append(cpu)
    return firsts


def measure(threads, budget_s, npoints=1024, max_frames=200, model=None):
    """{frames, median_ms, p10_ms, p90_ms, frames_per_s, ...} of B=1 frames at `threads` intra-op threads, ~budget_s of CPU work."""
    import torch
    sys.path[:0] = [ROOT, os.path.join(ROOT, "network")]
    from netinit import deterministic_init, make_cfg, synthetic_frames
    from models import pointnet_utils
    from models.hand_network import HandTrackNet
    from oracle import cpu_reference
    saved = pointnet_utils._OPS
    pointnet_utils.set_operator_backend(cpu_reference)
    try:
        if model is None:
            torch.manual_seed(0)
            model = HandTrackNet(make_cfg("cpu"), elide_dead_attention=False)
            deterministic_init(model)
            model = model.eval()
        frames = [synthetic_frames(1000 + i, 1, npoints) for i in range(4)]
        torch.set_num_threads(threads)
        with torch.no_grad():
            t0 = time.perf_counter()
            model(frames[0], dict(FLAGS))  # first warm-up frame, also a guard against a collapsing thread pool
            first = time.perf_counter() - t0
            if first > budget_s / 2:
                return {"threads": threads, "frames": 1, "median_ms": round(first * 1e3, 2), "p10_ms": None, "p90_ms": None,
                        "frames_per_s": round(1.0 / first, 3), "note": "thread pool collapses at this size: one frame only"}
            t_w, nw = time.perf_counter(), 1
            while nw < 20 and time.perf_counter() - t_w < budget_s * 0.15:  # up to 20 warm-up frames
                model(frames[nw % 4], dict(FLAGS))
                nw += 1
            ts, t_all = [], time.perf_counter()
            while len(ts) < max_frames and time.perf_counter() - t_all < budget_s * 0.8:
                t0 = time.perf_counter()
                model(frames[len(ts) % 4], dict(FLAGS))
                ts.append(time.perf_counter() - t0)
    finally:
        pointnet_utils.set_operator_backend(saved)
    ts.sort()
    q = lambda f: ts[min(len(ts) - 1, int(f * len(ts)))]
    return {"threads": threads, "frames": len(ts), "warmup_frames": nw, "median_ms": round(q(0.5) * 1e3, 2), "p10_ms": round(q(0.1) * 1e3, 2),
            "p90_ms": round(q(0.9) * 1e3, 2), "frames_per_s": round(1.0 / q(0.5), 3)}


def pinned_point(threads, budget_s, npoints=1024, timeout_s=None):
    """Runs this file as a child bound to `threads` physical cores; returns its JSON object (or {"error": ...})."""
    import subprocess
    cores = physical_cores()
    if threads > len(cores):
        return {"threads": threads, "error": f"only {len(cores)} physical cores in the affinity mask"}
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores", MKL_NUM_THREADS=str(threads),
               HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.abspath(__file__), "--threads", str(threads), "--budget", str(budget_s), "--npoints", str(npoints)]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s or (budget_s * 4 + 90), env=env, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"threads": threads, "error": "timeout"}
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if out.returncode or not lines:
        return {"threads": threads, "error": (out.stderr or out.stdout)[-300:]}
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, required=True)
    ap.add_argument("--budget", type=float, default=6.0)
    ap.add_argument("--npoints", type=int, default=1024)
    a = ap.parse_args()
    cores = physical_cores()[:a.threads]
    if hasattr(os, "sched_setaffinity"):
        os.sched_setaffinity(0, cores)  # before torch (and its OpenMP pool) is loaded
    res = measure(a.threads, a.budget, a.npoints)
    res.update(pinned=True, placement="first %d physical cores (one hardware thread each), OMP_PROC_BIND=close OMP_PLACES=cores" % len(cores))
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
