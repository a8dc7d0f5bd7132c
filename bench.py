#!/usr/bin/env python
"""bench.py -- HandTrackNet point-cloud frames/s (N=1024) on 1..8 MI355X  (BASELINE.json metric).

One "step" = one HandTrackNet.forward (eval, no_grad) over one batch of synthetic clouds that is
already resident in HBM.  Workload at every N: BASELINE.json configs[1] per GPU -- batch 64 x 1024
points (+21 keypoints) -- i.e. weak scaling: each rank owns its own batches (per-frame batches shard
across GPUs with no data-path collective).

Serving-loop shape of the timed region: POOL distinct batches stay resident in HBM; step s copies batch
s % POOL into the static input buffers of the captured HIP graph of stream s % inflight (device-to-device,
on that stream) and replays it -- so consecutive steps see different data.  The K-step region demanded by
the driver contract (barrier + synchronize on both sides, max over ranks) is REPEATED until at least
--min-time seconds have been timed; `ms_per_step` / `value` are the median region, every region is listed.
`value_tied_inputs`: the same loop on batches whose first cloud contains duplicated points, so its FPS
arg-maxima tie and the real second-level FPS pass (not the prefix shortcut) is priced.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29500 bench.py --gpus 8 --steps 50 --warmup 10

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant hand-written kernel, timed live with HIP events on the launching stream
  cpu_baseline  the reference's CPU path (port, oracle/cpu_reference.py) on this box's host cores
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "network"))

# The serving loop keeps several batches in flight, one HIP stream + captured graph each.  The HIP runtime multiplexes streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4, shared with the process's other streams): with the default, two to
# four of our streams end up behind each other on one queue and the chip sees at most two of them at a time
# (78 k frames/s whether 2 or 4 are in flight); with 8 queues every stream has its own (4 in flight: 87 k).  Must be set
# before the runtime initialises; a deployment sets it in the service's environment.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ALG_BYTES_PER_FRAME_1024 = 12_046_456  # SURVEY.md 8(d) / BASELINE.md: operator-API compulsory bytes per frame
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3           # MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_16x16x4_f32) peak
FLAGS = {"track_flag": False, "test_flag": True, "save_flag": False, "IKNet_flag": False}


def _to(d, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else _to(v, dev)) for k, v in d.items()}


def build_model(dev, elide=True):
    from netinit import deterministic_init, make_cfg
    from models.hand_network import HandTrackNet
    torch.manual_seed(0)
    model = HandTrackNet(make_cfg(dev), elide_dead_attention=elide)
    deterministic_init(model)  # stands in for xavier-random weights: there is no checkpoint to download
    return model.to(dev).eval()


class KernelTimer:
    """HIP-event timing of the hand-written kernels: hotrack_amd.pointnet2_hip.PROFILE makes the bindings
    record two events immediately around each C-ABI enqueue, on the stream the kernel is launched on."""

    def __init__(self):
        from hotrack_amd import pointnet2_hip
        self.native = pointnet2_hip
        self.records = []

    def start(self):
        self.native.PROFILE = self.records

    def stop(self):
        self.native.PROFILE = None

    def summary(self, passes):
        by = {}
        for name, a, s, e in self.records:
            if name in ("fps_kernel", "fps_prefix_kernel", "fps_knn_kernel"):
                key = (name, a[0], a[1], a[2])  # (b, n, m)
            elif name == "sa_mlp_max_pair_kernel":  # both scales of a query module in one launch: (b, s, (k0, k1), c1, c2, c3)
                q0, q1 = a[4]._obj, a[5]._obj
                key = (name, a[0], q0.s, (q0.k, q1.k), a[1], a[2], a[3])
            elif name == "mlp2_rows_kernel":  # two fused per-point layers: (rows, c1, c2, c3)
                key = (name, a[0], a[1], a[2], a[3])
            elif name == "sa_mlp_max_kernel":
                key = ("sa_mlp_max_kernel", a[0], a[2], a[3], a[4], a[5], a[6])  # (b, s, k, c1, c2, c3)
            else:  # hooked launches without a roofline entry of their own (three_nn_interp_kernel, ball_tie_kernel, linear_small_kernel)
                continue
            by.setdefault(key, []).append(s.elapsed_time(e) * 1e-3)
        # key -> (mean seconds per launch, launches per step)
        return {k: (sum(v) / len(v), len(v) / max(passes, 1)) for k, v in by.items()}


def roofline_of(key, sec, per_step):
    """Roofline entry of one kernel: algorithmic work per launch / measured launch duration."""
    if key[0].startswith("sa_mlp_max"):
        kname, B, S, K, C1, C2, C3 = key
        ksum = sum(K) if isinstance(K, tuple) else K
        flops = 2.0 * B * S * ksum * (C1 * C2 + C2 * C3)  # the two MFMA layers the kernel executes per (s,k) position
        return {"bound": "mfma", "kernel": "%s<%d,%d,%d> (B=%d,S=%d,K=%s)" % (kname, C1, C2, C3, B, S, "+".join(map(str, K)) if isinstance(K, tuple) else K),
                "achieved": round(flops / sec / 1e12, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(flops / sec / 1e12 / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                "flops_per_launch": flops, "us_per_launch": round(sec * 1e6, 2), "launches_per_step": per_step}
    if key[0] == "mlp2_rows_kernel":
        _, R, C1, C2, C3 = key
        flops = 2.0 * R * (C1 * C2 + C2 * C3)
        return {"bound": "mfma", "kernel": "mlp2_rows_kernel<%d,%d,%d> (rows=%d)" % (C1, C2, C3, R),
                "achieved": round(flops / sec / 1e12, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(flops / sec / 1e12 / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                "flops_per_launch": flops, "us_per_launch": round(sec * 1e6, 2), "launches_per_step": per_step}
    name, B, N, M = key
    kb = B * (12 * N + 4 * M)  # FPS bytes = B(12N + 4M)
    if name == "fps_prefix_kernel":
        return {"bound": "hbm", "kernel": "fps second level (B=%d,N=%d,M=%d)" % (B, N, M), "achieved": round(kb / sec / 1e9, 3),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(kb / sec / 1e9 / HBM_PEAK_GBS, 6), "traffic": None,
                "bytes_per_launch": kb, "us_per_launch": round(sec * 1e6, 2), "launches_per_step": per_step,
                "note": "FPS over level 1's samples = level 1's first M picks unless an arg-max tied (pn2_ext.h): per cloud "
                        "the launch returns at once (no tie: all clouds of this synthetic batch) or runs the real pass"}
    label = "fps_kernel" if name == "fps_kernel" else "fps_knn_kernel: sampling + the keypoints' k-NN lists in one launch"
    return {"bound": "hbm", "kernel": "%s (B=%d,N=%d,M=%d)" % (label, B, N, M), "achieved": round(kb / sec / 1e9, 3),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(kb / sec / 1e9 / HBM_PEAK_GBS, 6), "traffic": None,
            "bytes_per_launch": kb, "us_per_launch": round(sec * 1e6, 2), "launches_per_step": per_step,
            "us_per_fps_iteration": round(sec * 1e6 / max(M - 1, 1), 4),
            "note": "bound by its M-step dependency chain, not by HBM (SURVEY.md section 7, hard part 2)"}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(npoints, budget_s=24.0):
    """Reference CPU path (port) on the host, SURVEY.md 8(d) protocol: B=1 frames through the same network with NO dead-work
    elision, warm-up then up to 200 timed frames per point, per-frame median / p10 / p90 (scripts/cpu_point.py).  Points:
    1 thread (what the reference's test.py:26 sets) and 16 threads in this process; 64 and 128 threads as child processes
    PINNED to that many physical cores (one hardware thread per core, OMP_PROC_BIND=close) -- "all physical cores" of 8(d),
    capped by what the host has.  Bounded to ~budget_s of CPU work; the frames actually timed are stated in `sample`."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import cpu_point
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    phys = len(cpu_point.physical_cores())
    inproc = sorted({1, min(16, avail)})
    pinned = sorted({min(64, phys), min(128, phys)} - set(inproc))
    per = budget_s / (len(inproc) + len(pinned))
    out = {}
    model = build_model("cpu", elide=False)
    for threads in inproc:
        out[threads] = cpu_point.measure(threads, per, npoints, model=model)
    torch.set_num_threads(min(avail, 32))
    for threads in pinned:
        out[threads] = cpu_point.pinned_point(threads, per, npoints)
    ok = {t: v for t, v in out.items() if "frames_per_s" in v}
    best = max(ok, key=lambda t: ok[t]["frames_per_s"])
    return {
        "value": ok[best]["frames_per_s"], "unit": "frames/s", "cores": best, "kind": "port",
        "sample": f"{ok[best]['frames']} frames (median of per-frame times), B=1, N={npoints}, eval forward, reference fallback "
                  f"algorithms (oracle/cpu_reference.py), attention not elided, torch CPU intra-op threads={best} "
                  f"(best of {sorted(out)}; host: {phys} physical cores / {avail} hardware threads)",
        "cpu_model": _cpu_model(), "host_cores": avail, "host_physical_cores": phys,
        "one_thread": out.get(1), "physical_cores": {str(t): out[t] for t in pinned},
        "by_threads": {str(t): v for t, v in sorted(out.items())},
        "note": "torch.set_num_threads(all %d hardware threads) is not a point of this table: at B=1 the intra-op pool collapses "
                "there (one frame took 37.5 s on the 256-thread driver box, BENCH_r03.json); the pinned physical-core points "
                "are the 'all physical cores' setting of SURVEY.md 8(d)" % avail,
    }


POOL = 5  # distinct resident batches rotated through the graphs' static input buffers (coprime with the streams in flight: every stream sees every batch)


def _leaves(d):
    for k in sorted(d):
        if isinstance(d[k], dict):
            yield from _leaves(d[k])
        elif torch.is_tensor(d[k]):
            yield d[k]


def _flatten(d, dev):
    """The tensors of one batch as views into ONE device buffer, so handing a batch over is a single device-to-device copy."""
    leaves = [(k, kk) for k in sorted(d) for kk in ([None] if torch.is_tensor(d[k]) else sorted(d[k]))]
    get = lambda k, kk: d[k] if kk is None else d[k][kk]
    flat = torch.empty(sum(get(k, kk).numel() for k, kk in leaves), dtype=torch.float32, device=dev)
    out, off = {}, 0
    for k, kk in leaves:
        t = get(k, kk)
        v = flat[off:off + t.numel()].view(t.shape)
        v.copy_(t)
        off += t.numel()
        if kk is None:
            out[k] = v
        else:
            out.setdefault(k, {})[kk] = v
    return out, flat


def make_pool(rank, batch, npoints, dev, tied=False):
    """POOL seeded batches (SURVEY.md 8(d) synthetic clouds), resident in HBM, each as (dict of views, flat buffer).
    tied=True: in cloud 0 of every batch 64 points are exact duplicates of their predecessors (what depth-quantised
    sensors produce), so FPS arg-maxima tie."""
    from netinit import synthetic_frames
    pool = []
    for r in range(POOL):
        d = synthetic_frames(1000 + POOL * rank + r, batch, npoints)
        if tied:
            d["hand_points"][0, 1:npoints:npoints // 64] = d["hand_points"][0, 0:npoints - 1:npoints // 64]
        pool.append(_flatten(d, dev))
    return pool


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="clouds per GPU per step (BASELINE configs[1])")
    ap.add_argument("--npoints", type=int, default=1024)
    ap.add_argument("--min-time", type=float, default=6.0, help="repeat the K-step timed region until this many seconds are timed "
                    "(default 6 s: longer than the 5-s tick of an external GPU-busy sampler)")
    ap.add_argument("--train-steps", type=int, default=20, help="steps of the training leg (configs[2] per GPU, 32 x 1024, whole step "
                    "as HIP graphs; WORLD_SIZE > 1: data parallel with a flat-gradient RCCL all-reduce) appended after the headline "
                    "regions; 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the secondary legs (child processes after the headline regions: "
                    "`train` = configs[2] per GPU, and at N = 1 `stress` = configs[4] per GPU and `latency_b1`)")
    ap.add_argument("--no-elide", action="store_true", help="also compute the attention the reference discards")
    ap.add_argument("--no-fused", action="store_true", help="disable the fused SA kernels (unfused torch MLPs)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--sa-cus", type=int, default=0, help="with more than one batch in flight: CUs the persistent SA grids occupy "
                    "(pn2x_sa_set_compute_units; a workgroup of that kernel fills its CU, the rest stay available to the other stream); "
                    "0 = all (default: 240 gives +0.7 %% frames/s with two batches in flight, but the SA launches themselves then run 8 "
                    "rounds of tiles instead of 7, i.e. the dominant kernel's own roofline fraction drops from 0.70 to 0.63)")
    ap.add_argument("--inflight", type=int, default=4, help="number of batches in flight: step i is replayed on HIP stream "
                    "i %% inflight (each stream has its own captured graph and buffers), so one batch's FPS / small "
                    "kernels overlap another batch's GEMMs")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    ndev = torch.cuda.device_count()
    backend = os.environ.get("PN2_BENCH_BACKEND", "nccl")  # "gloo": lets N ranks share one GPU (harness self-test only)
    if backend == "nccl":
        assert local < ndev, f"LOCAL_RANK {local} but only {ndev} GPUs are visible"
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI; only used for the barrier / max-time reduce
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from hotrack_amd import pointnet2_utils as hip_ops
    from models import pointnet_utils
    pointnet_utils.set_operator_backend(hip_ops)
    fused_on = False
    if not args.no_fused:
        from hotrack_amd import fused
        pointnet_utils.set_fused_backend(fused)
        fused_on = True

    model = build_model(dev, elide=not args.no_elide)
    pool = make_pool(rank, args.batch, args.npoints, dev)          # seeded per rank, resident in HBM
    tied_pool = make_pool(rank, args.batch, args.npoints, dev, tied=True)
    timer = KernelTimer()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    use_graph = not args.no_graph
    ninf = max(1, args.inflight) if use_graph else 1
    outs = [None] * ninf
    with torch.no_grad():
        for i in range(max(args.warmup, 3) if use_graph else args.warmup):
            outs[0] = model(pool[i % POOL][0], dict(FLAGS))
        eager_ms = None
        if use_graph:
            # one forward = ~50 short kernels: capture it once per stream, replay it per step (no host launch cost)
            sync_all()
            t0 = time.perf_counter()
            for i in range(5):
                model(pool[i % POOL][0], dict(FLAGS))
            torch.cuda.synchronize()
            eager_ms = (time.perf_counter() - t0) / 5 * 1e3
            from hotrack_amd import ext
            import hotrack_amd
            hotrack_amd.streams_in_flight(ninf)  # warns once when the hardware-queue setting would serialise the streams
            gstreams = [torch.cuda.Stream() for _ in range(ninf)]
            slots = [_flatten(pool[0][0], dev) for _ in range(ninf)]  # static inputs of each stream's graph
            graphs = []
            sa_cus = args.sa_cus if (ninf > 1 and fused_on and "PN2_SA_CUS" not in os.environ) else 0
            ext.sa_set_compute_units(sa_cus)  # grid sizes are baked into the graphs at capture
            for i in range(ninf):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    outs[i] = model(slots[i][0], dict(FLAGS))
                graphs.append(g)

            def replay_slot(i, src_flat):
                """Batch `src_flat` through slot i's graph, asynchronously."""
                with torch.cuda.stream(gstreams[i]):
                    slots[i][1].copy_(src_flat, non_blocking=True)
                    graphs[i].replay()
            g_single = None
            if ninf > 1:  # the single-stream reference point replays a graph captured with every CU available to the SA grids
                ext.sa_set_compute_units(0)
                g_single = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_single):
                    out_single = model(slots[0][0], dict(FLAGS))
                ext.sa_set_compute_units(sa_cus)

        def make_step(n, src):
            counter = [0]
            if not use_graph:
                def step():
                    outs[0] = model(src[counter[0] % POOL][0], dict(FLAGS))
                    counter[0] += 1
                return step

            def step():
                s = counter[0]
                counter[0] += 1
                i = s % n
                if n == 1 and g_single is not None:
                    with torch.cuda.stream(gstreams[0]):
                        slots[0][1].copy_(src[s % POOL][1], non_blocking=True)
                        g_single.replay()
                    return
                # the serving loop's hand-over: this step's batch goes into the graph's static inputs (one device-to-device
                # copy of the batch's flat buffer), on the graph's stream
                replay_slot(i, src[s % POOL][1])
            return step

        local_regions = []  # this rank's own wall time of every headline region (per-rank frames/s spread at N > 1)

        def timed_region(step):
            sync_all()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            local_regions.append(time.perf_counter() - t0)
            sync_all()
            return max_over_ranks(time.perf_counter() - t0)

        def repeat_regions(step, min_time):
            regions = [timed_region(step)]
            reps = min(1000, max(1, int(min_time / max(regions[0], 1e-6) + 0.999)))  # same on every rank: from the reduced time
            for _ in range(reps - 1):
                regions.append(timed_region(step))
            return regions

        step = make_step(ninf, pool)
        for _ in range(args.warmup):
            step()
        single_ms = None
        if use_graph and ninf > 1:  # informational: the same K steps strictly one after another on one stream
            one = make_step(1, pool)
            sync_all()  # a graph must never be replayed while an earlier replay of it is still running
            for _ in range(3):
                one()
            single_ms = timed_region(one) / args.steps * 1e3
        local_regions.clear()
        regions = repeat_regions(step, args.min_time)   # ---- THE timed regions: K steps each, barrier + sync on both sides
        headline_local = sorted(local_regions)[len(local_regions) // 2]
        tied_step = make_step(ninf, tied_pool)
        sync_all()
        for _ in range(3):
            tied_step()
        tied_regions = repeat_regions(tied_step, min(args.min_time, 0.3))
        sync_all()
        # separate short eager pass for the per-kernel HIP-event timing (events perturb the async pipeline)
        timed_passes = min(args.steps, 20)
        timer.start()
        for i in range(timed_passes):
            model(pool[i % POOL][0], dict(FLAGS))
        torch.cuda.synchronize()
        timer.stop()
        # ---- what the replays computed: every resident batch through every stream's graph against the eager forward of that
        # batch, at the headline size (a wrong-but-finite replay must not print a headline)
        replay_check = None
        if use_graph:
            eager_kp = [model(pool[r][0], dict(FLAGS))["pred_kp"].clone() for r in range(POOL)]
            worst = 0.0
            for i in range(ninf):
                for r in range(POOL):
                    replay_slot(i, pool[r][1])
                    gstreams[i].synchronize()
                    worst = max(worst, float((outs[i]["pred_kp"] - eager_kp[r]).abs().max()))
            assert worst <= 1e-5, f"graph replay differs from the eager forward of the same batch by {worst}"
            replay_check = {"max_abs_diff_pred_kp": worst, "batches": POOL, "streams": ninf, "tolerance": 1e-5}
    assert all(torch.isfinite(o["pred_kp"]).all() for o in outs if o is not None)

    # ---- secondary legs, all OUTSIDE the headline regions and each in a CHILD process (scripts/bench_legs.py): an exception, a
    # hang (bounded by a timeout, the child is killed by pid) or a crash in a leg costs that leg's entry ({"error": ...}), never
    # the headline.  N = 1: train (configs[2] per GPU), stress (configs[4] per GPU), latency_b1.  N > 1: per-rank spread of the
    # headline and the data-parallel training leg -- the children of all ranks form their OWN process group on a fresh port, so
    # a multi-GPU record shows RCCL carrying the gradient all-reduce (network/trainer.py dp=flat) while the parents' group
    # stays untouched; the parents meet again at the barrier below whatever happened to the children.
    per_rank = None
    legs = {}
    torch.cuda.synchronize()
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import bench_legs
    backend_world = dist.get_world_size() if world > 1 else 1
    assert backend_world == args.gpus, f"backend sees {backend_world} ranks, --gpus {args.gpus}"
    if world > 1:
        t = torch.tensor([args.batch * args.steps / headline_local], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        per_rank = [round(float(g), 1) for g in gathered]
    if use_graph and not args.no_legs:
        del graphs, slots, g_single
        torch.cuda.empty_cache()
    if not args.no_legs and args.train_steps > 0:
        if world > 1:
            import socket
            port = torch.zeros(1, dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
            if rank == 0:
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    port[0] = sk.getsockname()[1]
            dist.broadcast(port, src=0)
            # the children rendezvous among themselves: without the launcher's agent-store variables (TORCHELASTIC_USE_AGENT_STORE
            # makes a worker CONNECT to a store the elastic agent serves at MASTER_PORT -- nobody serves the fresh port, and the
            # children would wait for one until the leg's timeout), rank 0's child hosts the TCP store itself
            env = {k: v for k, v in os.environ.items() if not k.startswith(("TORCHELASTIC_", "TORCH_NCCL_ASYNC", "GROUP_", "ROLE_"))}
            env.update(MASTER_PORT=str(int(port[0])), MASTER_ADDR="127.0.0.1", PN2_DIST_BACKEND=backend, HOTRACK_KEEP_GRAPH="1")
            cmd = [sys.executable, os.path.join(ROOT, "scripts", "bench_train.py"), "--steps", str(args.train_steps), "--warmup", "5",
                   "--batch", "32", "--graph"]
            got = bench_legs.run_child(cmd, float(os.environ.get("PN2_BENCH_LEG_TIMEOUT", "300")), env=env, expect_json=(rank == 0))
            # every parent is alive here whatever its child did; one small collective tells rank 0 whether any child failed
            bad = torch.tensor([1 if "error" in got else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(bad, op=dist.ReduceOp.SUM)
            if int(bad[0]) and "error" not in got:
                got = {"error": "%d of %d rank children failed" % (int(bad[0]), world)}
            elif int(bad[0]):
                got["failed_ranks"] = int(bad[0])
            legs["train"] = got
        else:
            legs["train"] = bench_legs.run_child(bench_legs.leg_cmd("train", args.train_steps, 5, 32), 420)
    if not args.no_legs and world == 1:
        legs["stress"] = bench_legs.run_child(bench_legs.leg_cmd("stress"), 300)
        legs["latency_b1"] = bench_legs.run_child(bench_legs.leg_cmd("latency"), 300)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()  # before the CPU leg: the other ranks are done (rank 0 alone times the host)

    if rank == 0:
        med = sorted(regions)[len(regions) // 2]
        dt = med
        frames_per_region = args.batch * world * args.steps
        fps = frames_per_region / dt
        tied_fps = frames_per_region / sorted(tied_regions)[len(tied_regions) // 2]
        alg_bytes = ALG_BYTES_PER_FRAME_1024 if args.npoints == 1024 else None
        # dominant hand-written kernel = largest (mean launch time x launches per step) among the hooked kernels
        ks = timer.summary(timed_passes)
        per_kernel = sorted(({"key": k, "sec": v[0], "per_step": v[1]} for k, v in ks.items()),
                            key=lambda r: -r["sec"] * r["per_step"])
        # group launches of the same kernel instantiation (q1/q2 share the 128-128-192 instance at two K)
        inst = {}
        for r in per_kernel:
            name = "sa_mlp_max" + str(r["key"][4:]) if r["key"][0].startswith("sa_mlp_max") else r["key"][0]
            inst.setdefault(name, []).append(r)
        dom = max(inst.values(), key=lambda rs: sum(r["sec"] * r["per_step"] for r in rs)) if inst else []
        roof = None
        if dom:
            top = max(dom, key=lambda r: r["sec"] * r["per_step"])
            roof = roofline_of(top["key"], top["sec"], top["per_step"])
            roof["instance_us_per_step"] = round(sum(r["sec"] * r["per_step"] for r in dom) * 1e6, 1)
        other = [roofline_of(r["key"], r["sec"], r["per_step"]) for r in per_kernel]
        # HBM traffic per launch cannot be sampled from inside the process: it comes from the committed
        # rocprofv3 PMC passes of this same command (scripts/pmc_summary.py -> profiles/rNN_pmc_traffic.json, newest round)
        try:
            import glob
            src = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
            pmc = json.load(open(src))
            if roof is not None and roof["bound"] == "mfma":
                tag = "<%d, %d, %d" % top["key"][4:7]
                cands = [e for e in pmc["kernels"] if top["key"][0] in e["kernel"] and tag in e["kernel"]]
                if cands:  # the K=64 launch is the larger-grid one of the instance
                    roof["traffic"] = max(cands, key=lambda e: e["grid_threads"])["hbm_bytes"]
                    roof["traffic_source"] = ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, reads x2 per "
                                              "MI355X_MICROARCH.md)" % os.path.basename(src))
        except (OSError, KeyError, ValueError, IndexError):
            pass
        res = {
            "metric": "HandTrackNet point-cloud frames/sec (N=%d)" % args.npoints, "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "timed_region_s": round(sum(regions), 4), "timed_regions": len(regions),
            "region_ms_per_step": {"min": round(min(regions) / args.steps * 1e3, 4), "median": round(med / args.steps * 1e3, 4),
                                   "max": round(max(regions) / args.steps * 1e3, 4)},
            "value_tied_inputs": round(tied_fps, 2),
            "config": {"workload": "HandTrackNet forward, batch=%d synthetic %d-pt clouds per GPU (BASELINE configs[1])"
                                   % (args.batch, args.npoints),
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "npoints": args.npoints,
                       "parallelism": "dp%d (independent batches, no collective)" % world, "world_size": world,
                       "resident_batches_rotated": POOL,
                       "tied_inputs": "cloud 0 of every batch has 64 duplicated points: FPS arg-maxima tie, the second-level FPS "
                                      "runs its real pass for that cloud (value_tied_inputs, %d regions)" % len(tied_regions),
                       "dead_attention_elided": not args.no_elide, "fused_sa_kernels": fused_on,
                       "launch": "hipGraph replay" if use_graph else "eager", "batches_in_flight": ninf,
                       "hip_hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "sa_compute_units": (sa_cus or 256) if use_graph else 256,
                       "single_stream_ms_per_step": None if single_ms is None else round(single_ms, 4),
                       "eager_ms_per_step": None if eager_ms is None else round(eager_ms, 4),
                       "weights": "deterministic random init (no checkpoint available offline)"},
            "frame_alg_bytes": alg_bytes,
            "frame_hbm_frac": None if alg_bytes is None else round(alg_bytes * fps / world / 1e9 / HBM_PEAK_GBS, 6),
            "roofline": roof,
            "kernels": other,
            "replay_check": replay_check,
        }
        from hotrack_amd import gemm_tuning
        res["gemm_table"] = gemm_tuning.status()["gemm_table"]
        res["config"]["gemm_table_detail"] = gemm_tuning.status()["detail"]
        res["world_size_seen_by_backend"] = backend_world
        if per_rank is not None:
            res["per_rank_frames_per_s"] = {"min": min(per_rank), "max": max(per_rank), "ranks": per_rank}
        res.update(legs)  # train / stress / latency_b1: results of the child legs, or {"error": ...}
        if not args.no_cpu_baseline:
            try:  # (N > 1: rank 0 alone, after the process group is gone; per-GPU comparison = value / n_gpus)
                res["cpu_baseline"] = cpu_baseline(args.npoints)
                res["speedup_vs_cpu_baseline"] = round(fps / world / res["cpu_baseline"]["value"], 1)
            except Exception as exc:  # noqa: BLE001  (a reported baseline must never cost the line)
                res["cpu_baseline"] = {"error": repr(exc)[:300]}
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
