"""SDF-lookup row at the reference's sizes on one MI355X, next to the CPU oracle.

  object side: 2048 particles x 1024 points, 201^3 fp16 volume, trilinear (gf_optimize_obj.evaluate / optimize)
  hand side:   5120 particles x 778 vertices, 151^3 fp16 volume, nearest voxel (gf_optimize_hand_pose.query_sdf)

Prints one JSON object.  `pairs/s` = (particle, point) evaluations per second; `l2_gather_GBps` = voxel bytes
requested per second (8 x 2 B per trilinear pair, 2 B per nearest pair) -- the volume is cache-resident, so this is
a cache-gather rate, not HBM traffic; `unfused_bytes` = what the reference's chain of elementwise torch kernels
moves through HBM for the same call (every temporary written once and read once), the figure the fusion removes.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _sdf_cases import hand_particles, hand_pose_particles, make_volume, object_points, particles, random_pose  # noqa: E402
from hotrack_amd import sdf  # noqa: E402


def gpu_time(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    out = {"device": torch.cuda.get_device_name(0)}

    P, N, res, stride = 2048, 1024, 201, 0.002
    vol = make_volume(res, stride, "box", np.float16)
    pc = object_points(1, N, "box")
    R0, t0 = random_pose(1)
    cam = (pc @ R0.T + t0).astype(np.float32)
    rot, tr = particles(2, P, R0, t0)
    pre = np.random.default_rng(3).standard_normal((P, 6)).astype(np.float32)
    pre[0] = 0
    dvol, dcam, drot, dtr, dpre = d(vol), d(cam), d(rot), d(tr), d(pre)
    dR0, dt0 = d(R0), d(t0)
    t_eval_lin = gpu_time(lambda: sdf.particle_energy(dcam, drot, dtr, dvol, stride))
    t_opt_lin = gpu_time(lambda: sdf.obj_optimize(dcam, dR0, dt0, dpre, dvol, stride), iters=20)
    t_build = gpu_time(lambda: sdf.CornerVolume(dvol), iters=10, warm=2)
    lin, dvol = dvol, sdf.CornerVolume(dvol)  # everything below uses the corner layout
    t_eval = gpu_time(lambda: sdf.particle_energy(dcam, drot, dtr, dvol, stride))
    t_opt = gpu_time(lambda: sdf.obj_optimize(dcam, dR0, dt0, dpre, dvol, stride), iters=20)
    # Distance() alone on what the reference passes it: the cloud in every particle's object frame, (P*N, 3)
    flat = d(np.einsum("pnj,pjk->pnk", cam[None].astype(np.float64) - tr[:, None].astype(np.float64), rot.astype(np.float64))
             .reshape(-1, 3).astype(np.float32))
    t_dist = gpu_time(lambda: sdf.distance(flat, dvol, stride))
    rnd = d((np.random.default_rng(4).uniform(-0.2, 0.2, (P * N, 3))).astype(np.float32))
    t_rand = gpu_time(lambda: sdf.distance(rnd, dvol, stride))
    pairs = P * N
    # reference evaluate(): sub, bmm, then Distance = ~55 elementwise kernels over P*N fp32/int64 temporaries
    unfused = pairs * (3 * 4 * 4 + 3 * (4 * 8) + 3 * 8 * 3 + 8 * (8 * 3 + 8 + 2) + 15 * 12)
    out["object"] = {
        "config": f"{P} particles x {N} points, {res}^3 fp16 volume, trilinear",
        "layout": "corner cells (16 B per voxel, one load per lookup); linear-layout figures kept beside",
        "corner_volume_build_us": t_build * 1e6, "evaluate_linear_layout_us": t_eval_lin * 1e6,
        "optimize_10_iterations_linear_layout_us": t_opt_lin * 1e6,
        "evaluate_us": t_eval * 1e6, "evaluate_Gpairs_per_s": pairs / t_eval / 1e9,
        "evaluate_l2_gather_GBps": pairs * 16 / t_eval / 1e9,
        "distance_only_us": t_dist * 1e6, "distance_Gpoints_per_s": pairs / t_dist / 1e9,
        "distance_hbm_GBps": pairs * 16 / t_dist / 1e9,
        "distance_uniform_random_queries_us": t_rand * 1e6,
        "optimize_10_iterations_us": t_opt * 1e6, "optimize_calls_per_s": 1 / t_opt,
        "unfused_bytes_per_evaluate_estimate": unfused, "unfused_hbm_floor_us": unfused / 8e12 * 1e6,
    }
    if not a.no_cpu:
        from oracle import sdf_oracle as S
        sub = 64
        t0_ = time.perf_counter()
        S.particle_energy(cam, rot[:sub], tr[:sub], vol, stride)
        dt = time.perf_counter() - t0_
        out["object"]["cpu_oracle_1core_Gpairs_per_s"] = sub * N / dt / 1e9
        out["object"]["gpu_over_cpu_1core"] = (pairs / t_eval) / (sub * N / dt)

    B, Nv, res, scale = 5120, 778, 151, 0.003
    vol = make_volume(res, scale, "capsule", np.float16)
    hand = hand_pose_particles(5, B, Nv, R0, t0)             # candidate hands: one blob, small rigid perturbations
    dvol, dhand = d(vol), d(hand)
    t_q = gpu_time(lambda: sdf.query_sdf(dhand, dR0, dt0, dvol, scale, with_penetration=True))
    t_q1 = gpu_time(lambda: sdf.query_sdf(dhand, dR0, dt0, dvol, scale))
    dworst = d(hand_particles(5, B, Nv, R0, t0, extent=0.25))  # worst case: every vertex uniform over the whole volume
    t_qw = gpu_time(lambda: sdf.query_sdf(dworst, dR0, dt0, dvol, scale, with_penetration=True))
    pairs = B * Nv
    alg = pairs * (12 + 2) + B * 2  # hand read once, sdf written once
    out["hand"] = {
        "config": f"{B} particles x {Nv} vertices, {res}^3 fp16 volume, nearest voxel",
        "query_plus_penetration_us": t_q * 1e6, "query_only_us": t_q1 * 1e6,
        "query_uniform_random_vertices_us": t_qw * 1e6, "Gpairs_per_s": pairs / t_q / 1e9,
        "algorithmic_bytes": alg, "hbm_GBps": alg / t_q / 1e9, "hbm_frac_of_8TBps": alg / t_q / 8e12,
    }
    if not a.no_cpu:
        from oracle import sdf_oracle as S
        sub = 512
        t0_ = time.perf_counter()
        S.nearest(hand[:sub], R0, t0, vol, scale)
        dt = time.perf_counter() - t0_
        out["hand"]["cpu_oracle_1core_Gpairs_per_s"] = sub * Nv / dt / 1e9
        out["hand"]["gpu_over_cpu_1core"] = (pairs / t_q) / (sub * Nv / dt)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
