"""Generates tests/golden/hand_opt_sequence.npz from the IMPORTED reference's gf_optimize_hand_pose (SURVEY.md 2 row 9 /
VERDICT r2 row g).  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference); tests and the GPU box only read the .npz.

The reference class is created with object.__new__ (its __init__ needs the MANO / DeepSDF / contact-zone assets) and given
exactly the attributes its methods read; its hand model is this repository's SyntheticLBSHand (the reference hard-wires a
MANO layer: licensed), its silhouette PNG read (cv2.imread in set_init_para) returns a synthetic mask, its object volume is
an analytic capsule.  No reference file is touched or copied.  A four-frame sequence is tracked the way
HandTrackModel.forward drives the optimiser (track_network.py:203-213): frame t's `last_frame_kp` is frame t-1's result.
Stored: every input of every optimize() call, the pre-sampled particles, and the reference's outputs (keypoints, pose code,
rotation, translation per frame) plus the energies of frame 0's first candidate set (all energy terms at once)."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

from _sdf_cases import make_volume, random_pose, _axis_angle  # noqa: E402
from make_golden_sdf import import_reference  # noqa: E402

P, FRAMES, RES, STRIDE = 768, 4, 81, 0.005
PROJ = dict(fx=600.0, fy=600.0, cx=320.0, cy=240.0, w=640, h=480)
ENERGY_WEIGHT = {"penetrate_sum_loss": 1, "sil_loss": 0.1, "attraction_loss": 0.05, "vis_regu_loss": 10, "invis_regu_loss": 0,
                 "temporal_smooth": 1}


def _load(name):
    """This repository's module by FILE (the reference's own `models` package shadows the name once it is on sys.path)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("hotrack_" + name, os.path.join(ROOT, "network", "models", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def scenario():
    """Ground-truth hand motion around a capsule + what 'HandTrackNet + IKNet' would hand to the optimiser per frame."""
    SyntheticLBSHand = _load("hand_model").SyntheticLBSHand
    rot = _load("rotations")
    matrix_to_unit_quaternion, quaternion_to_axis_angle = rot.matrix_to_unit_quaternion, rot.quaternion_to_axis_angle
    hand = SyntheticLBSHand()
    rng = np.random.default_rng(7)
    R_obj, t_obj = random_pose(11)
    frames = []
    for f in range(FRAMES):
        # hand in the object frame: wrist below the capsule, fingers reaching into it (penetration > 0 for some candidates)
        R_ho = _axis_angle(np.array([0.3, 0.2, 1.0]), 0.15 + 0.05 * f)
        t_ho = np.array([0.005 * f, -0.135 + 0.004 * f, 0.045])
        R_gt = (R_obj.astype(np.float64) @ R_ho).astype(np.float32)
        t_gt = (R_obj.astype(np.float64) @ t_ho + t_obj).astype(np.float32)
        theta_gt = (0.25 * np.sin(np.arange(45) * 0.7 + 0.4 * f)).astype(np.float32)
        aa = quaternion_to_axis_angle(matrix_to_unit_quaternion(torch.from_numpy(R_gt)[None]))
        with torch.no_grad():
            _, kp_gt = hand.forward(th_pose_coeffs=torch.cat([aa, torch.from_numpy(theta_gt)[None]], 1), th_trans=torch.from_numpy(t_gt)[None])
        dR, dt = random_pose(100 + f, angle=0.05, trans=0.006)
        frames.append(dict(
            init_mano=(theta_gt + rng.normal(0, 0.05, 45)).astype(np.float32)[None],
            init_rot=(R_gt @ dR)[None].astype(np.float32), init_trans=(t_gt + dt).astype(np.float32).reshape(1, 3, 1),
            init_kp=(kp_gt.numpy() + rng.normal(0, 0.004, (1, 21, 3))).astype(np.float32),
            vis_mask=np.array([[k not in (8, 12, 7) for k in range(21)]]), gt_kp=kp_gt.numpy()))
    # silhouette: everything within 110 px of the projected object centre is foreground
    u = t_obj[0] / t_obj[2] * PROJ["fx"] + PROJ["cx"]
    v = t_obj[1] / t_obj[2] * PROJ["fy"] + PROJ["cy"]
    yy, xx = np.mgrid[0:PROJ["h"], 0:PROJ["w"]]
    fg = (xx - u) ** 2 + (yy - v) ** 2 < 110 ** 2
    return hand, R_obj, t_obj, frames, fg


def main():
    _, oh = import_reference()
    hand, R_obj, t_obj, frames, fg = scenario()
    vol = make_volume(RES, STRIDE, "capsule", np.float16)
    g = torch.Generator().manual_seed(5)
    pre = torch.randn(P, 16, generator=g)
    pre[0] = 0

    o = object.__new__(oh.gf_optimize_hand_pose)
    o.ncomps, o.optimize_dim, o.particle_size, o.iteration = 10, 16, P, 5
    o.root_dir, o.energy_weight, o.device = "", dict(ENERGY_WEIGHT), "cpu"
    o.theta_scale, o.beta, o.scaling_coefficient2 = 30, 0.9, 0.1
    o.volume_size, o.voxel_scale = RES, STRIDE
    o.initial_scale = torch.ones(16) * 0.005
    o.mano_layer_right = hand
    o.pre_sampled_particle = pre.clone()
    o.tips_region, o.finger_mask = [], []
    for i in range(5):
        prev = len(o.tips_region)
        o.tips_region.extend(hand.contact_zones[i + 1])
        o.finger_mask.append(list(range(prev, len(o.tips_region))))
    o.sdf_volume = torch.from_numpy(vol).reshape(RES, RES, RES)
    o.data_config, o.dataset_name = "data_info_SimGrasp.yml", "SimGrasp"
    # the reference reads <root>/masks/<category>/seq/<file>.png and calls `maskimg.sum(axis=-1) == 0` background (:323-326)
    img = np.repeat(fg[:, :, None].astype(np.uint8) * 255, 3, axis=2)
    oh.cv2.imread = lambda path, *a: img
    proj = {k: np.array([v]) for k, v in PROJ.items()}
    obj_pose = {"rotation": torch.from_numpy(R_obj)[None], "translation": torch.from_numpy(t_obj).reshape(1, 3, 1)}

    out = {"pre_sampled_particle": pre.numpy(), "volume": vol, "meta": np.array([RES, STRIDE]), "R_obj": R_obj, "t_obj": t_obj,
           "background_mask": ~fg, "proj": np.array([PROJ[k] for k in ("fx", "fy", "cx", "cy", "w", "h")], np.float64)}
    last = None
    with torch.no_grad():
        # all energy terms at once: frame 0's first candidate set
        f0 = frames[0]
        o.set_init_para(torch.from_numpy(f0["init_mano"]), {"rotation": torch.from_numpy(f0["init_rot"]), "translation": torch.from_numpy(f0["init_trans"])},
                        torch.from_numpy(f0["init_kp"]), None, torch.from_numpy(f0["vis_mask"]), obj_pose, "bottle", "seq/0000", torch.zeros(10), proj)
        sp = o.pre_sampled_particle * o.initial_scale
        sample = torch.cat([torch.sqrt(1 - sp[:, 0] ** 2 - sp[:, 1] ** 2 - sp[:, 2] ** 2).unsqueeze(1), sp], 1)
        hand_v, kp = o.get_kp_from_delta(sample)
        out["e0_energy"] = o.evaluate(hand_v, kp).float().numpy()
        out["e0_penetration"] = o.get_penetration_loss(o.query_sdf(hand_v)).float().numpy()
        for f, fr in enumerate(frames):
            res = o.optimize(torch.from_numpy(fr["init_mano"]), {"rotation": torch.from_numpy(fr["init_rot"]), "translation": torch.from_numpy(fr["init_trans"])},
                             torch.from_numpy(fr["init_kp"]), last, torch.from_numpy(fr["vis_mask"]), obj_pose, "bottle", f"seq/{f:04d}",
                             torch.zeros(10), proj)
            final_kp, theta, R, t = [x.detach().clone() for x in res]
            for k in ("init_mano", "init_rot", "init_trans", "init_kp", "vis_mask", "gt_kp"):
                out[f"f{f}_{k}"] = fr[k]
            out[f"f{f}_last_kp"] = np.zeros((0,), np.float32) if last is None else last.numpy()
            out[f"f{f}_final_kp"], out[f"f{f}_theta"], out[f"f{f}_R"], out[f"f{f}_t"] = final_kp.numpy(), theta.numpy(), R.numpy(), t.numpy()
            last = final_kp  # the tracker hands frame t-1's keypoints to frame t (track_network.py:205,:213)
    np.savez_compressed(os.path.join(HERE, "hand_opt_sequence.npz"), **out)
    errs = [float(np.linalg.norm(out[f"f{f}_final_kp"] - out[f"f{f}_gt_kp"], axis=-1).mean()) for f in range(FRAMES)]
    init_errs = [float(np.linalg.norm(out[f"f{f}_init_kp"] - out[f"f{f}_gt_kp"], axis=-1).mean()) for f in range(FRAMES)]
    rep = {"particles": P, "frames": FRAMES, "penetrating_candidates_frame0": int((out["e0_penetration"] > 0).sum()),
           "mean_kp_error_init_m": init_errs, "mean_kp_error_optimised_m": errs,
           "energy_frame0_min_max": [float(out["e0_energy"].min()), float(out["e0_energy"].max())]}
    json.dump(rep, open(os.path.join(HERE, "GOLDEN_REPORT_HAND.json"), "w"), indent=1)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
