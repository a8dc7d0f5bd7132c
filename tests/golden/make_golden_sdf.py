"""Generates tests/golden/sdf_*.npz from the IMPORTED reference.  RUNS ONLY IN THE BUILD
CONTAINER (needs /root/reference); tests and the GPU box only read the .npz files.

The reference's optimisers import on CPU with harness-side stand-ins for modules this image lacks (cv2,
open3d, chumpy, ... -- none is used by the functions exercised here) and for the dataset file
optimization_obj.py:12 loads at import time.  No reference file is touched or copied.  Objects are created
with object.__new__ (their __init__ needs the DeepSDF checkpoints) and given exactly the attributes the
methods read.

  sdf_distance.npz   gf_optimize_obj.Distance / evaluate                 (optimization_obj.py:184-237)
  sdf_optimize.npz   gf_optimize_obj.optimize, update_shape_flag False   (optimization_obj.py:244-301)
  sdf_query.npz      gf_optimize_hand_pose.query_sdf / get_penetration_loss (optimization_hand.py:252-268)
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"

from oracle import sdf_oracle as S  # noqa: E402
from _sdf_cases import make_volume, object_points, particles, random_pose, hand_particles  # noqa: E402


def import_reference():
    class _Stub(types.ModuleType):
        def __getattr__(self, item):
            if item.startswith("__"):
                raise AttributeError(item)
            return type(item, (), {})

    for name in ("chumpy", "cv2", "open3d", "trimesh", "plyfile", "skimage", "skimage.measure", "transforms3d"):
        sys.modules.setdefault(name, _Stub(name))
    sys.path[:0] = [REF, os.path.join(REF, "network"), os.path.join(REF, "network", "models")]
    orig = np.load
    np.load = lambda *a, **k: np.array({}, dtype=object) if "CatPose2InsPose" in str(a[0]) else orig(*a, **k)
    try:
        import optimization_obj as oo
        import optimization_hand as oh
    finally:
        np.load = orig
    return oo, oh


def ref_obj(oo, vol, res, stride, pre=None, iterations=10):
    o = object.__new__(oo.gf_optimize_obj)
    o.volume_size, o.voxel_scale = res, stride
    o.sdf_volume = torch.from_numpy(vol).reshape(res, res, res)
    o.device = "cpu"
    o.update_shape_flag = False
    o.iteration, o.scaling_coefficient1, o.scaling_coefficient2, o.beta = iterations, 0.02, 2, 0.9
    if pre is not None:
        o.pre_sampled_particle = torch.from_numpy(pre)
        o.particle_size = pre.shape[0]
    return o


def main():
    oo, oh = import_reference()
    report = {}

    # ---- Distance / evaluate -------------------------------------------------------------------------------
    out = {}
    exact = total = 0
    worst_e = 0.0
    for ci, (res, stride, dt, shape) in enumerate([(41, 0.01, np.float16, "sphere"), (41, 0.01, np.float32, "box"),
                                                   (21, 0.02, np.float16, "box"), (81, 0.005, np.float16, "capsule")]):
        vol = make_volume(res, stride, shape, dt)
        o = ref_obj(oo, vol, res, stride)
        rng = np.random.default_rng(500 + ci)
        # queries: inside the box, outside it (clamped), exactly on voxel centres and on the top face
        V = np.concatenate([
            rng.uniform(-0.2, 0.2, (3000, 3)), rng.uniform(-0.35, 0.35, (500, 3)),
            (rng.integers(0, res, (300, 3)) * stride - 0.2), np.full((4, 3), 0.2), np.full((4, 3), -0.2),
            np.array([[0.2, 0.0, 0.0], [0.0, 0.2, 0.0], [0.0, 0.0, 0.2], [0.2, 0.2, -0.2]]),
        ]).astype(np.float32)
        ref = o.Distance(torch.from_numpy(V)).numpy()
        mine = S.distance(V, vol, stride)
        exact += int((ref.view(np.int32) == mine.view(np.int32)).sum())
        total += ref.size
        out[f"d{ci}_vol"], out[f"d{ci}_V"], out[f"d{ci}_ref"] = vol, V, ref
        out[f"d{ci}_meta"] = np.array([res, stride])
        # evaluate: P particles around a pose
        pcld = object_points(600 + ci, 256, shape)
        R0, t0 = random_pose(700 + ci)
        pcld_cam = (pcld @ R0.T + t0).astype(np.float32)  # object frame -> camera frame
        rot, trans = particles(800 + ci, 64, R0, t0)
        e_ref, s_ref = o.evaluate(torch.from_numpy(pcld_cam)[None], torch.from_numpy(rot), torch.from_numpy(trans)[:, :, None])
        s_mine = S.particle_energy(pcld_cam, rot, trans, vol, stride)
        worst_e = max(worst_e, float(np.abs(s_ref.numpy() - s_mine).max()))
        out[f"e{ci}_pcld"], out[f"e{ci}_rot"], out[f"e{ci}_trans"] = pcld_cam, rot, trans
        out[f"e{ci}_sdf_energy"], out[f"e{ci}_energy"] = s_ref.numpy(), e_ref.numpy()
    report["distance_bit_exact"] = (exact, total)
    report["evaluate_max_abs_err"] = worst_e
    np.savez_compressed(os.path.join(HERE, "sdf_distance.npz"), **out)

    # ---- optimize loop -------------------------------------------------------------------------------------
    out = {}
    worst_r = worst_t = 0.0
    for ci, (res, stride, shape, P, N) in enumerate([(41, 0.01, "box", 512, 256), (81, 0.005, "capsule", 2048, 512)]):
        vol = make_volume(res, stride, shape, np.float16)
        rng = np.random.default_rng(900 + ci)
        pre = rng.standard_normal((P, 6)).astype(np.float32)
        pre[0] = 0
        o = ref_obj(oo, vol, res, stride, pre)
        pcld = object_points(910 + ci, N, shape)
        R_gt, t_gt = random_pose(920 + ci)
        pcld_cam = (pcld @ R_gt.T + t_gt).astype(np.float32)
        # jittered initial pose (the tracker hands over last frame's pose)
        dR, dt = random_pose(930 + ci, angle=0.06, trans=0.008)
        R_init = (R_gt @ dR).astype(np.float32)
        t_init = (t_gt + dt).astype(np.float32)
        init = {"rotation": torch.from_numpy(R_init)[None], "translation": torch.from_numpy(t_init).reshape(1, 3, 1)}
        proj = {"w": [640], "h": [480]}
        ret = o.optimize(torch.from_numpy(pcld_cam)[None], init, "cat", "file", proj)
        R_ref = ret["rotation"].numpy().reshape(3, 3)
        t_ref = ret["translation"].numpy().reshape(3)
        R_me, t_me = S.obj_optimize(pcld_cam, R_init, t_init, pre, vol, stride)
        worst_r = max(worst_r, float(np.abs(R_ref - R_me).max()))
        worst_t = max(worst_t, float(np.abs(t_ref - t_me).max()))
        e0 = S.particle_energy(pcld_cam, R_init[None], t_init[None], vol, stride)[0]
        e1 = S.particle_energy(pcld_cam, R_ref[None], t_ref[None], vol, stride)[0]
        report[f"optimize{ci}_sdf_energy_before_after"] = (float(e0), float(e1))
        for k, v in dict(vol=vol, pre=pre, pcld=pcld_cam, R_init=R_init, t_init=t_init, R_ref=R_ref, t_ref=t_ref,
                         meta=np.array([res, stride])).items():
            out[f"o{ci}_{k}"] = v
    report["optimize_max_abs_err_R_t"] = (worst_r, worst_t)
    np.savez_compressed(os.path.join(HERE, "sdf_optimize.npz"), **out)

    # ---- query_sdf / penetration ---------------------------------------------------------------------------
    out = {}
    same = total = 0
    pen_same = pen_total = 0
    for ci, (res, scale, dt, shape) in enumerate([(31, 0.015, np.float16, "sphere"), (51, 0.009, np.float16, "box"),
                                                  (31, 0.015, np.float32, "capsule")]):
        vol = make_volume(res, scale, shape, dt, centre_index=res // 2)
        h = object.__new__(oh.gf_optimize_hand_pose)
        h.volume_size, h.voxel_scale = res, scale
        h.sdf_volume = torch.from_numpy(vol).reshape(res, res, res)
        R0, t0 = random_pose(1000 + ci)
        h.obj_r = torch.from_numpy(R0)
        h.obj_t = torch.from_numpy(t0).reshape(1, 1, 3)
        hand = hand_particles(1100 + ci, 96, 200, R0, t0, extent=res // 2 * scale * 1.3)
        q_ref = h.query_sdf(torch.from_numpy(hand))
        p_ref = h.get_penetration_loss(q_ref)
        idx, sdf, pen = S.nearest(hand, R0, t0, vol, scale)
        same += int((q_ref.numpy() == sdf).sum())
        total += sdf.size
        pen_same += int((p_ref.numpy() == pen).sum())
        pen_total += pen.size
        for k, v in dict(vol=vol, hand=hand, obj_r=R0, obj_t=t0, sdf_ref=q_ref.numpy(), pen_ref=p_ref.numpy(),
                         meta=np.array([res, scale])).items():
            out[f"q{ci}_{k}"] = v
    report["query_sdf_same_value"] = (same, total)
    report["penetration_same_value"] = (pen_same, pen_total)
    # the floor-division helper against torch's own `//`
    a = np.concatenate([np.random.default_rng(7).uniform(-0.3, 0.3, 200000), np.arange(-80, 80) * 0.003]).astype(np.float32)
    t = (torch.from_numpy(a) // 0.003).numpy()
    report["div_floor_vs_torch"] = (int((t == S.div_floor(a, 0.003)).sum()), int(a.size))
    np.savez_compressed(os.path.join(HERE, "sdf_query.npz"), **out)

    with open(os.path.join(HERE, "GOLDEN_REPORT_SDF.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
