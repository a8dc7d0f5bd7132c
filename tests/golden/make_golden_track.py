"""Generates tests/golden/track_sequence.npz from the IMPORTED reference tracking loop.
RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference); tests and the GPU box only read the .npz.

What runs is the reference's own `HandTrackModel.forward` (network/models/track_network.py:139-226, the
HandTrackNet-only branch :214-217): frame t is seeded with `last_frame_kp + mean(hand_points)` (:163) and hands
`pred_kp - mean(hand_points)` to frame t+1 (:217), the palm template is fixed for the sequence (:150-152).
Harness-side stand-ins, no reference file touched: stub modules for packages this image lacks (none is used by
the branch exercised), the dataset file optimization_obj.py:12 loads at import time, and -- because the MANO layer
needs licensed assets -- `self.manolayer` is a callable returning a fixed seeded 21-keypoint rest pose, from which
the reference's own `handkp2palmkp` derives the template.  The object is created with object.__new__ (its __init__
builds MANO / DeepSDF objects) and given exactly the attributes forward() reads.  The operator functions are the
CUDA-semantics oracle, as in make_golden.py.
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

from oracle import torch_ops  # noqa: E402
from _netinit import deterministic_init, make_cfg  # noqa: E402
from datasets.synthetic import SyntheticSequences  # noqa: E402

FRAMES, NUM_POINTS = 6, 1024
HEAD_SCALE = 0.01  # random weights predict ~6 hand-frame units of offset per frame; a trained head predicts centimetres


def import_reference():
    class _Stub(types.ModuleType):
        def __getattr__(self, item):
            if item.startswith("__"):
                raise AttributeError(item)
            return type(item, (), {})

    for name in ("chumpy", "cv2", "open3d", "trimesh", "plyfile", "skimage", "skimage.measure", "transforms3d",
                 "transforms3d.quaternions", "transforms3d.euler", "transforms3d.axangles", "tensorboardX"):
        sys.modules.setdefault(name, _Stub(name))
    sys.path[:0] = [REF, os.path.join(REF, "network"), os.path.join(REF, "network", "models")]
    torch.Tensor.cuda = lambda self, *a, **k: self  # transformer.py:110 hard-codes .cuda()
    orig = np.load
    np.load = lambda *a, **k: np.array({}, dtype=object) if "CatPose2InsPose" in str(a[0]) else orig(*a, **k)
    try:
        import pointnet_utils as ref_pu
        import hand_network as ref_hn
        import track_network as ref_tn
    finally:
        np.load = orig
    return ref_pu, ref_hn, ref_tn


def main():
    ref_pu, ref_hn, ref_tn = import_reference()
    ref_pu.CUDA = True
    ref_pu.futils = torch_ops
    cfg = make_cfg("cpu")
    cfg.update(num_points=NUM_POINTS, hand_jitter_cfg={"rand_scale": 0.01})
    model = object.__new__(ref_tn.HandTrackModel)
    torch.nn.Module.__init__(model)
    model.device, model.use_optimization, model.IKnet = "cpu", False, None
    model.handnet = ref_hn.HandTrackNet(cfg)
    deterministic_init(model)  # keys 'handnet.*', as in the checkpoint layout trainer.py:206-215 builds
    with torch.no_grad():  # keep the sequence in the tracking regime (prediction stays on the hand): scale the last layer
        for p in model.handnet.final_mlp[2].parameters():
            p.mul_(HEAD_SCALE)
    model.eval()
    rest = torch.from_numpy(np.random.default_rng(77).normal(0, 0.04, (1, 21, 3)).astype(np.float32))
    model.manolayer = lambda **kw: (None, rest)
    seq = SyntheticSequences(cfg, 1, FRAMES)[0]
    flags = {"track_flag": True, "test_flag": True, "save_flag": False, "IKNet_flag": False}
    with torch.no_grad():
        rets = model(seq, flags)
    out = {"head_scale": np.float32(HEAD_SCALE), "palm_template": ref_tn.handkp2palmkp(rest).numpy(),
           "hand_points": np.stack([f["hand_points"][0].numpy() for f in seq]),
           "first_jittered_kp": seq[0]["jittered_hand_kp"].numpy(),
           "jittered_kp_seen": np.stack([f["jittered_hand_kp"][0].numpy() for f in seq]),  # what forward() wrote back
           "pred_kp": np.stack([r["pred_kp"][0].numpy() for r in rets]),
           "rotation": np.stack([r["canon_pose"]["rotation"][0].numpy() for r in rets]),
           "translation": np.stack([r["canon_pose"]["translation"][0].numpy() for r in rets])}
    np.savez_compressed(os.path.join(HERE, "track_sequence.npz"), **out)
    rep = {"frames": FRAMES, "num_points": NUM_POINTS,
           "pred_kp_drift_first_to_last": float(np.abs(out["pred_kp"][-1] - out["pred_kp"][0]).max())}
    with open(os.path.join(HERE, "GOLDEN_REPORT_TRACK.json"), "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep, indent=1))


def main_obj(ref_tn):
    """tests/golden/track_obj_sequence.npz: the reference's ObjTrackModel_Optimization.forward (track_network.py:338-383) on a
    synthetic object sequence.  Harness-side: `load_obj_for_opt` (reads DeepSDF / mesh assets) returns a placeholder tuple,
    the optimiser is a gf_optimize_obj created with object.__new__ (its __init__ loads the decoder) holding an analytic
    capsule volume and seeded particles, its `load_obj` (decodes the latent into that volume) is a no-op.  The per-frame
    loop, the pose hand-off and `optimize` are the reference's own code."""
    import optimization_obj as oo
    from _sdf_cases import make_volume, object_points, random_pose, _axis_angle
    from oracle import sdf_oracle as S
    res, stride, P, N, F = 81, 0.005, 512, 256, 5
    vol = make_volume(res, stride, "capsule", np.float16)
    rng = np.random.default_rng(4242)
    pre = rng.standard_normal((P, 6)).astype(np.float32)
    pre[0] = 0
    o = object.__new__(oo.gf_optimize_obj)
    o.volume_size, o.voxel_scale, o.device, o.update_shape_flag = res, stride, "cpu", False
    o.sdf_volume = torch.from_numpy(vol).reshape(res, res, res)
    o.iteration, o.scaling_coefficient1, o.scaling_coefficient2, o.beta = 10, 0.02, 2, 0.9
    o.pre_sampled_particle, o.particle_size = torch.from_numpy(pre), P
    o.load_obj = lambda *a, **k: None
    model = object.__new__(ref_tn.ObjTrackModel_Optimization)
    torch.nn.Module.__init__(model)
    model.device, model.optimizer, model.root_dir, model.dataset_name, model.sdf_code_source = "cpu", o, "", "HO3D", "pred"
    ref_tn.load_obj_for_opt = lambda *a, **k: (None, {"scale": [1.0]}, None, "gt_path", "recon_path")
    R, t = random_pose(31)
    R, t = R.astype(np.float64), t.astype(np.float64)
    dR, vel = _axis_angle(rng.standard_normal(3), 0.012), rng.normal(0, 0.002, 3)
    Rj, tj = random_pose(32, angle=0.06, trans=0.008)
    seq, pts, gts = [], [], []
    for k in range(F):
        p = (object_points(5000 + k, N, "capsule").astype(np.float64) @ R.T + t).astype(np.float32)
        fr = {"obj_points": torch.from_numpy(p)[None], "category": ["bottle"], "file_name": [f"seq/{k:04d}"],
              "gt_obj_pose": {"rotation": torch.from_numpy(R.astype(np.float32)), "translation": torch.from_numpy(t.astype(np.float32))},
              "projection": {"w": [640], "h": [480]}}
        if k == 0:
            fr["jittered_obj_pose"] = {"rotation": torch.from_numpy((R @ Rj).astype(np.float32)),
                                       "translation": torch.from_numpy((t + tj).astype(np.float32))}
        seq.append(fr); pts.append(p); gts.append((R.astype(np.float32), t.astype(np.float32)))
        R, t = R @ dR, t + vel
    init_R, init_t = seq[0]["jittered_obj_pose"]["rotation"].numpy().copy(), seq[0]["jittered_obj_pose"]["translation"].numpy().copy()
    with torch.no_grad():
        rets = model(seq, {"track_flag": True, "test_flag": True, "save_flag": False})
    R_ref = np.stack([r["rotation"].numpy().reshape(3, 3) for r in rets])
    t_ref = np.stack([r["translation"].numpy().reshape(3) for r in rets])
    # the oracle chained the same way
    Ro, to, worst = init_R.reshape(3, 3), init_t.reshape(3), 0.0
    for k in range(F):
        Ro, to = S.obj_optimize(pts[k], Ro, to, pre, vol, stride)
        worst = max(worst, float(np.abs(Ro - R_ref[k]).max()), float(np.abs(to - t_ref[k]).max()))
    np.savez_compressed(os.path.join(HERE, "track_obj_sequence.npz"), vol=vol, meta=np.array([res, stride]), pre=pre, pts=np.stack(pts),
                        init_R=init_R, init_t=init_t, R_ref=R_ref, t_ref=t_ref,
                        gt_R=np.stack([g[0] for g in gts]), gt_t=np.stack([g[1] for g in gts]))
    err_t = [float(np.linalg.norm(t_ref[k] - gts[k][1])) for k in range(F)]
    return {"obj_frames": F, "obj_oracle_vs_reference_max_abs": worst, "obj_translation_error_per_frame_m": err_t}


if __name__ == "__main__":
    assert os.path.isdir(REF), "golden vectors can only be regenerated where /root/reference exists"
    main()
    rep = main_obj(sys.modules["track_network"])
    with open(os.path.join(HERE, "GOLDEN_REPORT_TRACK.json")) as f:
        full = json.load(f)
    full.update(rep)
    with open(os.path.join(HERE, "GOLDEN_REPORT_TRACK.json"), "w") as f:
        json.dump(full, f, indent=1)
    print(json.dumps(rep, indent=1))
