"""Seeded synthetic hand clouds standing in for SimGrasp / HO3D / DexYCB, whose data
is not available offline.  Mirrors what the reference loaders hand to the network
(`datasets/dataset.py`: SingleFrameData for training, SequenceData for tracking tests):
dict(hand_points (N,3), jittered_hand_kp (21,3), gt_hand_kp (21,3), gt_hand_pose.palm_template (6,3)).

A frame is a hand-sized Gaussian blob clipped to a 0.15 m ball at z = 0.5 m plus 21 keypoints; the
initial keypoints are the ground truth + N(0, rand_scale^2) jitter (config hand_jitter_cfg)."""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset

PALM = [0, 1, 5, 9, 13, 17]


def make_frame(seed: int, num_points: int, jitter: float, motion: np.ndarray | None = None):
    rng = np.random.default_rng(seed)
    pts = rng.normal(0, 0.05, (num_points, 3))
    r = np.linalg.norm(pts, axis=-1, keepdims=True)
    pts = np.where(r > 0.15, pts * 0.15 / np.maximum(r, 1e-9), pts)
    off = np.array([0.0, 0.0, 0.5]) + (0 if motion is None else motion)
    gt = rng.normal(0, 0.04, (21, 3))
    f = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    return {"hand_points": f(pts + off), "gt_hand_kp": f(gt + off),
            "jittered_hand_kp": f(gt + off + rng.normal(0, jitter, (21, 3))),
            "gt_hand_pose": {"palm_template": f(gt[PALM] - gt[:1])}}


class SyntheticFrames(Dataset):
    """Independent frames (training)."""

    def __init__(self, cfg, length: int, base_seed: int = 0):
        self.n, self.len, self.seed = cfg["num_points"], length, base_seed
        self.jitter = cfg["hand_jitter_cfg"]["rand_scale"]

    def __len__(self):
        return self.len

    def __getitem__(self, i):
        return make_frame(self.seed + i, self.n, self.jitter)


class SyntheticSequences(Dataset):
    """Temporal sequences (tracking test): the hand drifts smoothly; each item is a list of frames with a
    leading batch dimension of 1, like the reference's SequenceData collate."""

    def __init__(self, cfg, num_sequences: int, frames: int):
        self.cfg, self.ns, self.nf = cfg, num_sequences, frames

    def __len__(self):
        return self.ns

    def __getitem__(self, s):
        rng = np.random.default_rng(10_000 + s)
        vel = rng.normal(0, 0.002, 3)
        base = make_frame(20_000 + s, self.cfg["num_points"], self.cfg["hand_jitter_cfg"]["rand_scale"])
        seq = []
        for t in range(self.nf):
            d = torch.from_numpy((vel * t).astype(np.float32))
            noise = make_frame(30_000 + 1000 * s + t, self.cfg["num_points"], 0.0)
            fr = {"hand_points": (noise["hand_points"] + d).unsqueeze(0), "gt_hand_kp": (base["gt_hand_kp"] + d).unsqueeze(0),
                  "jittered_hand_kp": (base["jittered_hand_kp"] + d).unsqueeze(0),
                  "gt_hand_pose": {"palm_template": base["gt_hand_pose"]["palm_template"].unsqueeze(0)},
                  "file_name": [f"synthetic_{s:03d}_{t:03d}"]}
            seq.append(fr)
        return seq


def get_dataloader(cfg, mode="train", shuffle=False, num_workers=0, distributed=False, length=None):
    syn = cfg["data_cfg"].get("synthetic", {})
    if cfg.get("track") == "hand_IKNet" and cfg.get("use_optimization") and cfg.get("hand_model") is not None:
        ds = SyntheticHandObjectSequences(cfg, syn.get("test_sequences", 2), syn.get("sequence_frames", 20) if length is None else length)
        return torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, collate_fn=lambda b: b[0])
    if cfg.get("track") == "obj_opt":
        ds = SyntheticObjectSequences(cfg, syn.get("test_sequences", 2), syn.get("sequence_frames", 30) if length is None else length)
        return torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, collate_fn=lambda b: b[0])
    if cfg.get("track"):
        ds = SyntheticSequences(cfg, syn.get("test_sequences", 4),
                                syn.get("sequence_frames", cfg["data_cfg"].get("num_frames", 100)) if length is None else length)
        return torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, collate_fn=lambda b: b[0])
    n = length or (syn.get("train_frames", 2048) if mode == "train" else max(cfg["batch_size"] * 4, 64))
    ds = SyntheticFrames(cfg, n, base_seed=0 if mode == "train" else 1_000_000)
    sampler = torch.utils.data.distributed.DistributedSampler(ds, shuffle=shuffle) if distributed else None
    return torch.utils.data.DataLoader(ds, batch_size=cfg["batch_size"], shuffle=shuffle and sampler is None, sampler=sampler,
                                       num_workers=num_workers, drop_last=(mode == "train"))


# ---- object sequences (track: obj_opt) ---------------------------------------------------------------------------------
def _capsule_sdf(p):
    """Signed distance to a bottle-sized capsule (radius 4 cm, 14 cm core) in the object frame, metres."""
    z = np.clip(p[..., 2], -0.07, 0.07)
    return np.sqrt(p[..., 0] ** 2 + p[..., 1] ** 2 + (p[..., 2] - z) ** 2) - 0.04


def capsule_volume(res: int = 201, stride: float = 0.002) -> np.ndarray:
    """(res,res,res) fp16 SDF volume on the +-0.2 m box the reference voxelises its DeepSDF decoder on
    (optimization_obj.py:133-143: 201^3, stride 0.002, clamped), indexed [ix,iy,iz]."""
    ax = (np.arange(res) - res // 2) * float(stride)
    g = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), axis=-1)
    return np.clip(_capsule_sdf(g), -0.1, 0.1).astype(np.float16)


def _rot(axis, angle):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


def capsule_surface(rng, n, noise=0.0015):
    """n points on the capsule surface (object frame) + sensor noise."""
    z = rng.uniform(-0.11, 0.11, n)
    phi = rng.uniform(0, 2 * np.pi, n)
    zc = np.clip(z, -0.07, 0.07)
    r = np.sqrt(np.maximum(0.04 ** 2 - (z - zc) ** 2, 0.0))
    p = np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=-1)
    return p + rng.normal(0, noise, p.shape)


class SyntheticObjectSequences(Dataset):
    """Sequences for `track: obj_opt` (reference SequenceData items as ObjTrackModel_Optimization.forward reads them,
    track_network.py:338-383): per frame obj_points (1,N,3) camera frame, gt_obj_pose, category / file_name / projection;
    frame 0 also carries jittered_obj_pose (obj_jitter_cfg: r degrees, t metres) and -- in place of the DeepSDF latent the
    reference decodes -- the object's SDF volume."""

    def __init__(self, cfg, num_sequences: int, frames: int, res: int = 201, stride: float = 0.002):
        self.cfg, self.ns, self.nf = cfg, num_sequences, frames
        self.res, self.stride = res, stride
        self._vol = None

    def __len__(self):
        return self.ns

    def __getitem__(self, s):
        if self._vol is None:
            self._vol = torch.from_numpy(capsule_volume(self.res, self.stride))
        rng = np.random.default_rng(40_000 + s)
        n = self.cfg["num_points"]
        jit = self.cfg.get("obj_jitter_cfg", {})
        R = _rot(rng.standard_normal(3), rng.uniform(0, np.pi))
        t = np.array([0.0, 0.0, 0.5]) + rng.uniform(-0.05, 0.05, 3)
        w_axis, w = rng.standard_normal(3), rng.normal(0, 0.01)   # per-frame rotation increment (rad) and velocity (m)
        vel = rng.normal(0, 0.002, 3)
        f = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
        seq = []
        for k in range(self.nf):
            pts = capsule_surface(rng, n) @ R.T + t
            fr = {"obj_points": f(pts).unsqueeze(0), "category": [self.cfg["obj_category"][0]], "file_name": [f"synthetic_obj_{s:03d}/{k:04d}"],
                  "gt_obj_pose": {"rotation": f(R).reshape(1, 1, 3, 3), "translation": f(t).reshape(1, 1, 3, 1)},
                  "projection": {"w": [640], "h": [480]}}
            if k == 0:
                ang = np.deg2rad(float(jit.get("r", 5))) * rng.standard_normal()
                Rj = R @ _rot(rng.standard_normal(3), ang)
                tj = t + rng.normal(0, float(jit.get("t", 0.03)) / 3, 3)
                fr["jittered_obj_pose"] = {"rotation": f(Rj).reshape(1, 3, 3), "translation": f(tj).reshape(1, 3, 1)}
                fr["sdf_volume"], fr["voxel_scale"] = self._vol, self.stride
            seq.append(fr)
            R = R @ _rot(w_axis, w)
            t = t + vel
        return seq


class SyntheticHandObjectSequences(Dataset):
    """Sequences for `track: hand_IKNet` with `use_optimization` and a hand model (HandTrackModel's particle-optimisation
    branch, reference track_network.py:142-156, :203-211): a hand (cfg['hand_model'], models/hand_model.HandModel) grasping the
    synthetic capsule.  Per frame: hand_points sampled on the posed hand's vertices (+ sensor noise), gt / jittered keypoints,
    gt_hand_pose (palm template of the model's rest pose, rotation, translation), gt_obj_pose, projection, the silhouette's
    background mask; frame 0 also carries the object's SDF volume."""

    def __init__(self, cfg, num_sequences: int, frames: int, res: int = 151, stride: float = 0.003):
        self.cfg, self.ns, self.nf, self.res, self.stride = cfg, num_sequences, frames, res, stride
        self.hand = cfg["hand_model"]
        self._vol = None

    def __len__(self):
        return self.ns

    def __getitem__(self, s):
        if self._vol is None:
            self._vol = torch.from_numpy(capsule_volume(self.res, self.stride))
        rng = np.random.default_rng(50_000 + s)
        n = self.cfg["num_points"]
        jitter = self.cfg["hand_jitter_cfg"]["rand_scale"]
        f = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
        R_obj = _rot(rng.standard_normal(3), rng.uniform(0, np.pi))
        t_obj = np.array([0.0, 0.0, 0.5]) + rng.uniform(-0.03, 0.03, 3)
        proj = dict(fx=600.0, fy=600.0, cx=320.0, cy=240.0, w=640, h=480)
        u, v = t_obj[0] / t_obj[2] * proj["fx"] + proj["cx"], t_obj[1] / t_obj[2] * proj["fy"] + proj["cy"]
        yy, xx = np.mgrid[0:proj["h"], 0:proj["w"]]
        background = torch.from_numpy(~((xx - u) ** 2 + (yy - v) ** 2 < 110 ** 2))
        import copy
        hm = copy.deepcopy(self.hand).cpu()  # (the optimiser's instance lives on the GPU: nn.Module.cpu() moves in place)
        with torch.no_grad():
            _, rest_kp = hm.forward(th_pose_coeffs=torch.zeros(1, 3 + hm.num_pose), th_trans=torch.zeros(1, 3))
        palm = rest_kp[:, PALM]
        seq = []
        for k in range(self.nf):
            R_ho = _rot(np.array([0.3, 0.2, 1.0]), 0.15 + 0.02 * k)
            t_ho = np.array([0.002 * k, -0.135 + 0.002 * k, 0.045])
            R = (R_obj @ R_ho).astype(np.float32)
            t = (R_obj @ t_ho + t_obj).astype(np.float32)
            theta = (0.2 * np.sin(np.arange(hm.num_pose) * 0.7 + 0.2 * k)).astype(np.float32)
            # axis-angle of R: the hand model takes the global rotation that way
            ang = np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))
            ax = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2 * np.sin(ang) + 1e-12)
            with torch.no_grad():
                verts, kp = hm.forward(th_pose_coeffs=torch.cat([f(ax * ang)[None], f(theta)[None]], 1), th_trans=f(t)[None])
            pick = rng.integers(0, verts.shape[1], n)
            pts = verts[0, pick].numpy() + rng.normal(0, 0.0015, (n, 3))
            fr = {"hand_points": f(pts).unsqueeze(0), "gt_hand_kp": kp.clone(), "jittered_hand_kp": kp + f(rng.normal(0, jitter, (1, 21, 3))),
                  "gt_hand_pose": {"palm_template": palm.clone(), "rotation": f(R).reshape(1, 3, 3), "translation": f(t).reshape(1, 3, 1),
                                   "mano_pose": f(theta)[None]},
                  "gt_obj_pose": {"rotation": f(R_obj).reshape(1, 3, 3), "translation": f(t_obj).reshape(1, 3, 1)},
                  "projection": {kk: [vv] for kk, vv in proj.items()}, "background_mask": background,
                  "category": [self.cfg["obj_category"][0]], "file_name": [f"synthetic_handobj_{s:03d}/{k:04d}"]}
            if k == 0:
                fr["sdf_volume"], fr["voxel_scale"] = self._vol, self.stride
            seq.append(fr)
        return seq
