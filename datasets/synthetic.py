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
    if cfg.get("track"):
        ds = SyntheticSequences(cfg, syn.get("test_sequences", 4), cfg["data_cfg"].get("num_frames", 100) if length is None else length)
        return torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, collate_fn=lambda b: b[0])
    n = length or (syn.get("train_frames", 2048) if mode == "train" else max(cfg["batch_size"] * 4, 64))
    ds = SyntheticFrames(cfg, n, base_seed=0 if mode == "train" else 1_000_000)
    sampler = torch.utils.data.distributed.DistributedSampler(ds, shuffle=shuffle) if distributed else None
    return torch.utils.data.DataLoader(ds, batch_size=cfg["batch_size"], shuffle=shuffle and sampler is None, sampler=sampler,
                                       num_workers=num_workers, drop_last=(mode == "train"))
