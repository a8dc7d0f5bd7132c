"""Deterministic, name-keyed parameter initialisation shared by bench.py, smoke(), the golden-vector generator
(which applies it to the imported reference model) and the tests (which apply it to ours):
identical weights by construction, no multi-MB state-dict fixture."""
import zlib

import torch


def deterministic_init(model: torch.nn.Module) -> None:
    sd = model.state_dict()
    with torch.no_grad():
        for name in sorted(sd.keys()):
            t = sd[name]
            if not t.is_floating_point():
                continue  # num_batches_tracked
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
            if name.endswith("running_var"):
                v = 0.5 + torch.rand(t.shape, generator=g)
            elif name.endswith("running_mean"):
                v = 0.1 * torch.randn(t.shape, generator=g)
            elif t.dim() >= 2:
                fan_out, fan_in = t.shape[0], t[0].numel()
                v = torch.randn(t.shape, generator=g) * (2.0 * (2.0 / (fan_in + fan_out)) ** 0.5)
            elif name.endswith("weight"):  # norm scales
                v = 1.0 + 0.1 * torch.randn(t.shape, generator=g)
            else:  # biases, in_proj_bias ...
                v = 0.05 * torch.randn(t.shape, generator=g)
            t.copy_(v.to(t.dtype))


def make_cfg(device="cpu", backbone_out_dim=384):
    """cfg dict HandTrackNet needs (what configs/config.py builds from the YAMLs)."""
    cam = {
        "sa1": {"npoint": 256, "radius_list": [0.1], "nsample_list": [32], "mlp_list": [[32, 32, 64]]},
        "sa2": {"npoint": 128, "radius_list": [0.2], "nsample_list": [32], "mlp_list": [[64, 64, 128]]},
        "sa3": {"mlp": [128, 128, 512]},
        "fp3": {"mlp": [256, 256]},
        "fp2": {"mlp": [256, 128]},
        "fp1": {"mlp": [128, 128]},
    }
    return {"device": device, "network": {"handframe": "kp", "backbone_out_dim": backbone_out_dim, "type": "HandTrackNet"},
            "pointnet": {"camera": cam}}


def synthetic_frames(seed, B, N=1024):
    """SURVEY.md 8(d) synthetic input: hand cloud ~N(0,0.05^2) clipped to 0.15 m at z=0.5 m, 21 keypoints."""
    import numpy as np
    rng = np.random.default_rng(seed)
    pts = rng.normal(0, 0.05, (B, N, 3))
    r = np.linalg.norm(pts, axis=-1, keepdims=True)
    pts = np.where(r > 0.15, pts * 0.15 / np.maximum(r, 1e-9), pts)
    off = np.array([0.0, 0.0, 0.5])
    gt_kp = rng.normal(0, 0.04, (B, 21, 3))
    jit = gt_kp + rng.normal(0, 0.01, (B, 21, 3))
    palm = gt_kp[:, [0, 1, 5, 9, 13, 17]] - gt_kp[:, :1]
    f = lambda a: torch.from_numpy(a.astype(np.float32))
    return {"hand_points": f(pts + off), "jittered_hand_kp": f(jit + off), "gt_hand_kp": f(gt_kp + off),
            "gt_hand_pose": {"palm_template": f(palm)}}
