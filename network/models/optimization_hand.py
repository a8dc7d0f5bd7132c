"""Hand-pose particle optimiser -- the SDF part of the reference's gf_optimize_hand_pose
(network/models/optimization_hand.py:138-330): `query_sdf` and `get_penetration_loss` for 5120 candidate hands
x 778 MANO vertices against the object's 151^3 volume.  The rest of that class (MANO
forward, silhouette / regularisation terms) needs the MANO assets and is out of scope.
"""
from __future__ import annotations

import torch

from hotrack_amd import sdf as _sdf


class gf_optimize_hand_pose:
    def __init__(self, cfg=None, device="cuda"):
        self.particle_size = 5120  # :140
        self.volume_size = 151  # :148-149
        self.voxel_scale = 0.003
        self.device = torch.device((cfg or {}).get("device", device))
        self.sdf_volume = None
        self.obj_r = None
        self.obj_t = None

    def load_volume(self, sdf_volume: torch.Tensor, voxel_scale: float | None = None):
        V = sdf_volume.shape[0]
        assert sdf_volume.dim() == 3 and tuple(sdf_volume.shape) == (V, V, V) and V % 2 == 1
        self.volume_size = V
        if voxel_scale is not None:
            self.voxel_scale = float(voxel_scale)
        self.sdf_volume = sdf_volume.to(self.device).contiguous()

    def set_obj_pose(self, init_obj_pose):  # the two lines of set_init_para that matter here (:312-313)
        self.obj_r = init_obj_pose["rotation"].to(self.device).reshape(3, 3).float()
        self.obj_t = init_obj_pose["translation"].to(self.device).reshape(1, 1, 3).float()

    def query_sdf(self, hand):
        return _sdf.query_sdf(hand.float(), self.obj_r, self.obj_t, self.sdf_volume, self.voxel_scale)

    def get_penetration_loss(self, queried_sdf, threshold=0):
        abs_distance = queried_sdf.abs()
        penetrate_mask = (queried_sdf < -threshold).bool()
        return torch.max(abs_distance * penetrate_mask, dim=-1)[0]

    def query_sdf_and_penetration(self, hand):
        """One launch for both (threshold 0): returns (queried_sdf (B,N), penetrate_max (B,))."""
        return _sdf.query_sdf(hand.float(), self.obj_r, self.obj_t, self.sdf_volume, self.voxel_scale, with_penetration=True)
