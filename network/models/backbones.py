"""PointNet++ MSG backbone of HandTrackNet (counterpart of the reference's backbones.py).

`PointNet2Msg_fast` (reference backbones.py:74-133) is the one HandTrackNet uses; the
reference reshapes (B,C,N) to (B,1,C,N) and folds the unit "parts" axis back into the batch
inside every layer -- numerically a no-op.  Here both `PointNet2Msg` and `PointNet2Msg_fast`
run the 3-D modules directly; parameter names (sa1/sa2/sa3/fp3/fp2/fp1/conv1/bn1) and
channel orders are the reference's, so its checkpoints load unchanged.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet_utils
from .pointnet_utils import (PointNetFeaturePropagation, PointNetSetAbstraction, PointNetSetAbstractionMsg)


class PointNet2Msg(nn.Module):
    def __init__(self, cfg, out_dim, net_type="camera", use_xyz_feat=False, init_feature_dim=0):
        super().__init__()
        net = cfg["pointnet"][net_type]
        self.out_dim = out_dim
        self.use_xyz_feat = use_xyz_feat
        self.in_dim = init_feature_dim + 3 if use_xyz_feat else init_feature_dim
        self.sa1 = PointNetSetAbstractionMsg(net["sa1"]["npoint"], net["sa1"]["radius_list"], net["sa1"]["nsample_list"],
                                             self.in_dim + 3, net["sa1"]["mlp_list"])
        self.sa2 = PointNetSetAbstractionMsg(net["sa2"]["npoint"], net["sa2"]["radius_list"], net["sa2"]["nsample_list"],
                                             self.sa1.out_channel + 3, net["sa2"]["mlp_list"])
        self.sa3 = PointNetSetAbstraction(None, None, None, self.sa2.out_channel + 3, net["sa3"]["mlp"], group_all=True)
        self.fp3 = PointNetFeaturePropagation(self.sa2.out_channel + self.sa3.out_channel, net["fp3"]["mlp"])
        self.fp2 = PointNetFeaturePropagation(self.sa1.out_channel + self.fp3.out_channel, net["fp2"]["mlp"])
        self.fp1 = PointNetFeaturePropagation(self.in_dim + 3 + self.fp2.out_channel, net["fp1"]["mlp"])
        self.conv1 = nn.Conv1d(self.fp1.out_channel, out_dim, 1)
        self.bn1 = nn.BatchNorm1d(out_dim)
        self.device = cfg["device"]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x (B, 3[+F], N) -> per-point features (B, out_dim, N)."""
        l0_xyz = x[:, :3].contiguous()
        l0_points = x if self.use_xyz_feat else x[:, 3:]
        if l0_points.shape[1] == 0:
            l0_points = None
        l1_xyz, l1_points = self.sa1(l0_xyz, l0_points)
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points)
        l3_xyz, l3_points = self.sa3(l2_xyz, l2_points)
        l2_points = self.fp3(l2_xyz, l3_xyz, l2_points, l3_points)
        l1_points = self.fp2(l1_xyz, l2_xyz, l1_points, l2_points)
        # fp1 skip input = [xyz | input features]; with zero feature channels just xyz (backbones.py:127-130)
        skip = l0_xyz if l0_points is None else torch.cat([l0_xyz, l0_points], dim=1)
        l0_out = self.fp1(l0_xyz, l1_xyz, skip, l1_points)
        fused = pointnet_utils.fused_backend()
        if fused is not None and l0_out.is_cuda and not torch.is_grad_enabled() and not self.training:
            return fused.conv_bn_relu(l0_out, self.conv1, self.bn1)
        return F.relu(self.bn1(self.conv1(l0_out)))


class PointNet2Msg_fast(PointNet2Msg):
    """Same network; the name HandTrackNet imports (reference backbones.py:74)."""
