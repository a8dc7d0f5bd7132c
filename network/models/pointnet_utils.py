"""Set-abstraction / feature-propagation modules of HandTrackNet on the MI355X operator stack.

Counterpart of the reference's network/models/pointnet_utils.py (module names, constructor
arguments, parameter names and numerics are kept so that reference checkpoints load:
`conv_blocks.i.j`, `bn_blocks.i.j`, `mlp_convs.i`, `mlp_bns.i`).  What differs:

  * every point operator (FPS, ball query, kNN, three-NN, interpolate, gather, group) runs on
    the hand-written gfx950 kernels of `hotrack_amd` -- including gather/group, which the
    reference re-routed to advanced indexing (pointnet_utils.py:100-109);
  * there is no `CUDA = torch.cuda.is_available()` switch and no pure-torch fallback: the
    operator backend is `hotrack_amd.pointnet2_utils` and a CPU tensor raises.  Tests inject
    the CPU oracle explicitly through `set_operator_backend`;
  * indices stay int32 end-to-end inside the modules (the reference converts to int64 and
    back around every call); the free functions below still return int64 like the reference;
  * in eval mode the grouped MLP + max of an SA scale runs as one fused kernel when the
    fused backend is enabled (hotrack_amd.fused), never materialising (B, C, S, K).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

_OPS = None
_FUSED = None  # hotrack_amd.fused, when enabled


def _ops():
    global _OPS
    if _OPS is None:
        from hotrack_amd import pointnet2_utils as hip_ops  # fails loudly if the extension is missing
        _OPS = hip_ops
    return _OPS


def set_operator_backend(module) -> None:
    """Replace the operator namespace (same API as hotrack_amd.pointnet2_utils).

    Exists for tests (CPU oracle on CPU tensors).  Nothing in the product selects a backend
    automatically: the default is the HIP one and it has no CPU path.
    """
    global _OPS
    _OPS = module


def hip_backend_active() -> bool:
    """True when the operator namespace is the product's HIP one (not a test-injected CPU oracle)."""
    return _OPS is None or getattr(_OPS, "__name__", "") == "hotrack_amd.pointnet2_utils"


def set_fused_backend(module) -> None:
    """Enable / disable (None) the eval-time fused set-abstraction kernels."""
    global _FUSED
    _FUSED = module


def fused_backend():
    return _FUSED


# ---------------------------------------------------------------------------------------
# free functions: the dispatch API of the reference (pointnet_utils.py:12-167)
# ---------------------------------------------------------------------------------------
def knn_point(k: int, pos2: torch.Tensor, pos1: torch.Tensor):
    """k nearest points of pos1 (B,N,3) for each query pos2 (B,M,3) -> (dist (B,M,k), idx int64)."""
    val, idx = _ops().knn(k, pos2, pos1)
    return val, idx.long()


def three_nn(xyz1: torch.Tensor, xyz2: torch.Tensor):
    """3 nearest points of xyz2 (B,S,3) for each xyz1 (B,N,3) -> (Euclidean dist, idx int64)."""
    dists, idx = _ops().three_nn(xyz1, xyz2)
    return dists, idx.long()


def three_interpolate(points: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """points (B,C,M), idx/weight (B,N,3) -> (B,C,N)."""
    return _ops().three_interpolate(points, idx.int(), weight)


def gather_operation(feature: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """(B,C,N), (B,S) -> (B,C,S)."""
    return _ops().gather_operation(feature.contiguous(), idx.int())


def group_operation(feature: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """(B,C,N), (B,S,K) -> (B,C,S,K)."""
    return _ops().grouping_operation(feature.contiguous(), idx)


def farthest_point_sample(xyz: torch.Tensor, npoint: int) -> torch.Tensor:
    """xyz (B,N,3) -> (B,npoint) int64; deterministic start at index 0 (the CUDA semantics)."""
    return _ops().furthest_point_sample(xyz, npoint).long()


def query_ball_point(radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
    """xyz (B,N,3), new_xyz (B,S,3) -> (B,S,nsample) int64."""
    return _ops().ball_query(radius, nsample, xyz, new_xyz).long()


def square_distance(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """(B,N,C),(B,M,C) -> (B,N,M) squared distances (plain torch helper, not on the hot path)."""
    return torch.cdist(src, dst).pow(2)


def index_points(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """points (B,N,C), idx (B,S...) -> (B,S...,C) (plain torch helper kept for API parity)."""
    B = points.shape[0]
    batch = torch.arange(B, device=points.device).view(B, *([1] * (idx.dim() - 1))).expand_as(idx)
    return points[batch, idx.long(), :]


def sample_and_group_all(xyz: torch.Tensor, points: Optional[torch.Tensor]):
    """xyz (B,N,3), points (B,N,D) -> origin centroid (B,1,3), (B,1,N,3+D) with xyz FIRST."""
    B, N, C = xyz.shape
    new_xyz = xyz.new_zeros(B, 1, C)
    grouped = xyz.view(B, 1, N, C)
    if points is not None:
        grouped = torch.cat([grouped, points.view(B, 1, N, -1)], dim=-1)
    return new_xyz, grouped


# ---------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------
def _make_mlp(in_channel: int, widths: List[int], dims: int):
    conv = nn.Conv2d if dims == 2 else nn.Conv1d
    bn = nn.BatchNorm2d if dims == 2 else nn.BatchNorm1d
    convs, bns = nn.ModuleList(), nn.ModuleList()
    last = in_channel
    for w in widths:
        convs.append(conv(last, w, 1))
        bns.append(bn(w))
        last = w
    return convs, bns, last


def _run_mlp(x: torch.Tensor, convs, bns) -> torch.Tensor:
    if _FUSED is not None and x.is_cuda and not torch.is_grad_enabled() and not bns[0].training:
        return _FUSED.mlp_stack(x, convs, bns)  # eval: BN folded, GEMM + one bias/ReLU pass
    for conv, bn in zip(convs, bns):
        x = F.relu(bn(conv(x)))  # training / unfused: the convolution library (measured faster than batched GEMMs here)
    return x


def _t(x: torch.Tensor) -> torch.Tensor:
    """(B,C,N) <-> (B,N,C), contiguous (the operators want point-major xyz)."""
    return x.transpose(1, 2).contiguous()


class _MultiScaleGroupedMLP(nn.Module):
    """conv_blocks[i][j] / bn_blocks[i][j]: one [Conv2d 1x1 + BN + ReLU]* stack per scale."""

    def __init__(self, in_channel: int, mlp_list: List[List[int]]):
        super().__init__()
        self.conv_blocks = nn.ModuleList()
        self.bn_blocks = nn.ModuleList()
        self.out_channel = 0
        for widths in mlp_list:
            convs, bns, last = _make_mlp(in_channel, widths, dims=2)
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)
            self.out_channel += last

    def _scale(self, i: int, xyz: torch.Tensor, points: Optional[torch.Tensor], new_xyz: torch.Tensor,
               group_idx: torch.Tensor, center_feat: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One scale: group -> [feat | xyz - centre | centre feat] -> MLP -> max over K.

        xyz (B,3,N), points (B,D,N) or None, new_xyz (B,3,S), group_idx (B,S,K) int32.
        Channel order = reference pointnet_utils.py:396 (SA-MSG) / :570,:575 (GivenCenterPoints).
        """
        fused = _FUSED
        if fused is not None and not self.training and not torch.is_grad_enabled() and xyz.is_cuda:
            pts = points if points is not None and points.shape[1] > 0 else None
            out = fused.sa_scale(self.conv_blocks[i], self.bn_blocks[i], xyz, pts, new_xyz, group_idx, center_feat)
            if out is not None:
                return out  # None: shape not covered by the fused kernel -> unfused operator path below
        ops = _ops()
        grouped_xyz = ops.grouping_operation(xyz, group_idx)
        grouped_xyz = grouped_xyz - new_xyz.unsqueeze(-1)
        if points is not None and points.shape[1] > 0:
            grouped = torch.cat([ops.grouping_operation(points, group_idx), grouped_xyz], dim=1)
        else:
            grouped = grouped_xyz  # zero feature channels: HandTrackNet's sa1 (use_xyz_feat=False)
        if center_feat is not None:
            K = grouped.shape[-1]
            grouped = torch.cat([grouped, center_feat.unsqueeze(-1).expand(-1, -1, -1, K)], dim=1)
        grouped = _run_mlp(grouped, self.conv_blocks[i], self.bn_blocks[i])
        return torch.max(grouped, -1)[0]


class PointNetSetAbstractionMsg(_MultiScaleGroupedMLP):
    """FPS -> gather -> per scale {ball query | kNN} -> grouped MLP -> max (reference :187-235)."""

    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list, knn=False):
        super().__init__(in_channel, mlp_list)
        self.npoint = npoint
        self.radius_list = radius_list
        self.nsample_list = nsample_list
        self.knn = knn

    def forward(self, xyz: torch.Tensor, points: Optional[torch.Tensor]):
        """xyz (B,3,N), points (B,D,N)|None -> new_xyz (B,3,S), new_points (B,D',S)."""
        ops = _ops()
        xyz = xyz.contiguous()
        xyz_t = _t(xyz)
        fps_idx = ops.furthest_point_sample(xyz_t, self.npoint)  # (B,S) int32
        new_xyz = ops.gather_operation(xyz, fps_idx)  # (B,3,S)
        new_xyz_t = _t(new_xyz)
        if points is not None:
            points = points.contiguous()
        outs = []
        for i, radius in enumerate(self.radius_list):
            K = self.nsample_list[i]
            if self.knn:
                _, group_idx = ops.knn(K, new_xyz_t, xyz_t)
            else:
                group_idx = ops.ball_query(radius, K, xyz_t, new_xyz_t)
            outs.append(self._scale(i, xyz, points, new_xyz, group_idx))
        return new_xyz, torch.cat(outs, dim=1)


class PointNetFeaturePropagation(nn.Module):
    """three-NN inverse-distance interpolation + [Conv1d + BN + ReLU]* (reference :238-285)."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs, self.mlp_bns, self.out_channel = _make_mlp(in_channel, mlp, dims=1)

    def forward(self, xyz1, xyz2, points1, points2):
        """xyz1 (B,3,N), xyz2 (B,3,S), points1 (B,D,N)|None, points2 (B,D2,S) -> (B,D',N)."""
        ops = _ops()
        B, _, N = xyz1.shape
        S = xyz2.shape[2]
        if S == 1:
            interpolated = points2.expand(-1, -1, N)
        else:
            dist, idx = ops.three_nn(_t(xyz1), _t(xyz2))
            recip = 1.0 / (dist + 1e-8)
            weight = recip / recip.sum(dim=2, keepdim=True)
            interpolated = ops.three_interpolate(points2.contiguous(), idx, weight)
        x = interpolated if points1 is None else torch.cat([points1, interpolated], dim=1)
        return _run_mlp(x, self.mlp_convs, self.mlp_bns)


class PointNetSetAbstraction(nn.Module):
    """Group-all SA layer: [xyz | feat] of every point -> MLP -> max over N (reference :288-343)."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all, knn=False):
        super().__init__()
        if not group_all:
            raise NotImplementedError("only group_all=True exists in the reference (pointnet_utils.py:330)")
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.mlp_convs, self.mlp_bns, self.out_channel = _make_mlp(in_channel, mlp, dims=2)
        self.group_all = group_all
        self.knn = knn

    def forward(self, xyz, points):
        """xyz (B,3,N), points (B,D,N)|None -> new_xyz (B,3,1) zeros, new_points (B,D',1)."""
        B, C, N = xyz.shape
        x = xyz if points is None else torch.cat([xyz, points], dim=1)  # xyz FIRST, centre = origin
        x = _run_mlp(x.unsqueeze(-1), self.mlp_convs, self.mlp_bns)  # (B,D',N,1)
        return xyz.new_zeros(B, C, 1), torch.max(x, 2)[0]


class PointNetSetAbstractionMsg_GivenCenterPoints(_MultiScaleGroupedMLP):
    """SA-MSG around GIVEN centres (the 21 hand keypoints), kNN or ball grouping (reference :515-590)."""

    def __init__(self, radius_list, nsample_list, mlp_list, in_channel, knn=False):
        super().__init__(in_channel, mlp_list)
        self.radius_list = radius_list
        self.nsample_list = nsample_list
        self.knn = knn

    def forward(self, xyz, points, new_xyz, new_points, return_4nn=False, pre_group_idx=None,
                return_group_idx=False):
        """xyz (B,3,N), points (B,D,N), new_xyz (B,3,S), new_points (B,D2,S)|None -> (B,D',S)."""
        ops = _ops()
        xyz = xyz.contiguous()
        new_xyz = new_xyz.contiguous()
        points = None if points is None else points.contiguous()
        xyz_t = new_xyz_t = None
        outs, idx_list = [], []
        for i, radius in enumerate(self.radius_list):
            K = self.nsample_list[i]
            if pre_group_idx is not None:
                group_idx = pre_group_idx[i]
                if group_idx.dtype != torch.int32:
                    group_idx = group_idx.int()
            else:
                if xyz_t is None:
                    xyz_t, new_xyz_t = _t(xyz), _t(new_xyz)
                if self.knn:
                    _, group_idx = ops.knn(K, new_xyz_t, xyz_t)
                else:
                    group_idx = ops.ball_query(radius, K, xyz_t, new_xyz_t)
            idx_list.append(group_idx)
            outs.append(self._scale(i, xyz, points, new_xyz, group_idx, center_feat=new_points))
        out = torch.cat(outs, dim=1)
        if return_4nn:
            g = ops.grouping_operation(xyz, idx_list[-1][..., :4].contiguous()) - new_xyz.unsqueeze(-1)
            return out, g.norm(dim=1, keepdim=True).mean(dim=-1)
        if return_group_idx:
            return out, idx_list
        return out


# ---------------------------------------------------------------------------------------
# "_fast" variants: the reference's (B, P, C, N) interface (P parts share one point set;
# indices are computed once on part 0 and the features are folded into the batch).
# ---------------------------------------------------------------------------------------
class PointNetSetAbstractionMsg_fast(PointNetSetAbstractionMsg):
    def forward(self, xyz: torch.Tensor, points: Optional[torch.Tensor]):
        """xyz (B,P,3,N), points (B,P,D,N)|None -> (B,P,3,S), (B,P,D',S) (reference :346-409)."""
        B, P, C, N = xyz.shape
        if P == 1:
            pts = None if points is None or points.shape[2] == 0 else points[:, 0]
            new_xyz, new_points = super().forward(xyz[:, 0], pts)
            return new_xyz.unsqueeze(1), new_points.unsqueeze(1)
        ops = _ops()
        xyz0 = xyz[:, 0].contiguous()
        xyz_t = _t(xyz0)
        fps_idx = ops.furthest_point_sample(xyz_t, self.npoint)
        new_xyz = ops.gather_operation(xyz0, fps_idx)
        new_xyz_t = _t(new_xyz)
        rep = lambda t: t.unsqueeze(1).expand(-1, P, *t.shape[1:]).reshape(B * P, *t.shape[1:]).contiguous()
        pts = None if points is None or points.shape[2] == 0 else points.reshape(B * P, -1, N).contiguous()
        outs = []
        for i, radius in enumerate(self.radius_list):
            K = self.nsample_list[i]
            if self.knn:
                _, group_idx = ops.knn(K, new_xyz_t, xyz_t)
            else:
                group_idx = ops.ball_query(radius, K, xyz_t, new_xyz_t)
            outs.append(self._scale(i, rep(xyz0), pts, rep(new_xyz), rep(group_idx)))
        S = self.npoint
        return new_xyz.unsqueeze(1).expand(-1, P, -1, -1), torch.cat(outs, dim=1).reshape(B, P, -1, S)


class PointNetFeaturePropagation_fast(PointNetFeaturePropagation):
    def forward(self, xyz1, xyz2, points1, points2):
        """(B,P,3,N), (B,P,3,S), (B,P,D,N), (B,P,D2,S) -> (B,P,D',N) (reference :412-464)."""
        B, P, _, N = xyz1.shape
        S = xyz2.shape[-1]
        p1 = points1.reshape(B * P, -1, N)
        p2 = points2.reshape(B * P, -1, S)
        if P == 1:
            return super().forward(xyz1[:, 0], xyz2[:, 0], p1, p2).unsqueeze(1)
        ops = _ops()
        if S == 1:
            interpolated = p2.expand(-1, -1, N)
        else:
            dist, idx = ops.three_nn(_t(xyz1[:, 0]), _t(xyz2[:, 0]))
            recip = 1.0 / (dist + 1e-8)
            weight = recip / recip.sum(dim=2, keepdim=True)
            rep = lambda t: t.unsqueeze(1).expand(-1, P, -1, -1).reshape(B * P, N, 3).contiguous()
            interpolated = ops.three_interpolate(p2.contiguous(), rep(idx), rep(weight))
        x = torch.cat([p1, interpolated], dim=1)
        return _run_mlp(x, self.mlp_convs, self.mlp_bns).reshape(B, P, -1, N)


class PointNetSetAbstraction_fast(PointNetSetAbstraction):
    def forward(self, xyz, points):
        """(B,P,3,N), (B,P,D,N) -> (B,P,3,1), (B,P,D',1) (reference :467-512)."""
        B, P, C, N = xyz.shape
        new_xyz, new_points = super().forward(xyz.reshape(B * P, C, N), points.reshape(B * P, -1, N))
        return new_xyz.reshape(B, P, C, 1), new_points.reshape(B, P, -1, 1)
