"""The reference's CPU path, restated -- BASELINE TIMING / TEST INFRASTRUCTURE ONLY.

When no GPU is present the reference runs pure-PyTorch fallbacks for every point operator
(network/models/pointnet_utils.py, `CUDA = torch.cuda.is_available()` False branch).  That
is "the reference's CPU path" of BASELINE.json configs[0].  The reference's Python cannot
travel to the GPU box, so bench.py's cpu_baseline leg times THIS restatement of the same
algorithms (kind = "port"), driving our HandTrackNet with elide_dead_attention=False (the
reference computes the discarded attention too).  Each function cites the lines it follows.
It was checked in the build container against the imported reference (tests/test_cpu_reference.py).

Same API as hotrack_amd.pointnet2_utils so it plugs into pointnet_utils.set_operator_backend.
Semantics are the FALLBACK's, not the CUDA kernels' (SURVEY.md 8(c)): three_nn returns squared
distances, ball query uses `>` and float(r**2), FPS start is configurable (reference: random).
"""
from __future__ import annotations

import torch

FPS_START = 0  # reference draws torch.randint (pointnet_utils.py:128); fixed here for repeatable timing


def _sqdist_matmul(src, dst):
    """pointnet_utils.py:56-77: -2 src.dst^T + |src|^2 + |dst|^2."""
    d = -2 * torch.matmul(src, dst.transpose(1, 2))
    d += (src ** 2).sum(-1)[:, :, None]
    d += (dst ** 2).sum(-1)[:, None, :]
    return d


def _index(points, idx):
    """pointnet_utils.py:80-97: batched advanced indexing, points (B,N,C), idx (B,...) -> (B,...,C)."""
    B = points.shape[0]
    shape = [B] + [1] * (idx.dim() - 1)
    batch = torch.arange(B).view(shape).expand_as(idx)
    return points[batch, idx, :]


def furthest_point_sample(xyz, npoint):
    """pointnet_utils.py:126-137: Python loop of npoint torch ops."""
    B, N, _ = xyz.shape
    centroids = torch.zeros(B, npoint, dtype=torch.long)
    distance = torch.ones(B, N) * 1e10
    farthest = torch.full((B,), FPS_START, dtype=torch.long)
    batch = torch.arange(B)
    for i in range(npoint):
        centroids[:, i] = farthest
        c = xyz[batch, farthest, :].view(B, 1, 3)
        dist = ((xyz - c) ** 2).sum(-1)
        mask = dist < distance
        distance[mask] = dist[mask]
        farthest = distance.max(-1)[1]
    return centroids.int()


def ball_query(radius, nsample, xyz, new_xyz):
    """pointnet_utils.py:156-167: full distance matrix + sort over N."""
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    group = torch.arange(N).view(1, 1, N).repeat(B, S, 1)
    group[_sqdist_matmul(new_xyz, xyz) > radius ** 2] = N
    group = group.sort(dim=-1)[0][:, :, :nsample]
    first = group[:, :, 0].view(B, S, 1).repeat(1, 1, nsample)
    first[first == N] = 0
    mask = group == N
    group[mask] = first[mask]
    return group.int()


def knn(k, unknown, known):
    """pointnet_utils.py:26-32: materialise (B,M,N,3) with repeat, then topk."""
    B, N, _ = known.shape
    M = unknown.shape[1]
    p1 = known.view(B, 1, N, -1).repeat(1, M, 1, 1)
    p2 = unknown.view(B, M, 1, -1).repeat(1, 1, N, 1)
    val, idx = (-(p1 - p2) ** 2).sum(-1).topk(k=k, dim=-1)
    return torch.sqrt(-val), idx.int()


def three_nn(unknown, known):
    """pointnet_utils.py:40-43: full sort; returns SQUARED distances (unlike the CUDA path)."""
    d, idx = _sqdist_matmul(unknown, known).sort(dim=-1)
    return d[:, :, :3], idx[:, :, :3].int()


def three_interpolate(points, idx, weight):
    """pointnet_utils.py:50-53."""
    B, N = idx.shape[:2]
    g = _index(points.permute(0, 2, 1), idx.long())  # (B,N,3,C)
    return (g * weight.view(B, N, 3, 1)).sum(dim=2).permute(0, 2, 1)


def gather_operation(features, idx):
    """pointnet_utils.py:100-103."""
    return _index(features.transpose(-1, -2), idx.long()).transpose(-1, -2)


def grouping_operation(features, idx):
    """pointnet_utils.py:106-109."""
    return _index(features.transpose(-1, -2), idx.long()).permute(0, 3, 1, 2)
