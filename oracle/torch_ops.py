"""CPU-tensor operator namespace backed by the C oracle -- TEST INFRASTRUCTURE ONLY.

Mirrors the public API of hotrack_amd.pointnet2_utils (and of the reference's
pointnet2_utils.py) on CPU tensors so that tests can (a) compare the HIP operators against
it and (b) drive the network counterparts on CPU (state-dict / gloo-DDP tests, golden-vector
generation against the imported reference).  Never imported by hotrack_amd/ or network/.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.autograd import Function

from . import pn2_oracle as O


def _np(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().contiguous().numpy()


class _FPS(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        out = torch.from_numpy(O.furthest_point_sample(_np(xyz.float()), int(npoint)))
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, a=None):
        return None, None


class _Gather(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.for_backwards = (idx, features.shape[2])
        return torch.from_numpy(O.gather_points(_np(features), _np(idx.int())))

    @staticmethod
    def backward(ctx, grad_out):
        idx, N = ctx.for_backwards
        return torch.from_numpy(O.gather_points_grad(_np(grad_out), _np(idx.int()), N)), None


class _KNN(Function):
    @staticmethod
    def forward(ctx, k, unknown, known):
        d2, i = O.knn(int(k), _np(unknown.float()), _np(known.float()))
        d, i = torch.sqrt(torch.from_numpy(d2)), torch.from_numpy(i)
        ctx.mark_non_differentiable(d, i)
        return d, i

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None


class _ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        d2, i = O.three_nn(_np(unknown.float()), _np(known.float()))
        d, i = torch.sqrt(torch.from_numpy(d2)), torch.from_numpy(i)
        ctx.mark_non_differentiable(d, i)
        return d, i

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


class _Interp(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        ctx.saved = (idx, weight, features.shape[2])
        return torch.from_numpy(O.three_interpolate(_np(features), _np(idx.int()), _np(weight)))

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, m = ctx.saved
        return torch.from_numpy(O.three_interpolate_grad(_np(grad_out), _np(idx.int()), _np(weight), m)), None, None


class _Group(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.for_backwards = (idx, features.shape[2])
        return torch.from_numpy(O.group_points(_np(features), _np(idx.int())))

    @staticmethod
    def backward(ctx, grad_out):
        idx, N = ctx.for_backwards
        return torch.from_numpy(O.group_points_grad(_np(grad_out), _np(idx.int()), N)), None


class _Ball(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        out = torch.from_numpy(O.ball_query(float(radius), int(nsample), _np(xyz.float()), _np(new_xyz.float())))
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


furthest_point_sample = _FPS.apply
gather_operation = _Gather.apply
knn = _KNN.apply
three_nn = _ThreeNN.apply
three_interpolate = _Interp.apply
grouping_operation = _Group.apply
ball_query = _Ball.apply
