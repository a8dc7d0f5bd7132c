"""Rotation helpers of the particle optimisers (counterparts of the reference's pose_utils/rotations.py:6-9, :105-132,
:144-152, :328-369 and network/models/hand_utils.py:13-19), restated with the same arithmetic and the same epsilons so
that the hand-pose optimiser reproduces the reference's numbers; everything stays on the input's device."""
from __future__ import annotations

import torch

EPS_Q = 1e-8  # pose_utils/rotations.py:5


def normalize_quaternion(q: torch.Tensor) -> torch.Tensor:
    return q.div(q.norm(dim=-1, keepdim=True) + EPS_Q)


def unit_quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """(…,4) (w,x,y,z) -> (…,3,3), pose_utils/rotations.py:105-113 (no normalisation: the caller's job)."""
    w, x, y, z = torch.unbind(q, dim=-1)
    m = torch.stack((1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w,
                     2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w,
                     2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y), dim=-1)
    return m.view(list(m.shape[:-1]) + [3, 3]).contiguous()


def matrix_to_unit_quaternion(matrix: torch.Tensor) -> torch.Tensor:
    """(…,3,3) -> (…,4), pose_utils/rotations.py:116-132 (trace form, w >= 0)."""
    trace = torch.clamp(1 + matrix[..., 0, 0] + matrix[..., 1, 1] + matrix[..., 2, 2], min=0.)
    r = torch.sqrt(trace)
    s = 1.0 / (2 * r + 1e-7)
    q = torch.stack((0.5 * r, (matrix[..., 2, 1] - matrix[..., 1, 2]) * s, (matrix[..., 0, 2] - matrix[..., 2, 0]) * s,
                     (matrix[..., 1, 0] - matrix[..., 0, 1]) * s), dim=-1)
    return normalize_quaternion(q)


def quaternion_to_axis_angle(quat: torch.Tensor) -> torch.Tensor:
    """(B, 4) -> (B, 3) axis * angle: hand_utils.mano_quat2axisang on one joint (:13-19) over
    rotations.quater_to_axis_theta (:144-152)."""
    q = normalize_quaternion(quat)
    cosa = q[..., 0]
    norm = torch.sqrt(1 - cosa ** 2).unsqueeze(-1)
    axis = q[..., 1:] / torch.max(norm, (norm < 1e-8).float())
    theta = 2 * torch.acos(torch.clamp(cosa, min=-1, max=1))
    return axis * theta[:, None]


def _normalize_vector(v: torch.Tensor) -> torch.Tensor:  # pose_utils/rotations.py:328-340
    mag = torch.norm(v, p=2, dim=1)
    eps = torch.full((1,), 1e-8, dtype=v.dtype, device=v.device)
    valid = (mag > eps).to(v.dtype).view(-1, 1)
    backup = torch.tensor([1.0, 0.0, 0.0], dtype=v.dtype, device=v.device).view(1, 3).expand(v.shape[0], 3)
    v = v / torch.max(mag, eps).view(-1, 1)
    return v * valid + backup * (1 - valid)


def _cross(u: torch.Tensor, v: torch.Tensor) -> torch.Tensor:  # :343-353
    return torch.stack((u[:, 1] * v[:, 2] - u[:, 2] * v[:, 1], u[:, 2] * v[:, 0] - u[:, 0] * v[:, 2],
                        u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]), dim=1)


def rotation_from_ortho6d(poses: torch.Tensor) -> torch.Tensor:
    """(B, 6) -> (B, 3, 3) with columns x, y, z (Gram-Schmidt), pose_utils/rotations.py:356-369."""
    x = _normalize_vector(poses[:, 0:3])
    z = _normalize_vector(_cross(x, poses[:, 3:6]))
    y = _cross(z, x)
    return torch.stack((x, y, z), dim=2)
