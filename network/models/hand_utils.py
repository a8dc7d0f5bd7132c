"""Hand-frame helpers of HandTrackNet (counterpart of the reference's hand_utils.py:30-124)."""
from __future__ import annotations

import torch


def canonicalize(data: torch.Tensor, canon_pose: dict) -> torch.Tensor:
    """data (B,3,N) camera frame -> hand frame: R^T (data - t) / scale."""
    return torch.matmul(canon_pose["rotation"].transpose(-1, -2), data - canon_pose["translation"]) \
        / canon_pose["scale"][:, None, None]


def decanonicalize(data: torch.Tensor, canon_pose: dict) -> torch.Tensor:
    """hand frame (B,3,N) -> camera frame: scale * R data + t."""
    return canon_pose["scale"][:, None, None] * torch.matmul(canon_pose["rotation"], data) + canon_pose["translation"]


def solve_rot_and_trans(x: torch.Tensor, y: torch.Tensor, cpu: bool = True):
    """Least-squares R (B,3,3), t (B,3,1) with y ~= R x + t for x, y (B,num,3).

    The reference moves the 3x3 cross-covariance to the CPU for torch.svd on every forward
    (hand_utils.py:55-60).  GPU tensors are solved by the device kernel pn2x_kabsch instead
    (no host round trip; `cpu` is accepted for signature parity and ignored); CPU tensors, and
    fits that must be differentiated (training losses), use the SVD form of the same algorithm
    on the tensor's own device.
    """
    needs_grad = torch.is_grad_enabled() and (y.requires_grad or x.requires_grad)
    if y.is_cuda and not needs_grad:
        from hotrack_amd import ext
        return ext.kabsch(x.to(y.device), y)
    # differentiable form (training losses back-propagate through the fit), on y's own device
    if x.dim() == 2:
        x = x.unsqueeze(0)
    x = x.expand(y.shape[0], -1, -1).to(y.dtype)
    cx, cy = x.mean(dim=1, keepdim=True), y.mean(dim=1, keepdim=True)
    w = torch.bmm((x - cx).transpose(-1, -2), y - cy)
    u, _, vh = torch.linalg.svd(w)
    v = vh.transpose(-1, -2)
    d = torch.det(torch.bmm(v, u.transpose(-1, -2)))
    fix = torch.eye(3, dtype=y.dtype, device=y.device).repeat(y.shape[0], 1, 1)
    fix[:, 2, 2] = d
    R = torch.bmm(torch.bmm(v, fix), u.transpose(-1, -2))
    t = cy - torch.bmm(cx, R.transpose(-1, -2))
    return R, t.transpose(-1, -2)


def ransac_rt(x, y, n=0, cpu=True):
    """n == 0 (the only mode HandTrackNet uses): plain Kabsch over all points.
    Returns (R, t, None, None, None) like the reference (hand_utils.py:68-81)."""
    if n != 0:
        raise NotImplementedError("only n=0 is used by HandTrackNet (hand_network.py:100)")
    R, t = solve_rot_and_trans(x, y, cpu)
    return R, t, None, None, None


_PALM_21 = [0, 1, 5, 9, 13, 17]
_PALM_29 = [0, 1, 5, 6, 7, 11, 12, 13, 17, 18, 19, 23, 24, 25]
_IDX_CACHE = {}


def _palm_index(kind: int, device) -> torch.Tensor:
    """Device-resident index tensor, created once per device (a Python-list index would issue a
    host-to-device copy on every call, which also cannot be captured into a HIP graph)."""
    key = (kind, str(device))
    idx = _IDX_CACHE.get(key)
    if idx is None:
        idx = torch.tensor(_PALM_21 if kind == 21 else _PALM_29, dtype=torch.long, device=device)
        _IDX_CACHE[key] = idx
    return idx


def handkp2palmkp(kp: torch.Tensor) -> torch.Tensor:
    """(B, 21|29, 3) hand keypoints -> the rigid palm subset (B, 6|14, 3)."""
    if kp.shape[1] not in (21, 29):
        raise NotImplementedError(f"unsupported keypoint count {kp.shape[1]}")
    return kp.index_select(1, _palm_index(kp.shape[1], kp.device))
