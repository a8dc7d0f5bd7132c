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
    if y.is_cuda and y.dtype == torch.float32 and not x.requires_grad:
        # value and gradient (with respect to y) from the device kernels, in closed form: no solver-library SVD in the
        # training step, one launch per direction (the element-wise form below is ~100 launches on (B,3,3) tensors)
        from hotrack_amd import ext
        return ext.KabschFit.apply(x.to(y.device), y)
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


def _hat(u):
    """(B,3) -> (B,3,3) skew matrices with hat(u) v = u x v."""
    z = torch.zeros_like(u[:, 0])
    return torch.stack([torch.stack([z, -u[:, 2], u[:, 1]], -1), torch.stack([u[:, 2], z, -u[:, 0]], -1),
                        torch.stack([-u[:, 1], u[:, 0], z], -1)], -2)


def _inv3(K):
    """Closed-form inverse of (B,3,3) matrices (adjugate / determinant): element-wise ops only, so it can be captured
    into a HIP graph (torch.linalg.inv / solve / svd call the solver library, which cannot)."""
    a, b, c = K[:, 0, 0], K[:, 0, 1], K[:, 0, 2]
    d, e, f = K[:, 1, 0], K[:, 1, 1], K[:, 1, 2]
    g, h, i = K[:, 2, 0], K[:, 2, 1], K[:, 2, 2]
    A, Bc, C = e * i - f * h, c * h - b * i, b * f - c * e
    D, E, Fc = f * g - d * i, a * i - c * g, c * d - a * f
    G, H, I = d * h - e * g, b * g - a * h, a * e - b * d
    det = a * A + b * D + c * G
    adj = torch.stack([torch.stack([A, Bc, C], -1), torch.stack([D, E, Fc], -1), torch.stack([G, H, I], -1)], -2)
    return adj / det[:, None, None]


class _KabschRotation(torch.autograd.Function):
    """R(w) of the Kabsch fit as a differentiable function of the 3x3 cross-covariance w = (x-cx)^T (y-cy), with the
    value supplied by the caller and the gradient in closed form.  This is the element-wise statement of the formula
    (checked against autograd through an fp64 SVD on the CPU, tests/test_network.py); on the GPU the training losses use
    hotrack_amd.ext.KabschFit, the same derivative as one kernel per direction (pn2x_kabsch / pn2x_kabsch_backward).

    R = V diag(1,1,d) U^T (w = U S V^T) makes  R w = Sym  symmetric, i.e. w = Q Sym with Q = R^T the polar factor of
    w.  Differentiating  w = Q Sym :  Q^T dw - dw^T Q = X Sym + Sym X  with X = Q^T dQ skew; in axial vectors
    x = K^-1 z,  K = tr(Sym) I - Sym,  z = axial(Q^T dw - dw^T Q).  For a loss gradient G = dL/dR this gives
        dL/dw = 2 Q hat(u),   u = K^-1 axial((B - B^T)/2),   B = Q^T G^T = R G^T
    -- the same derivative autograd takes through torch.linalg.svd (tests/test_network.py checks it in fp64), without
    the solver library in either direction: no host round trip, capturable into a HIP graph.
    """

    @staticmethod
    def forward(ctx, w, R):
        ctx.save_for_backward(w, R)
        return R.clone()

    @staticmethod
    def backward(ctx, G):
        w, R = ctx.saved_tensors
        sym = torch.bmm(R, w)
        sym = 0.5 * (sym + sym.transpose(-1, -2))
        tr = sym[:, 0, 0] + sym[:, 1, 1] + sym[:, 2, 2]
        K = tr[:, None, None] * torch.eye(3, dtype=w.dtype, device=w.device) - sym
        Bm = torch.bmm(R, G.transpose(-1, -2))
        P = 0.5 * (Bm - Bm.transpose(-1, -2))
        p = torch.stack([P[:, 2, 1], P[:, 0, 2], P[:, 1, 0]], -1)
        u = torch.bmm(_inv3(K), p.unsqueeze(-1)).squeeze(-1)
        return 2.0 * torch.bmm(R.transpose(-1, -2), _hat(u)), None


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
