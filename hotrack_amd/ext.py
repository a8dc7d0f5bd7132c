"""Python binding of the MI355X-side extensions (include/pn2_ext.h). GPU tensors only."""
from __future__ import annotations

import ctypes

import torch

from . import pointnet2_hip as _native

_lib = _native._lib
_vp, _ci = ctypes.c_void_p, ctypes.c_int
_lib.pn2x_kabsch.argtypes = [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp]
_lib.pn2x_kabsch.restype = _ci


def kabsch(x: torch.Tensor, y: torch.Tensor):
    """Rigid fit y ~= R x + t.  x (B|1, num, 3), y (B, num, 3) -> R (B,3,3), t (B,3,1)."""
    x = x.float().contiguous()
    y = y.float().contiguous()
    if x.dim() == 2:
        x = x.unsqueeze(0)
    B, num, _ = y.shape
    xb = x.shape[0]
    if xb not in (1, B) or x.shape[1] != num:
        raise ValueError(f"kabsch: x {tuple(x.shape)} does not match y {tuple(y.shape)}")
    px = _native._ptr(x, "x", torch.float32, xb * num * 3)
    py = _native._ptr(y, "y", torch.float32, B * num * 3)
    R = torch.empty((B, 3, 3), dtype=torch.float32, device=y.device)
    t = torch.empty((B, 3, 1), dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        _native._check(_lib.pn2x_kabsch(B, xb, num, px, py, R.data_ptr(), t.data_ptr(), _native._stream(y)), "kabsch")
    return R, t
