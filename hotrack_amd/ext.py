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


_lib.pn2x_sa_mlp_max.argtypes = [_ci] * 7 + [_vp] * 8 + [_vp]
_lib.pn2x_sa_mlp_max.restype = _ci
_lib.pn2x_sa_mlp_max_supported.argtypes = [_ci] * 4
_lib.pn2x_sa_mlp_max_supported.restype = _ci


def sa_mlp_max_supported(k: int, c1: int, c2: int, c3: int) -> bool:
    return bool(_lib.pn2x_sa_mlp_max_supported(k, c1, c2, c3))


def sa_mlp_max(a1: torch.Tensor, c1v: torch.Tensor, idx: torch.Tensor, w2, b2, w3, b3) -> torch.Tensor:
    """a1 (B,N,C1) per-point layer-1 term, c1v (B,S,C1) per-centroid term, idx (B,S,K) int32,
    w2 (C2,C1), w3 (C3,C2) BN-folded -> (B,C3,S) = relu(max_k W3 relu(W2 relu(a1[idx]+c1v)+b2)+b3)."""
    B, N, C1 = a1.shape
    _, S, K = idx.shape
    C2, C3 = w2.shape[0], w3.shape[0]
    f32, i32 = torch.float32, torch.int32
    pa = _native._ptr(a1, "a1", f32, B * N * C1)
    pc = _native._ptr(c1v, "c1v", f32, B * S * C1)
    pi = _native._ptr(idx, "idx", i32, B * S * K)
    pw2 = _native._ptr(w2, "w2", f32, C2 * C1)
    pb2 = _native._ptr(b2, "b2", f32, C2)
    pw3 = _native._ptr(w3, "w3", f32, C3 * C2)
    pb3 = _native._ptr(b3, "b3", f32, C3)
    out = torch.empty((B, C3, S), dtype=f32, device=a1.device)
    with torch.cuda.device(a1.device):
        _native._check(_lib.pn2x_sa_mlp_max(B, N, S, K, C1, C2, C3, pa, pc, pi, pw2, pb2, pw3, pb3, out.data_ptr(),
                                            _native._stream(a1)), "sa_mlp_max")
    return out


_lib.pn2x_bias_act.argtypes = [_ci, _ci, _ci, _vp, _vp, _ci, _vp]
_lib.pn2x_bias_act.restype = _ci


def bias_act_(y: torch.Tensor, bias: torch.Tensor, relu: bool = True) -> torch.Tensor:
    """In place y[b,c,n] = act(y + bias[c]); y (B,C,N) contiguous fp32."""
    B, C, N = y.shape
    py = _native._ptr(y, "y", torch.float32, B * C * N)
    pb = _native._ptr(bias, "bias", torch.float32, C)
    with torch.cuda.device(y.device):
        _native._check(_lib.pn2x_bias_act(B, C, N, py, pb, 1 if relu else 0, _native._stream(y)), "bias_act")
    return y
