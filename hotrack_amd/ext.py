"""Python binding of the MI355X-side extensions (include/pn2_ext.h). GPU tensors only."""
from __future__ import annotations

import ctypes
import os

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


_lib.pn2x_kabsch_backward.argtypes = [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
_lib.pn2x_kabsch_backward.restype = _ci


def kabsch_backward(x: torch.Tensor, y: torch.Tensor, R: torch.Tensor, grad_R, grad_t) -> torch.Tensor:
    """dL/dy (B,num,3) through (R, t) = kabsch(x, y) given dL/dR (B,3,3) and dL/dt (B,3,1) (either may be None)."""
    x = x.float().contiguous()
    y = y.float().contiguous()
    if x.dim() == 2:
        x = x.unsqueeze(0)
    B, num, _ = y.shape
    f32 = torch.float32
    gR = None if grad_R is None else _native._ptr(grad_R.contiguous(), "grad_R", f32, B * 9)
    gt = None if grad_t is None else _native._ptr(grad_t.contiguous(), "grad_t", f32, B * 3)
    dy = torch.empty_like(y)
    with torch.cuda.device(y.device):
        _native._check(_lib.pn2x_kabsch_backward(B, x.shape[0], num, _native._ptr(x, "x", f32, x.shape[0] * num * 3),
                                                 _native._ptr(y, "y", f32, B * num * 3), _native._ptr(R.contiguous(), "R", f32, B * 9),
                                                 gR, gt, dy.data_ptr(), _native._stream(y)), "kabsch_backward")
    return dy


class KabschFit(torch.autograd.Function):
    """(R, t) = kabsch(x, y), differentiable with respect to y: one launch forward (pn2x_kabsch), one backward."""

    @staticmethod
    def forward(ctx, x, y):
        R, t = kabsch(x, y)
        ctx.save_for_backward(x, y, R)
        return R, t

    @staticmethod
    def backward(ctx, grad_R, grad_t):
        x, y, R = ctx.saved_tensors
        return None, kabsch_backward(x, y, R, grad_R, grad_t)


_cl = ctypes.c_long
_lib.pn2x_sa_mlp_max.argtypes = [_ci] * 7 + [_vp, _ci, _vp, _vp, _vp, _vp, _vp, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _cl, _ci, _ci, _vp]
_lib.pn2x_sa_mlp_max.restype = _ci
_lib.pn2x_sa_mlp_max_supported.argtypes = [_ci] * 4
_lib.pn2x_sa_mlp_max_supported.restype = _ci
_lib.pn2x_three_nn_weights.argtypes = [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp]
_lib.pn2x_three_nn_weights.restype = _ci
_lib.pn2x_three_interpolate_pm.argtypes = [_ci, _ci, _ci, _ci, _vp, _ci, _vp, _vp, _vp, _ci, _vp]
_lib.pn2x_three_interpolate_pm.restype = _ci
_lib.pn2x_gather_rows.argtypes = [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp]
_lib.pn2x_gather_rows.restype = _ci
_lib.pn2x_bias_act_pm.argtypes = [_cl, _ci, _vp, _ci, _vp, _cl, _ci, _vp]
_lib.pn2x_bias_act_pm.restype = _ci


def sa_mlp_max_supported(k: int, c1: int, c2: int, c3: int) -> bool:
    return bool(_lib.pn2x_sa_mlp_max_supported(k, c1, c2, c3))


def _rows(t: torch.Tensor, name: str, cols: int):
    """Validate a point-major (B, R, >=cols) fp32 tensor whose rows may be a column block of a wider
    buffer (last dim contiguous, uniform row stride).  Returns (data_ptr, row_stride)."""
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 3:
        raise TypeError(f"{name}: expected a 3-D float32 GPU tensor")
    B, R, C = t.shape
    if C < cols or t.stride(2) != 1 or (R > 1 and B > 1 and t.stride(0) != R * t.stride(1)):
        raise ValueError(f"{name}: unsupported layout shape={tuple(t.shape)} strides={t.stride()}")
    return t.data_ptr(), t.stride(1)


def sa_mlp_max(idx: torch.Tensor, w2, b2, w3, b3, *, a1f=None, xyz=None, cxyz=None, wx=None, b1=None, cadd=None,
               out=None, point_major=False) -> torch.Tensor:
    """Fused SA scale (include/pn2_ext.h: pn2x_sa_mlp_max).
    idx (B,S,K) int32; a1f (B,N,>=C1) rows; xyz (B,N,3); cxyz (B,S,3); wx (C1,3); b1 (C1); cadd (B,S,>=C1).
    Returns (B,C3,S), or (B,S,C3) if point_major; `out` may be a (B,S,C3) column block of a wider buffer."""
    B, S, K = idx.shape
    C1, C2, C3 = w2.shape[1], w2.shape[0], w3.shape[0]
    f32 = torch.float32
    N = a1f.shape[1] if a1f is not None else xyz.shape[1]
    pa, lda = (None, 0) if a1f is None else _rows(a1f, "a1f", C1)
    pc, ldc = (None, 0) if cadd is None else _rows(cadd, "cadd", C1)
    px = None if xyz is None else _native._ptr(xyz, "xyz", f32, B * N * 3)
    pcx = None if cxyz is None else _native._ptr(cxyz, "cxyz", f32, B * S * 3)
    pwx = None if wx is None else _native._ptr(wx, "wx", f32, C1 * 3)
    pb1 = None if b1 is None else _native._ptr(b1, "b1", f32, C1)
    pi = _native._ptr(idx, "idx", torch.int32, B * S * K)
    pw2, pb2 = _native._ptr(w2, "w2", f32, C2 * C1), _native._ptr(b2, "b2", f32, C2)
    pw3, pb3 = _native._ptr(w3, "w3", f32, C3 * C2), _native._ptr(b3, "b3", f32, C3)
    if out is None:
        out = torch.empty((B, S, C3) if point_major else (B, C3, S), dtype=f32, device=idx.device)
        ob, os_, oc = (S * C3, C3, 1) if point_major else (C3 * S, 1, S)
        po = out.data_ptr()
    else:
        po, ld = _rows(out, "out", C3)
        ob, os_, oc = out.stride(0), ld, 1
    with torch.cuda.device(idx.device):
        _native._check(_native._call(_lib.pn2x_sa_mlp_max, "sa_mlp_max_kernel", None, B, N, S, K, C1, C2, C3, pa, lda, px, pcx,
                                     pwx, pb1, pc, ldc, pi, pw2, pb2, pw3, pb3, po, ob, os_, oc, _native._stream(idx)),
                       "sa_mlp_max")
    return out


class _SaProblem(ctypes.Structure):
    """include/pn2_ext.h: pn2x_sa_problem."""
    _fields_ = [("n", _ci), ("s", _ci), ("k", _ci), ("a1f", _vp), ("a1f_ld", _ci), ("xyz", _vp), ("cxyz", _vp), ("wx", _vp),
                ("b1", _vp), ("cadd", _vp), ("cadd_ld", _ci), ("idx", _vp), ("w2", _vp), ("b2", _vp), ("w3", _vp), ("b3", _vp),
                ("out", _vp), ("out_b", _cl), ("out_s", _ci), ("out_c", _ci)]


_lib.pn2x_sa_mlp_max_pair.argtypes = [_ci] * 4 + [ctypes.POINTER(_SaProblem)] * 2 + [_vp]
_lib.pn2x_sa_mlp_max_pair.restype = _ci
_lib.pn2x_sa_mlp_max_pair_supported.argtypes = [_ci] * 5
_lib.pn2x_sa_mlp_max_pair_supported.restype = _ci


def _sa_problem(idx, w2, b2, w3, b3, a1f, xyz, cxyz, wx, b1, cadd, out) -> _SaProblem:
    B, S, K = idx.shape
    C1, C2, C3 = w2.shape[1], w2.shape[0], w3.shape[0]
    f32 = torch.float32
    N = a1f.shape[1] if a1f is not None else xyz.shape[1]
    pa, lda = (None, 0) if a1f is None else _rows(a1f, "a1f", C1)
    pc, ldc = (None, 0) if cadd is None else _rows(cadd, "cadd", C1)
    po, ld = _rows(out, "out", C3)
    return _SaProblem(
        N, S, K, pa, lda, None if xyz is None else _native._ptr(xyz, "xyz", f32, B * N * 3),
        None if cxyz is None else _native._ptr(cxyz, "cxyz", f32, B * S * 3), None if wx is None else _native._ptr(wx, "wx", f32, C1 * 3),
        None if b1 is None else _native._ptr(b1, "b1", f32, C1), pc, ldc, _native._ptr(idx, "idx", torch.int32, B * S * K),
        _native._ptr(w2, "w2", f32, C2 * C1), _native._ptr(b2, "b2", f32, C2), _native._ptr(w3, "w3", f32, C3 * C2),
        _native._ptr(b3, "b3", f32, C3), po, out.stride(0), ld, 1)


def sa_mlp_max_pair(p0: dict, p1: dict) -> None:
    """Both scales of a keypoint-query module in ONE launch (pn2x_sa_mlp_max_pair).  p0 / p1: the keyword arguments of
    sa_mlp_max (idx, w2, b2, w3, b3, a1f, xyz, cxyz, wx, b1, cadd, out -- `out` required, a (B,S,C3) row block).  Falls back
    to two launches when the combination is not covered by the pair kernel."""
    keys = ("idx", "w2", "b2", "w3", "b3", "a1f", "xyz", "cxyz", "wx", "b1", "cadd", "out")
    a = [{k: p.get(k) for k in keys} for p in (p0, p1)]
    C1, C2, C3 = a[0]["w2"].shape[1], a[0]["w2"].shape[0], a[0]["w3"].shape[0]
    same = all(tuple(a[1][k].shape) == tuple(a[0][k].shape) for k in ("w2", "w3"))
    modes = [(p["a1f"] is not None, p["xyz"] is not None, p["cadd"] is not None) for p in a]
    if not (same and modes[0] == modes[1] and modes[0][0] and modes[0][1]
            and _lib.pn2x_sa_mlp_max_pair_supported(a[0]["idx"].shape[2], a[1]["idx"].shape[2], C1, C2, C3)):
        for p in a:
            sa_mlp_max(p["idx"], p["w2"], p["b2"], p["w3"], p["b3"], a1f=p["a1f"], xyz=p["xyz"], cxyz=p["cxyz"], wx=p["wx"], b1=p["b1"],
                       cadd=p["cadd"], out=p["out"])
        return
    B = a[0]["idx"].shape[0]
    q0, q1 = (_sa_problem(**p) for p in a)
    with torch.cuda.device(a[0]["idx"].device):
        _native._check(_native._call(_lib.pn2x_sa_mlp_max_pair, "sa_mlp_max_pair_kernel", None, B, C1, C2, C3, ctypes.byref(q0),
                                     ctypes.byref(q1), _native._stream(a[0]["idx"])), "sa_mlp_max_pair")


_lib.pn2x_mlp2_rows.argtypes = [_cl] + [_ci] * 3 + [_vp, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _vp]
_lib.pn2x_mlp2_rows.restype = _ci
_lib.pn2x_mlp2_rows_supported.argtypes = [_ci] * 3
_lib.pn2x_mlp2_rows_supported.restype = _ci


_lib.pn2x_sa_set_compute_units.argtypes = [_ci]
_lib.pn2x_sa_set_compute_units.restype = _ci


def sa_set_compute_units(n: int) -> None:
    """CUs the persistent SA grids may occupy (0 = all): a serving loop with several batches in flight leaves a few to the
    other stream (include/pn2_ext.h: pn2x_sa_set_compute_units)."""
    _native._check(_lib.pn2x_sa_set_compute_units(int(n)), "sa_set_compute_units")


def mlp2_rows_supported(c1: int, c2: int, c3: int) -> bool:
    return bool(_lib.pn2x_mlp2_rows_supported(c1, c2, c3))


def mlp2_rows(x: torch.Tensor, w2, b2, w3, b3, out: torch.Tensor = None, w2e=None) -> torch.Tensor:
    """relu(relu(x[:, :c1] W2^T + x[:, c1:c1+3] W2e^T + b2) W3^T + b3) over the rows of a 2-D float32 GPU tensor (last dim
    contiguous, any row stride) in one launch (include/pn2_ext.h: pn2x_mlp2_rows).  `w2e` (c2, 3) or None; `out` may be a
    column block of a wider buffer whose row stride is a multiple of 4 floats and whose first element is 16-byte aligned (rows are
    stored as 16-byte quads)."""
    if x.dim() != 2 or not x.is_cuda or x.dtype != torch.float32 or x.stride(1) != 1:
        raise TypeError("mlp2_rows: x must be a 2-D float32 GPU tensor with a contiguous last dimension")
    R = x.shape[0]
    C1, C2, C3 = w2.shape[1], w2.shape[0], w3.shape[0]
    f32 = torch.float32
    ldx = x.stride(0) if R > 1 else x.shape[1]
    if x.shape[1] < C1 + (3 if w2e is not None else 0) or (w2e is not None and ldx < C1 + 4):
        raise ValueError("mlp2_rows: x has too few columns")
    if out is None:
        out = torch.empty((R, C3), dtype=f32, device=x.device)
    if out.dim() != 2 or out.shape[0] != R or out.shape[1] < C3 or out.stride(1) != 1 or out.dtype != f32 or out.device != x.device:
        raise ValueError("mlp2_rows: out must be (rows, >= c3) float32 on the same device")
    if (R > 1 and out.stride(0) % 4) or out.data_ptr() % 16:
        raise ValueError("mlp2_rows: out needs a row stride that is a multiple of 4 floats and a 16-byte aligned first element")
    with torch.cuda.device(x.device):
        _native._check(_native._call(_lib.pn2x_mlp2_rows, "mlp2_rows_kernel", None, R, C1, C2, C3, x.data_ptr(), ldx,
                                     _native._ptr(w2, "w2", f32, C2 * C1), None if w2e is None else _native._ptr(w2e, "w2e", f32, C2 * 3),
                                     _native._ptr(b2, "b2", f32, C2), _native._ptr(w3, "w3", f32, C3 * C2),
                                     _native._ptr(b3, "b3", f32, C3), out.data_ptr(), out.stride(0) if R > 1 else out.shape[1], _native._stream(x)),
                       "mlp2_rows")
    return out


def three_nn_weights(unknown: torch.Tensor, known: torch.Tensor):
    """unknown (B,n,3), known (B,m>=3,3) -> (weight (B,n,3) normalised inverse distances, idx (B,n,3) int32)."""
    B, n, _ = unknown.shape
    m = known.shape[1]
    f32 = torch.float32
    w = torch.empty((B, n, 3), dtype=f32, device=unknown.device)
    idx = torch.empty((B, n, 3), dtype=torch.int32, device=unknown.device)
    with torch.cuda.device(unknown.device):
        _native._check(_lib.pn2x_three_nn_weights(B, n, m, _native._ptr(unknown, "unknown", f32, B * n * 3),
                                                  _native._ptr(known, "known", f32, B * m * 3), w.data_ptr(), idx.data_ptr(),
                                                  _native._stream(unknown)), "three_nn_weights")
    return w, idx


def three_interpolate_pm(points: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """points (B,M,C) rows, idx/weight (B,n,3) -> out (B,n,C) rows (may be a column block of a wider buffer)."""
    B, M, C = points.shape
    n = idx.shape[1]
    pp, ldp = _rows(points, "points", C)
    po, ldo = _rows(out, "out", C)
    with torch.cuda.device(points.device):
        _native._check(_lib.pn2x_three_interpolate_pm(B, C, M, n, pp, ldp, _native._ptr(idx, "idx", torch.int32, B * n * 3),
                                                      _native._ptr(weight, "weight", torch.float32, B * n * 3), po, ldo,
                                                      _native._stream(points)), "three_interpolate_pm")
    return out


_lib.pn2x_three_nn_interpolate_pm.argtypes = [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _ci, _vp, _ci, _vp]
_lib.pn2x_three_nn_interpolate_pm.restype = _ci
_lib.pn2x_three_nn_interpolate_pm_supported.argtypes = [_ci] * 6
_lib.pn2x_three_nn_interpolate_pm_supported.restype = _ci
NN_INTERP_FUSED = True  # (module attribute: tests / A-B runs set it False for the two-launch chain)


def three_nn_interpolate_pm(unknown: torch.Tensor, known: torch.Tensor, points: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out (B,n,C) rows <- three_interpolate_pm(points, *three_nn_weights(unknown, known)[::-1]) -- one launch when the sizes are
    covered (include/pn2_ext.h: pn2x_three_nn_interpolate_pm), the two launches otherwise; same floats either way."""
    B, n, _ = unknown.shape
    m, C = known.shape[1], points.shape[2]
    pp, ldp = _rows(points, "points", C)
    po, ldo = _rows(out, "out", C)
    f32 = torch.float32
    if (NN_INTERP_FUSED and _lib.pn2x_three_nn_interpolate_pm_supported(B, n, m, C, ldp, ldo) and pp % 16 == 0 and po % 16 == 0):
        with torch.cuda.device(points.device):
            _native._check(_native._call(_lib.pn2x_three_nn_interpolate_pm, "three_nn_interp_kernel", None, B, n, m, C,
                                         _native._ptr(unknown, "unknown", f32, B * n * 3), _native._ptr(known, "known", f32, B * m * 3),
                                         pp, ldp, po, ldo, _native._stream(points)), "three_nn_interpolate_pm")
        return out
    w, i3 = three_nn_weights(unknown, known)
    return three_interpolate_pm(points, i3, w, out)


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """src (B,N,C) contiguous, idx (B,M) int32 -> (B,M,C)."""
    B, N, C = src.shape
    M = idx.shape[1]
    out = torch.empty((B, M, C), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        _native._check(_lib.pn2x_gather_rows(B, N, M, C, _native._ptr(src, "src", torch.float32, B * N * C),
                                             _native._ptr(idx, "idx", torch.int32, B * M), out.data_ptr(),
                                             _native._stream(src)), "gather_rows")
    return out


def bias_act_pm_(y: torch.Tensor, bias: torch.Tensor, rows_per_bias: int, relu: bool = True) -> torch.Tensor:
    """In place on point-major y (B,R,C): y[b,r,:] = act(y[b,r,:] + bias[(b*R+r)//rows_per_bias, :])."""
    B, R, C = y.shape
    py, ldy = _rows(y, "y", C)
    with torch.cuda.device(y.device):
        _native._check(_lib.pn2x_bias_act_pm(B * R, C, py, ldy, _native._ptr(bias.contiguous(), "bias", torch.float32, bias.numel()),
                                             rows_per_bias, 1 if relu else 0, _native._stream(y)), "bias_act_pm")
    return y


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


_lib.pn2x_max_rows.argtypes = [_ci, _ci, _ci, _vp, _vp, _vp]
_lib.pn2x_max_rows.restype = _ci


def max_rows(x: torch.Tensor) -> torch.Tensor:
    """x (B,R,C) contiguous -> (B,C) max over the R rows."""
    B, R, C = x.shape
    out = torch.empty((B, C), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _native._check(_lib.pn2x_max_rows(B, R, C, _native._ptr(x, "x", torch.float32, B * R * C), out.data_ptr(),
                                          _native._stream(x)), "max_rows")
    return out


_lib.pn2x_hand_frame.argtypes = [_ci] * 5 + [_vp] * 4 + [ctypes.c_float] + [_vp] * 5
_lib.pn2x_hand_frame.restype = _ci
_lib.pn2x_hand_frame2.argtypes = [_ci] * 5 + [_vp] * 4 + [ctypes.c_float] + [_vp] * 5 + [_ci, _vp]
_lib.pn2x_hand_frame2.restype = _ci
_lib.pn2x_hand_frame3.argtypes = [_ci] * 5 + [_vp] * 4 + [ctypes.c_float] + [_vp] * 5 + [_ci, _vp, _vp]
_lib.pn2x_hand_frame3.restype = _ci


def _xyz_cols(t: torch.Tensor, name: str, B: int, R: int):
    """A (B, R, 3) column block of a wider fp32 row buffer -> (data_ptr, row stride)."""
    if not t.is_cuda or t.dtype != torch.float32 or tuple(t.shape) != (B, R, 3) or t.stride(2) != 1 or \
            (B > 1 and t.stride(0) != R * t.stride(1)):
        raise ValueError(f"{name}: expected a (B,{R},3) float32 column block with uniform row stride, got {tuple(t.shape)} {t.stride()}")
    return t.data_ptr(), t.stride(1)


def hand_frame(palm_template: torch.Tensor, kp: torch.Tensor, palm_idx: torch.Tensor, points: torch.Tensor, scale: float,
               xyz2_copy: torch.Tensor = None, nonfinite: torch.Tensor = None):
    """Kabsch(palm_template -> kp[:, palm_idx]) + canonicalisation in one launch.
    Returns R (B,3,3), t (B,3,1), xyz2 (B,N,3), xyz1 (B,J,3).  xyz2_copy: a (B,N,3) column block of a consumer's row
    buffer that receives a second copy of xyz2.  nonfinite: (B,) int32 that receives 1 for frames with a NaN / Inf input."""
    if palm_template.dim() == 2:
        palm_template = palm_template.unsqueeze(0)
    palm_template, kp, points = palm_template.float().contiguous(), kp.float().contiguous(), points.float().contiguous()
    B, N, _ = points.shape
    J = kp.shape[1]
    xb, num = palm_template.shape[0], palm_template.shape[1]
    f32 = torch.float32
    R = torch.empty((B, 3, 3), dtype=f32, device=points.device)
    t = torch.empty((B, 3, 1), dtype=f32, device=points.device)
    xyz2 = torch.empty((B, N, 3), dtype=f32, device=points.device)
    xyz1 = torch.empty((B, J, 3), dtype=f32, device=points.device)
    pc, ldc = (None, 0) if xyz2_copy is None else _xyz_cols(xyz2_copy, "xyz2_copy", B, N)
    with torch.cuda.device(points.device):
        _native._check(_lib.pn2x_hand_frame3(B, xb, num, N, J, _native._ptr(palm_template, "palm_template", f32, xb * num * 3),
                                             _native._ptr(kp, "kp", f32, B * J * 3), _native._ptr(palm_idx, "palm_idx", torch.int32, num),
                                             _native._ptr(points, "points", f32, B * N * 3), float(scale), R.data_ptr(), t.data_ptr(),
                                             xyz2.data_ptr(), xyz1.data_ptr(), pc, ldc,
                                             None if nonfinite is None else _native._ptr(nonfinite, "nonfinite", torch.int32, B),
                                             _native._stream(points)), "hand_frame")
    return R, t, xyz2, xyz1


_cf = ctypes.c_float
_lib.pn2x_add_layernorm.argtypes = [_cl, _ci, _vp, _vp, _vp, _vp, _vp, _cf, _vp, _vp, _cf, _vp, _vp]
_lib.pn2x_add_layernorm.restype = _ci
_lib.pn2x_pose_head2.argtypes = [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _cf, _vp, _vp, _vp, _vp]
_lib.pn2x_pose_head2.restype = _ci


def add_layernorm(x: torch.Tensor, ln1, y: torch.Tensor = None, bias: torch.Tensor = None, ln2=None) -> torch.Tensor:
    """LN2(LN1(x + y + bias)) over the last dim of row-major x (rows, C); ln1 / ln2 are nn.LayerNorm modules
    (ln2 optional), y (rows, C) and bias (C,) optional.  One launch (include/pn2_ext.h: pn2x_add_layernorm)."""
    rows, C = x.shape
    f32 = torch.float32
    px = _native._ptr(x, "x", f32, rows * C)
    py = None if y is None else _native._ptr(y, "y", f32, rows * C)
    pb = None if bias is None else _native._ptr(bias, "bias", f32, C)
    for ln in (ln1, ln2):
        if ln is not None and (tuple(ln.normalized_shape) != (C,) or ln.weight is None or ln.bias is None):
            raise ValueError("add_layernorm: LayerNorm over the last dimension with affine parameters expected")
    g1, b1 = _native._ptr(ln1.weight, "ln1.weight", f32, C), _native._ptr(ln1.bias, "ln1.bias", f32, C)
    g2 = b2 = None
    eps2 = 0.0
    if ln2 is not None:
        g2, b2, eps2 = _native._ptr(ln2.weight, "ln2.weight", f32, C), _native._ptr(ln2.bias, "ln2.bias", f32, C), ln2.eps
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _native._check(_lib.pn2x_add_layernorm(rows, C, px, py, pb, g1, b1, ln1.eps, g2, b2, eps2, out.data_ptr(), _native._stream(x)),
                       "add_layernorm")
    return out


def pose_head(h: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, xyz1: torch.Tensor, R: torch.Tensor, t: torch.Tensor, scale: float,
              nonfinite: torch.Tensor = None):
    """h (B*J, C), w (3, C), bias (3,), xyz1 (B,J,3), R (B,3,3), t (B,3,1) -> (kp_hand (B,J,3), kp_cam (B,J,3)):
    kp_hand = h w^T + bias + xyz1;  kp_cam = (kp_hand R^T) * scale + t  (include/pn2_ext.h: pn2x_pose_head)."""
    B, J, _ = xyz1.shape
    C = h.shape[1]
    f32 = torch.float32
    ptrs = (_native._ptr(h, "h", f32, B * J * C), _native._ptr(w, "w", f32, 3 * C), _native._ptr(bias, "bias", f32, 3),
            _native._ptr(xyz1, "xyz1", f32, B * J * 3), _native._ptr(R, "R", f32, B * 9), _native._ptr(t, "t", f32, B * 3))
    kp_hand = torch.empty((B, J, 3), dtype=f32, device=h.device)
    kp_cam = torch.empty((B, J, 3), dtype=f32, device=h.device)
    with torch.cuda.device(h.device):
        _native._check(_lib.pn2x_pose_head2(B, J, C, *ptrs, float(scale), kp_hand.data_ptr(), kp_cam.data_ptr(),
                                            None if nonfinite is None else _native._ptr(nonfinite, "nonfinite", torch.int32, B),
                                            _native._stream(h)), "pose_head")
    return kp_hand, kp_cam


_lib.pn2x_linear_small.argtypes = [_ci, _ci, _ci, _vp, _ci, _vp, _ci, _vp, _ci, _vp, _ci, _vp]
_lib.pn2x_linear_small.restype = _ci
# where pn2x_linear_small beats the library's best recorded solution (scripts/probes/linear_small_bench.py, profiles/
# r03_linear_small.json): 2 ... 512 rows, reductions up to 384 deep, up to 512 outputs -- e.g. 128 x 128 -> 256: 4.7 vs 20 us,
# 128 x 131 -> 128: 7.0 vs 12.8; it loses on one row, on deep reductions (21 x 1024 -> 384: 17 vs 5.9) and from ~1000 rows
LINEAR_SMALL_MAX_ROWS = 512
LINEAR_SMALL_MAX_K, LINEAR_SMALL_MAX_N = 384, 512


def linear(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor = None, relu: bool = False) -> torch.Tensor:
    """act(x w^T + bias) for x (M, K) rows, w (N, K): small problems (the B = 1 / B = 8 tracking loop; bounds above) through
    pn2x_linear_small (one workgroup per 32 x 32 output block), everything else through the BLAS library (torch, with its fused bias + ReLU epilogue).  Inference only."""
    M, K = x.shape
    N = w.shape[0]
    if (M > LINEAR_SMALL_MAX_ROWS or M < 2 or K > LINEAR_SMALL_MAX_K or N > LINEAR_SMALL_MAX_N or not x.is_cuda or x.dtype != torch.float32 or w.dtype != torch.float32 or x.stride(1) != 1
            or w.stride(1) != 1 or (bias is not None and not bias.is_contiguous()) or torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)):
        if relu and bias is not None:
            return torch._addmm_activation(bias, x, w.t())
        y = torch.nn.functional.linear(x, w, bias)
        return torch.relu_(y) if relu else y
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _native._check(_lib.pn2x_linear_small(M, K, N, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0),
                                              None if bias is None else bias.data_ptr(), 1 if relu else 0, y.data_ptr(), N,
                                              _native._stream(x)), "linear_small")
    return y


_lib.pn2x_ln_linear_small.argtypes = [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _cf, _vp, _vp, _cf, _vp, _vp, _ci, _vp, _ci, _vp, _ci, _vp]
_lib.pn2x_ln_linear_small.restype = _ci
LN_LINEAR_MAX_ROWS = 256  # 0: always the two launches (B = 8: 0.419 -> 0.410 ms with 168 rows through it)


def ln_linear_supported(rows: int, c: int) -> bool:
    """The LayerNorm launch in front of a small Linear folded into it (pn2x_ln_linear_small): few rows, c <= 384."""
    return 0 < rows <= LN_LINEAR_MAX_ROWS and c <= 384


def ln_linear(x: torch.Tensor, ln1, w: torch.Tensor, bias: torch.Tensor = None, relu: bool = False, y: torch.Tensor = None,
              ybias: torch.Tensor = None, ln2=None):
    """(xn, act(xn w^T + bias)) with xn = ln2(ln1(x + y + ybias)) over the last dimension of x (rows, C) -- add_layernorm and
    linear in one launch (include/pn2_ext.h: pn2x_ln_linear_small; same bits as the two).  Inference only."""
    rows, C = x.shape
    N = w.shape[0]
    f32 = torch.float32
    for ln in (ln1, ln2):
        if ln is not None and (tuple(ln.normalized_shape) != (C,) or ln.weight is None or ln.bias is None):
            raise ValueError("ln_linear: LayerNorm over the last dimension with affine parameters expected")
    if x.dtype != f32 or not x.is_cuda or not x.is_contiguous() or w.dtype != f32 or w.stride(1) != 1 or w.shape[1] != C:
        raise TypeError("ln_linear: contiguous float32 GPU rows and a (N, C) float32 weight expected")
    if y is not None and (y.shape != x.shape or not y.is_contiguous() or y.dtype != f32):
        raise TypeError("ln_linear: y must match x")
    xn = torch.empty_like(x)
    out = torch.empty((rows, N), dtype=f32, device=x.device)
    with torch.cuda.device(x.device):
        _native._check(_lib.pn2x_ln_linear_small(
            rows, C, N, x.data_ptr(), None if y is None else y.data_ptr(), None if ybias is None else _native._ptr(ybias, "ybias", f32, C),
            _native._ptr(ln1.weight, "ln1.weight", f32, C), _native._ptr(ln1.bias, "ln1.bias", f32, C), ln1.eps,
            None if ln2 is None else _native._ptr(ln2.weight, "ln2.weight", f32, C), None if ln2 is None else _native._ptr(ln2.bias, "ln2.bias", f32, C),
            0.0 if ln2 is None else ln2.eps, xn.data_ptr(), w.data_ptr(), w.stride(0), None if bias is None else _native._ptr(bias, "bias", f32, N),
            1 if relu else 0, out.data_ptr(), N, _native._stream(x)), "ln_linear_small")
    return xn, out


_lib.pn2x_knn_indices.argtypes = [_ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp]
_lib.pn2x_knn_indices.restype = _ci


def knn_indices(k: int, unknown: torch.Tensor, known: torch.Tensor, k2: int = 0):
    """Indices (B,n,k) int32 of the k nearest `known` points of every `unknown` point, sorted by (distance, index) --
    pointnet2_utils.knn without the sqrt / distance output the caller would discard.  k2 > 0: also the first k2 of
    every list as a second contiguous (B,n,k2) tensor (returns a pair)."""
    B, n, _ = unknown.shape
    m = known.shape[1]
    pu, pk = _native._ptr(unknown, "unknown", torch.float32, B * n * 3), _native._ptr(known, "known", torch.float32, B * m * 3)
    idx = torch.empty((B, n, k), dtype=torch.int32, device=unknown.device)
    idx2 = torch.empty((B, n, k2), dtype=torch.int32, device=unknown.device) if k2 else None
    with torch.cuda.device(unknown.device):
        _native._check(_lib.pn2x_knn_indices(B, n, m, k, k2, pu, pk, idx.data_ptr(), None if idx2 is None else idx2.data_ptr(),
                                             _native._stream(unknown)), "knn_indices")
    return (idx, idx2) if k2 else idx


_lib.pn2x_furthest_point_sampling_radii.argtypes = [_ci, _ci, _ci, _vp, _vp, _vp, _vp]
_lib.pn2x_furthest_point_sampling_radii.restype = _ci
_lib.pn2x_fps_radii_knn.argtypes = [_ci, _ci, _ci, _vp, _vp, _vp, _ci, _ci, _ci, _vp, _vp, _vp, _vp]
_lib.pn2x_fps_radii_knn.restype = _ci
_lib.pn2x_fps_radii_knn_supported.argtypes = [_ci, _ci, _ci]
_lib.pn2x_fps_radii_knn_supported.restype = _ci
_lib.pn2x_fps_prefix_ties.argtypes = [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp]
_lib.pn2x_fps_prefix_ties.restype = _ci
_lib.pn2x_fps_prefix_flags.argtypes = [_ci]
_lib.pn2x_fps_prefix_flags.restype = _ci
_lib.pn2x_furthest_point_sampling_prefix.argtypes = [_ci, _ci, _ci, _vp, _vp, _ci, _vp, _vp]
_lib.pn2x_furthest_point_sampling_prefix.restype = _ci


_lib.pn2x_ball_query_picks2.argtypes = [_ci, _ci, _ci, ctypes.c_float, _ci, _vp, _vp, _vp, _vp, _vp, _ci, _vp]
_lib.pn2x_ball_query_picks2.restype = _ci


def ball_query_picks(radius: float, nsample: int, xyz: torch.Tensor, picks: torch.Tensor, xyz_copy: torch.Tensor = None):
    """Ball query around the centroids xyz[picks] (picks (B,S) int32 from FPS) -> (idx (B,S,nsample) int32,
    new_xyz (B,S,3) = the centroids' coordinates): pointnet2_utils.ball_query + the gather before it, one launch.
    xyz_copy: a (B,S,3) column block of a consumer's row buffer that receives a second copy of new_xyz."""
    B, N, _ = xyz.shape
    S = picks.shape[1]
    px = _native._ptr(xyz, "xyz", torch.float32, B * N * 3)
    pp = _native._ptr(picks, "picks", torch.int32, B * S)
    idx = torch.empty((B, S, nsample), dtype=torch.int32, device=xyz.device)
    new_xyz = torch.empty((B, S, 3), dtype=torch.float32, device=xyz.device)
    pc, ldc = (None, 0) if xyz_copy is None else _xyz_cols(xyz_copy, "xyz_copy", B, S)
    with torch.cuda.device(xyz.device):
        _native._check(_lib.pn2x_ball_query_picks2(B, N, S, float(radius), nsample, px, pp, new_xyz.data_ptr(), idx.data_ptr(),
                                                   pc, ldc, _native._stream(xyz)), "ball_query_picks")
    return idx, new_xyz


FPS_KNN_COLAUNCH = True  # (module attributes: tests compare the co-launches with the separate launches)
BALL_TIE_COLAUNCH = True
_lib.pn2x_ball_query_picks_ties.argtypes = [_ci, _ci, _ci, ctypes.c_float, _ci, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _vp, _vp, _vp]
_lib.pn2x_ball_query_picks_ties.restype = _ci
_lib.pn2x_ball_query_picks_ties_supported.argtypes = [_ci] * 4
_lib.pn2x_ball_query_picks_ties_supported.restype = _ci


def fps_two_level(xyz: torch.Tensor, m1: int, m2: int, query=None, knn=None):
    """The reference's two chained samplings  i1 = FPS(xyz, m1); l1 = xyz[i1]; i2 = FPS(l1, m2)  (backbones.py:98-104)
    -> (i1 (B,m1), l1 (B,m1,3), i2 (B,m2)) int32/float32, bit-identical to running both.  The second pass is
    skipped per cloud when level 1 had no tied arg-max among its first m2 picks (include/pn2_ext.h).
    query=(radius, nsample): level 1's ball query is done by the launch that produces l1 (ball_query_picks) and its
    index tensor (B,m1,nsample) is returned as a fourth value.
    knn=(points (B,nq,3), k, k2): also knn_indices(k, points, xyz, k2) -- appended to the result as one more value, the pair
    (idx (B,nq,k), idx2 (B,nq,k2) | None) -- in the launch of the first sampling level when the kernels cover the sizes
    (include/pn2_ext.h: pn2x_fps_radii_knn), as its own launch otherwise."""
    from . import pointnet2_utils as ops
    B, N, _ = xyz.shape
    if not 1 <= m2 <= m1:
        raise ValueError("fps_two_level: need 1 <= m2 <= m1")
    xyz = xyz.contiguous()

    def knn_alone():
        r = knn_indices(knn[1], knn[0], xyz, k2=knn[2])
        return r if knn[2] else (r, None)
    if m2 > 1024 or N > 16384:  # beyond the shortcut's kernels: two plain passes
        i1 = ops.furthest_point_sample(xyz, m1)
        l1 = gather_rows(xyz, i1)
        i2 = ops.furthest_point_sample(l1, m2)
        res = (i1, l1, i2) if query is None else (i1, l1, i2, ops.ball_query(query[0], query[1], xyz, l1))
        return res if knn is None else res + (knn_alone(),)
    px = _native._ptr(xyz, "xyz", torch.float32, B * N * 3)
    nf = _lib.pn2x_fps_prefix_flags(N)
    i1 = torch.empty((B, m1), dtype=torch.int32, device=xyz.device)
    radii = torch.empty((B, m1), dtype=torch.float32, device=xyz.device)
    flags = torch.empty((B, nf), dtype=torch.int32, device=xyz.device)
    i2 = torch.empty((B, m2), dtype=torch.int32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        st = _native._stream(xyz)
        lists = None
        if knn is not None and FPS_KNN_COLAUNCH and _lib.pn2x_fps_radii_knn_supported(N, knn[0].shape[1], knn[1]):
            pts, k, k2 = knn
            nq = pts.shape[1]
            gi = torch.empty((B, nq, k), dtype=torch.int32, device=xyz.device)
            gi2 = torch.empty((B, nq, k2), dtype=torch.int32, device=xyz.device) if k2 else None
            _native._check(_native._call(_lib.pn2x_fps_radii_knn, "fps_knn_kernel", None, B, N, m1, px, i1.data_ptr(), radii.data_ptr(),
                                         nq, k, k2, _native._ptr(pts, "knn points", torch.float32, B * nq * 3), gi.data_ptr(),
                                         None if gi2 is None else gi2.data_ptr(), st), "fps_two_level/1+knn")
            lists = (gi, gi2)
        else:
            _native._check(_native._call(_lib.pn2x_furthest_point_sampling_radii, "fps_kernel", None, B, N, m1, px, i1.data_ptr(),
                                         radii.data_ptr(), st), "fps_two_level/1")
        if query is not None and BALL_TIE_COLAUNCH and _lib.pn2x_ball_query_picks_ties_supported(B, N, m1, m2):
            # level 1's ball query and the tie check of the sampling run both start from the picks: one launch
            idx1 = torch.empty((B, m1, query[1]), dtype=torch.int32, device=xyz.device)
            l1 = torch.empty((B, m1, 3), dtype=torch.float32, device=xyz.device)
            _native._check(_native._call(_lib.pn2x_ball_query_picks_ties, "ball_tie_kernel", None, B, N, m1, float(query[0]), query[1], px,
                                         i1.data_ptr(), l1.data_ptr(), idx1.data_ptr(), None, 0, m2, radii.data_ptr(), flags.data_ptr(), st),
                           "fps_two_level/query+ties")
        else:
            if query is None:
                l1 = gather_rows(xyz, i1)
            else:
                idx1, l1 = ball_query_picks(query[0], query[1], xyz, i1)
            _native._check(_lib.pn2x_fps_prefix_ties(B, N, m1, m2, px, i1.data_ptr(), radii.data_ptr(), flags.data_ptr(), st), "fps_two_level/ties")
        _native._check(_native._call(_lib.pn2x_furthest_point_sampling_prefix, "fps_prefix_kernel", None, B, m1, m2, l1.data_ptr(),
                                     flags.data_ptr(), nf, i2.data_ptr(), st), "fps_two_level/2")
    res = (i1, l1, i2) if query is None else (i1, l1, i2, idx1)
    if knn is None:
        return res
    return res + (lists if lists is not None else knn_alone(),)


_lib.pn2x_hand_losses.argtypes = [_ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp]
_lib.pn2x_hand_losses.restype = _ci
_lib.pn2x_hand_losses_backward.argtypes = [_ci, _ci, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _vp]
_lib.pn2x_hand_losses_backward.restype = _ci
_lib.pn2x_hand_losses2.argtypes = _lib.pn2x_hand_losses.argtypes[:-1] + [_vp, _vp]
_lib.pn2x_hand_losses2.restype = _ci
_lib.pn2x_hand_losses_backward2.argtypes = [_ci, _ci, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
_lib.pn2x_hand_losses_backward2.restype = _ci
HAND_LOSS_NAMES = ("hand_pred_kp_loss", "hand_pred_r_loss", "hand_pred_t_loss", "hand_pred_kp_diff", "hand_init_kp_diff",
                   "hand_init_r_diff", "hand_init_t_diff", "hand_pred_r_diff", "hand_pred_t_diff")


class HandLosses(torch.autograd.Function):
    """The nine entries of HandTrackNet.compute_loss's dictionary (include/pn2_ext.h: pn2x_hand_losses) as a (9,) tensor;
    differentiable with respect to pred_hf through the first three (keypoint L1, rotation L1 and translation L1 of the palm
    fit -- the closed-form Kabsch gradient), the rest are metrics."""

    @staticmethod
    def forward(ctx, pred_hf, init_hf, gt_kp, pred_kp, R, t, scale, palm, weights=None):
        # weights (9,) on the device: also returns sum_i weights[i] out[i] as a second (0-dim) output (the trainer's weighted
        # total, trainer.py:157-165) -- no multiply / sum launches, and its gradient goes straight into the backward kernel
        f32 = torch.float32
        B = pred_hf.shape[0]
        ctx.set_materialize_grads(False)
        pred_hf = pred_hf.contiguous()
        palm = palm.contiguous().float()
        if palm.dim() == 2:
            palm = palm.unsqueeze(0)
        args = [x.detach().contiguous().float() for x in (init_hf, gt_kp, pred_kp, R, t)]
        out = torch.empty(10, dtype=f32, device=pred_hf.device)
        saved = torch.empty((B, 87), dtype=f32, device=pred_hf.device)
        wptr = None if weights is None else _native._ptr(weights, "weights", f32, 9)
        with torch.cuda.device(pred_hf.device):
            _native._check(_lib.pn2x_hand_losses2(B, palm.shape[0], _native._ptr(pred_hf, "pred_hf", f32, B * 63), _native._ptr(args[0], "init_hf", f32, B * 63),
                                                  _native._ptr(args[1], "gt_kp", f32, B * 63), _native._ptr(args[2], "pred_kp", f32, B * 63),
                                                  _native._ptr(args[3], "R", f32, B * 9), _native._ptr(args[4], "t", f32, B * 3), float(scale),
                                                  _native._ptr(palm, "palm", f32, palm.shape[0] * 18), out.data_ptr(), saved.data_ptr(), wptr,
                                                  _native._stream(pred_hf)), "hand_losses")
        ctx.save_for_backward(pred_hf, palm, saved, *([weights] if weights is not None else []))
        ctx.scale = float(scale)
        ctx.set_materialize_grads(False)  # (backward handles None: no zero-fill launch for the output nobody differentiates)
        if weights is None:
            return out[:9]
        return out[:9], out[9]

    @staticmethod
    def backward(ctx, grad, grad_total=None):
        pred_hf, palm, saved = ctx.saved_tensors[:3]
        weights = ctx.saved_tensors[3] if len(ctx.saved_tensors) > 3 else None
        B = pred_hf.shape[0]
        if grad is None and grad_total is None:
            return (None,) * 9
        g3 = None if grad is None else grad[:3].contiguous().float()
        gt = None if grad_total is None else grad_total.contiguous().float()
        d = torch.empty_like(pred_hf)
        with torch.cuda.device(pred_hf.device):
            _native._check(_lib.pn2x_hand_losses_backward2(B, palm.shape[0], pred_hf.data_ptr(), ctx.scale, palm.data_ptr(), saved.data_ptr(),
                                                           None if g3 is None else g3.data_ptr(), None if gt is None else gt.data_ptr(),
                                                           None if weights is None else weights.data_ptr(), d.data_ptr(),
                                                           _native._stream(pred_hf)), "hand_losses_backward")
        return d, None, None, None, None, None, None, None, None


_lib.pn2x_copy_multi_max.restype = _ci
_lib.pn2x_copy_multi.argtypes = [_ci, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_long), _vp]
_lib.pn2x_copy_multi.restype = _ci


def copy_multi(dsts, srcs) -> None:
    """dsts[i].copy_(srcs[i]) for same-shaped, same-dtype contiguous tensors on ONE GPU as one launch (pn2x_copy_multi)."""
    n = len(dsts)
    if n == 0:
        return
    cap = int(_lib.pn2x_copy_multi_max())
    dev = dsts[0].device
    for d, s_ in zip(dsts, srcs):
        if (not d.is_cuda or d.device != dev or s_.device != dev or d.dtype != s_.dtype or d.shape != s_.shape
                or not d.is_contiguous() or not s_.is_contiguous()):
            raise ValueError("copy_multi: same-shaped contiguous tensors of one dtype on one GPU")
    with torch.cuda.device(dev):
        for i0 in range(0, n, cap):
            k = min(cap, n - i0)
            D, S, Bn = (_vp * k)(), (_vp * k)(), (ctypes.c_long * k)()
            for j in range(k):
                D[j], S[j], Bn[j] = dsts[i0 + j].data_ptr(), srcs[i0 + j].data_ptr(), dsts[i0 + j].numel() * dsts[i0 + j].element_size()
            _native._check(_lib.pn2x_copy_multi(k, D, S, Bn, _native._stream(dsts[0])), "copy_multi")
