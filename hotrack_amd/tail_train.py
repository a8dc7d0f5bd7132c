"""The 21-token tail of HandTrackNet in training mode as autograd Functions over csrc/tail_train.hip
(include/pn2_ext.h: pn2x_tail_ln_fwd / _bwd, pn2x_tail_relu_drop_fwd / _bwd).

    TailGrads(device, sizes)             one zero-filled buffer per forward for every dgamma / dbeta / dbias of the tail
    ln(x, norm_a, norm_b, ...)           LN_b(LN_a(x + dropout(y + bias)))        (y / bias / norm_b optional)
    relu_dropout(z, bias, p, ...)        dropout(relu(z + bias))

torch semantics (LayerNorm: biased variance, eps inside the sqrt; dropout: kept elements scaled by 1 / (1 - p)); the dropout
masks are a hash of (per-forward seed, site, element index), regenerated in the backward.  GPU tensors only."""
from __future__ import annotations

import ctypes

import torch

from . import pointnet2_hip as _native

_lib = _native._lib
_vp, _ci, _cl, _cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
_lib.pn2x_tail_ln_fwd.argtypes = [_cl, _ci, _vp, _vp, _vp, _cf, _ci, _vp, _vp, _vp, _vp, _vp, _cf, _vp, _vp, _cf, _vp, _vp, _vp]
_lib.pn2x_tail_ln_fwd.restype = _ci
_lib.pn2x_tail_ln_bwd.argtypes = [_cl, _ci, _vp, _vp, _vp, _cf, _ci, _vp, _vp, _vp, _cf, _vp, _vp, _cf, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
_lib.pn2x_tail_ln_bwd.restype = _ci
_lib.pn2x_tail_relu_drop_fwd.argtypes = [_cl, _ci, _vp, _vp, _cf, _ci, _vp, _vp, _vp]
_lib.pn2x_tail_relu_drop_fwd.restype = _ci
_lib.pn2x_tail_relu_drop_bwd.argtypes = [_cl, _ci, _vp, _vp, _cf, _ci, _vp, _vp, _vp, _vp, _vp]
_lib.pn2x_tail_relu_drop_bwd.restype = _ci
_lib.pn2x_tail_pose_head_fwd.argtypes = [_ci, _ci, _ci] + [_vp] * 7 + [_ci] + [_vp] * 3
_lib.pn2x_tail_pose_head_fwd.restype = _ci
_lib.pn2x_tail_pose_head_bwd.argtypes = [_ci, _ci, _ci] + [_vp] * 6 + [_ci] + [_vp] * 4
_lib.pn2x_tail_pose_head_bwd.restype = _ci
_f32 = torch.float32


def _p(t):
    return None if t is None else t.data_ptr()


class TailGrads:
    """Zero-filled accumulators for the parameter gradients the backward kernels add into (one fill launch per forward)."""

    def __init__(self, device, n_floats: int):
        self.buf = torch.zeros(n_floats, dtype=_f32, device=device)
        self.used = 0

    def take(self, n: int) -> torch.Tensor:
        out = self.buf[self.used:self.used + n]
        self.used += n
        assert self.used <= self.buf.numel()
        return out


def _fresh(ctx, acc=None):
    """The forward's zeroed accumulators on the first backward through a node, fresh zeros on any later one (the kernels ADD)."""
    acc = ctx.acc if acc is None else acc
    if getattr(ctx, "_acc_used", False):
        return tuple(None if a is None else torch.zeros_like(a) for a in acc)
    ctx._acc_used = True
    return acc


class _Ln(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, bias, ga, ba, gb, bb, eps_a, eps_b, p, site, seed_in, seed_dev, seed_out, grads):
        rows, c = x.shape
        x = x.contiguous()
        y = None if y is None else y.contiguous()
        out = torch.empty_like(x)
        stats = torch.empty((rows, 4), dtype=_f32, device=x.device)
        with torch.cuda.device(x.device):
            _native._check(_lib.pn2x_tail_ln_fwd(rows, c, x.data_ptr(), _p(y), _p(bias), float(p), int(site), _p(seed_in), _p(seed_dev),
                                                 _p(seed_out), ga.data_ptr(), ba.data_ptr(), float(eps_a), _p(gb), _p(bb), float(eps_b),
                                                 out.data_ptr(), stats.data_ptr(), _native._stream(x)), "tail_ln_fwd")
        ctx.save_for_backward(x, y, bias, ga, ba, gb, bb, stats, seed_in)
        ctx.meta = (float(eps_a), float(eps_b), float(p), int(site))
        # accumulators of this call's parameter gradients (zeroed with the whole buffer at the start of the forward)
        ctx.acc = (grads.take(c), grads.take(c), grads.take(c) if gb is not None else None, grads.take(c) if gb is not None else None,
                   grads.take(c) if (y is not None and bias is not None) else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y, bias, ga, ba, gb, bb, stats, seed_in = ctx.saved_tensors
        eps_a, eps_b, p, site = ctx.meta
        rows, c = x.shape
        dout = dout.contiguous()
        dx = torch.empty_like(x)
        dy = torch.empty_like(x) if y is not None else None
        dga, dba, dgb, dbb, dbias = _fresh(ctx)
        with torch.cuda.device(x.device):
            _native._check(_lib.pn2x_tail_ln_bwd(rows, c, x.data_ptr(), _p(y), _p(bias), p, site, _p(seed_in), ga.data_ptr(), ba.data_ptr(), eps_a,
                                                 _p(gb), _p(bb), eps_b, stats.data_ptr(), dout.data_ptr(), dx.data_ptr(), _p(dy), dga.data_ptr(),
                                                 dba.data_ptr(), _p(dgb), _p(dbb), _p(dbias), _native._stream(x)), "tail_ln_bwd")
        # fresh view objects: autograd keeps a gradient it is the sole owner of instead of cloning it (a copy launch per parameter)
        v = lambda t: None if t is None else t[:]
        return dx, dy, v(dbias), v(dga), v(dba), v(dgb), v(dbb), None, None, None, None, None, None, None, None


def ln(x, norm_a, norm_b, grads, y=None, bias=None, p=0.0, site=0, seed_in=None, seed_dev=None, seed_out=None):
    """LN_b(LN_a(x + dropout(y + bias))) on rows (R, C); norm_a / norm_b: torch.nn.LayerNorm modules (norm_b may be None)."""
    gb, bb, eps_b = (None, None, 0.0) if norm_b is None else (norm_b.weight, norm_b.bias, norm_b.eps)
    return _Ln.apply(x, y, bias, norm_a.weight, norm_a.bias, gb, bb, norm_a.eps, eps_b, p, site, seed_in, seed_dev, seed_out, grads)


class _ReluDrop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, bias, p, site, seed_in, grads):
        rows, c = z.shape
        z = z.contiguous()
        out = torch.empty_like(z)
        with torch.cuda.device(z.device):
            _native._check(_lib.pn2x_tail_relu_drop_fwd(rows, c, z.data_ptr(), _p(bias), float(p), int(site), _p(seed_in), out.data_ptr(),
                                                        _native._stream(z)), "tail_relu_drop_fwd")
        ctx.save_for_backward(z, bias, seed_in)
        ctx.meta = (float(p), int(site))
        ctx.acc = grads.take(c) if bias is not None else None
        return out

    @staticmethod
    def backward(ctx, dh):
        z, bias, seed_in = ctx.saved_tensors
        p, site = ctx.meta
        rows, c = z.shape
        dh = dh.contiguous()
        dz = torch.empty_like(z)
        (acc,) = _fresh(ctx, (ctx.acc,))
        with torch.cuda.device(z.device):
            _native._check(_lib.pn2x_tail_relu_drop_bwd(rows, c, z.data_ptr(), _p(bias), p, site, _p(seed_in), dh.data_ptr(), dz.data_ptr(),
                                                        _p(acc), _native._stream(z)), "tail_relu_drop_bwd")
        return dz, (None if acc is None else acc[:]), None, None, None, None


def relu_dropout(z, bias, p, site, seed_in, grads):
    """dropout(relu(z + bias)) on rows (R, C), C % 4 == 0."""
    return _ReluDrop.apply(z, bias, p, site, seed_in, grads)


class _PoseHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, w, bias, xyz1, R, t, scale, acc):
        B, _, J = xyz1.shape
        C = h.shape[1]
        kp_hand = torch.empty((B, 3, J), dtype=_f32, device=h.device)
        kp_cam = torch.empty((B, J, 3), dtype=_f32, device=h.device)
        with torch.cuda.device(h.device):
            _native._check(_lib.pn2x_tail_pose_head_fwd(B, J, C, h.data_ptr(), w.data_ptr(), bias.data_ptr(), xyz1.data_ptr(), R.data_ptr(),
                                                        t.data_ptr(), scale.data_ptr(), 1 if scale.numel() > 1 else 0, kp_hand.data_ptr(),
                                                        kp_cam.data_ptr(),
                                                        _native._stream(h)), "tail_pose_head_fwd")
        ctx.save_for_backward(h, w, R, scale)
        ctx.acc, ctx.J = acc, J
        ctx.set_materialize_grads(False)  # (backward handles a missing g_hand / g_cam: no zero-fill launch for it)
        return kp_hand, kp_cam

    @staticmethod
    def backward(ctx, g_hand, g_cam):
        h, w, R, scale = ctx.saved_tensors
        rows, C = h.shape
        (acc,) = _fresh(ctx, (ctx.acc,))
        g_hand = None if g_hand is None else g_hand.contiguous()
        g_cam = None if g_cam is None else g_cam.contiguous()
        dh = torch.empty_like(h)
        if g_hand is None and g_cam is None:
            return torch.zeros_like(h), None, None, None, None, None, None, None
        with torch.cuda.device(h.device):
            _native._check(_lib.pn2x_tail_pose_head_bwd(rows // ctx.J, ctx.J, C, h.data_ptr(), w.data_ptr(), _p(g_hand), _p(g_cam), R.data_ptr(),
                                                        scale.data_ptr(), 1 if scale.numel() > 1 else 0, dh.data_ptr(), acc.data_ptr(),
                                                        acc.data_ptr() + 4 * 3 * C,
                                                        _native._stream(h)), "tail_pose_head_bwd")
        return dh, acc[:3 * C].view(3, C), acc[3 * C:3 * C + 3], None, None, None, None, None


def pose_head(h, w, bias, xyz1, R, t, scale, grads):
    """(kp_hand (B,3,J), kp_cam (B,J,3)) = (h w^T + bias + xyz1, scale R kp_hand + t): h (B*J, C) rows, w (3, C), bias (3,), xyz1
    (B,3,J), R (B,3,3), t (B,3[,1]), scale (B,).  Gradients to h, w, bias."""
    B = xyz1.shape[0]
    scale = scale.reshape(-1)  # (B,) or (1,): one value for every cloud
    args = [a.contiguous().float() for a in (h, w, bias, xyz1, R.reshape(B, 3, 3), t.reshape(B, 3), scale)]
    return _PoseHead.apply(*args, grads.take(3 * w.shape[1] + 3))
