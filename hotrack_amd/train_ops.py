"""Training-mode operators on point-major activations (include/pn2_ext.h, csrc/train_ops.hip) as autograd Functions.

  bn_relu(y, conv, bn, ws)          train-mode BatchNorm (+ ReLU) of the pre-activations y (R, C): batch statistics, running
                                    statistics update, saved mean / invstd; backward = one reduce + one apply kernel
  sa_layer1(...)                    layer-1 pre-activations of a set-abstraction scale from the linear split
                                    a1f[idx] + wx.(xyz[idx] - centre) + cadd, without materialising the grouped input;
                                    backward = scatter-add of rows (+ one small GEMM for d(wx), one sum for d(cadd))
  interpolate_rows(points, idx, w)  three-NN interpolation on rows; backward = weighted scatter-add of rows

GPU tensors only (no CPU path).  `Workspace` hands out zeroed fp64 accumulator slices: ONE fill launch per training step
instead of one per BatchNorm call and direction.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import pointnet2_hip as _native

_lib = _native._lib
_vp, _ci, _cl, _cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
_lib.pn2x_bn_stats.argtypes = [_cl, _ci, _vp, _ci, _vp, _vp]
_lib.pn2x_bn_relu_apply.argtypes = [_cl, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _cf, _cf, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _vp]
_lib.pn2x_bn_relu_bwd.argtypes = [_cl, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _ci, _vp, _vp, _ci, _vp, _vp, _vp, _vp]
_lib.pn2x_scatter_add_rows.argtypes = [_ci, _ci, _ci, _ci, _vp, _ci, _vp, _vp, _ci, _vp]
_lib.pn2x_three_interpolate_pm_grad.argtypes = [_ci, _ci, _ci, _ci, _vp, _ci, _vp, _vp, _vp, _ci, _vp]
_lib.pn2x_sa_layer1.argtypes = [_ci] * 5 + [_vp, _ci, _vp, _vp, _vp, _vp, _ci, _vp, _vp, _vp, _vp]
_lib.pn2x_sa_layer1_ld.argtypes = [_ci] * 5 + [_vp, _ci, _vp, _vp, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp]
_lib.pn2x_sa_layer1_ld.restype = _ci
_lib.pn2x_sa_layer1_stats.argtypes = [_ci] * 5 + [_vp, _ci, _vp, _vp, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _vp]
_lib.pn2x_sa_layer1_stats.restype = _ci
_lib.pn2x_bn_sums_doubles.argtypes = [_ci]
_lib.pn2x_bn_sums_doubles.restype = _ci
_SCALE = [_ci, _ci, _vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp]  # k, c1, a1f, a1f_ld, wx, wx_ld, cadd, cadd_ld, idx, out, rel_out, sums
_lib.pn2x_sa_layer1_stats_pair.argtypes = [_ci, _ci, _ci, _vp, _vp] + _SCALE + _SCALE + [_vp]
_lib.pn2x_sa_layer1_stats_pair.restype = _ci
_lib.pn2x_rows_segment_sum_pair.argtypes = [_ci, _ci] + [_ci, _ci, _vp, _ci, _vp, _vp, _vp, _ci] * 2 + [_ci, _vp]
_lib.pn2x_rows_segment_sum_pair.restype = _ci
PAIR_SCALES = True  # the two scales of a module: one launch where a pair kernel exists (False: one launch per scale; tests compare)
_lib.pn2x_rows_outer3.argtypes = [_cl, _ci, _vp, _ci, _vp, _vp, _vp, _cl, _vp]
_lib.pn2x_rows_outer3.restype = _ci
_lib.pn2x_rows_outer3_scratch_floats.argtypes = [_cl, _ci]
_lib.pn2x_rows_outer3_scratch_floats.restype = _cl
for _n in ("pn2x_bn_stats", "pn2x_bn_relu_apply", "pn2x_bn_relu_bwd", "pn2x_scatter_add_rows", "pn2x_three_interpolate_pm_grad",
           "pn2x_sa_layer1"):
    getattr(_lib, _n).restype = _ci

_lib.pn2x_inverse_index.argtypes = [_ci, _ci, _ci, _vp, _vp, _vp, _vp]
_lib.pn2x_inverse_index.restype = _ci
_lib.pn2x_rows_segment_sum.argtypes = [_ci, _ci, _ci, _ci, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _ci, _ci, _vp]
_lib.pn2x_rows_segment_sum.restype = _ci
INVERSE_MAX_ROWS = 16127  # pn2x_inverse_index keeps n_dst + 1 + 256 counters in 64 KiB of LDS
_lib.pn2x_bn_relu_max.argtypes = [_cl, _ci, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _cf, _cf, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
_lib.pn2x_bn_relu_max.restype = _ci
_lib.pn2x_bn_relu_max_bwd.argtypes = [_cl, _ci, _ci, _vp, _vp, _vp, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _vp, _vp, _vp, _vp]
_lib.pn2x_bn_relu_max_bwd.restype = _ci
_lib.pn2x_bn_sums_doubles.argtypes = [_ci]
_lib.pn2x_bn_sums_doubles.restype = _ci
_f32 = torch.float32


def _p(t):
    return None if t is None else t.data_ptr()


def _rows2d(t: torch.Tensor, name: str):
    if not t.is_cuda or t.dtype != _f32 or t.dim() != 2 or t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
        raise ValueError(f"{name}: expected a 2-D float32 GPU tensor with contiguous rows (stride multiple of 4, 16-byte aligned), "
                         f"got shape {tuple(t.shape)} strides {t.stride()} on {t.device}")
    return t.data_ptr(), t.stride(0)


class Workspace:
    """Zeroed fp64 accumulators for the BatchNorm reductions of one training step: reset() = one fill launch, take(n)
    hands out consecutive slices (fixed capacity: take() raises when it is exhausted).

    A slice handed out for a BACKWARD reduction is only valid for the first backward of the forward that took it, and only
    while no later forward has reset the workspace: `generation` counts the resets, `backward_slice()` returns the slice when
    it is still this forward's and has not been used, and a freshly zeroed tensor otherwise (gradient accumulation over two
    forwards, retain_graph / a second autograd.grad through the same graph) -- never sums into stale or already-used
    accumulators."""

    def __init__(self, device, capacity: int = 1 << 19):
        self.buf = torch.zeros(capacity, dtype=torch.float64, device=device)
        self.used = 0
        self.generation = 0

    def reset(self):
        self.buf.zero_()
        self.used = 0
        self.generation += 1

    def backward_slice(self, ctx) -> torch.Tensor:
        """The zeroed accumulator of ctx's backward (ctx.ws, ctx.ws_b, ctx.ws_gen set in forward)."""
        if ctx.ws_gen == self.generation and not getattr(ctx, "ws_used", False):
            ctx.ws_used = True
            return ctx.ws_b
        return torch.zeros(ctx.ws_b.numel(), dtype=torch.float64, device=ctx.ws_b.device)

    def take(self, n: int) -> torch.Tensor:
        if self.used + n > self.buf.numel():
            raise RuntimeError("train_ops.Workspace exhausted: construct it with a larger capacity")
        out = self.buf[self.used:self.used + n]
        self.used += n
        return out


class _BnRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, conv_bias, running_mean, running_var, nbt, eps, momentum, relu, ws_f, ws_b, ws):
        py, ldy = _rows2d(y, "y")
        R, C = y.shape
        h = torch.empty((R, C), dtype=_f32, device=y.device)
        saved = torch.empty((2, C), dtype=_f32, device=y.device)
        st = _native._stream(y)
        with torch.cuda.device(y.device):
            _native._check(_lib.pn2x_bn_stats(R, C, py, ldy, ws_f.data_ptr(), st), "bn_stats")
            _native._check(_lib.pn2x_bn_relu_apply(R, C, py, ldy, ws_f.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _p(conv_bias),
                                                   float(eps), float(momentum), _p(running_mean), _p(running_var), _p(nbt),
                                                   saved[0].data_ptr(), saved[1].data_ptr(), h.data_ptr(), C, 1 if relu else 0, st),
                           "bn_relu_apply")
        ctx.save_for_backward(y, gamma, beta, saved)
        ctx.ws_b, ctx.relu, ctx.has_bias = ws_b, relu, conv_bias is not None
        ctx.ws, ctx.ws_gen = ws, ws.generation
        return h

    @staticmethod
    def backward(ctx, dh):
        y, gamma, beta, saved = ctx.saved_tensors
        dh = dh.contiguous()
        R, C = y.shape
        dy = torch.empty((R, C), dtype=_f32, device=y.device)
        dpar = torch.empty((3, C), dtype=_f32, device=y.device)
        sums = ctx.ws.backward_slice(ctx)
        with torch.cuda.device(y.device):
            _native._check(_lib.pn2x_bn_relu_bwd(R, C, dh.data_ptr(), C, y.data_ptr(), y.stride(0), saved[0].data_ptr(), saved[1].data_ptr(),
                                                 gamma.data_ptr(), beta.data_ptr(), 1 if ctx.relu else 0, sums.data_ptr(), dy.data_ptr(), C,
                                                 dpar[0].data_ptr(), dpar[1].data_ptr(), dpar[2].data_ptr(), _native._stream(y)), "bn_relu_bwd")
        return dy, dpar[0], dpar[1], (dpar[2] if ctx.has_bias else None), None, None, None, None, None, None, None, None, None


def bn_relu(y: torch.Tensor, bn: torch.nn.Module, ws: Workspace, conv_bias: torch.Tensor = None, relu: bool = True) -> torch.Tensor:
    """relu(BatchNorm_train(y + conv_bias)) for y (R, C) point-major; `bn` supplies weight / bias / running statistics /
    momentum / eps (torch.nn.BatchNorm1d|2d semantics incl. the running-statistics update).  conv_bias only shifts the
    batch mean, so y is taken WITHOUT it; its gradient is identically zero and is returned as zeros."""
    C = y.shape[1]
    track = bn.track_running_stats and bn.running_mean is not None
    return _BnRelu.apply(y, bn.weight, bn.bias, conv_bias, bn.running_mean if track else None, bn.running_var if track else None,
                         bn.num_batches_tracked if track else None, bn.eps, bn.momentum if bn.momentum is not None else 0.1, relu,
                         ws.take(_lib.pn2x_bn_sums_doubles(C)), ws.take(_lib.pn2x_bn_sums_doubles(C)), ws)


class _BnReluMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, K, gamma, beta, conv_bias, running_mean, running_var, nbt, eps, momentum, ws_f, ws_b, ws):
        py, ldy = _rows2d(y, "y")
        R, C = y.shape
        G = R // K
        out = torch.empty((G, C), dtype=_f32, device=y.device)
        arg = torch.empty((G, C), dtype=torch.int32, device=y.device)
        saved = torch.empty((2, C), dtype=_f32, device=y.device)
        st = _native._stream(y)
        with torch.cuda.device(y.device):
            _native._check(_lib.pn2x_bn_stats(R, C, py, ldy, ws_f.data_ptr(), st), "bn_stats")
            _native._check(_lib.pn2x_bn_relu_max(G, K, C, py, ldy, ws_f.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _p(conv_bias), float(eps),
                                                 float(momentum), _p(running_mean), _p(running_var), _p(nbt), saved[0].data_ptr(),
                                                 saved[1].data_ptr(), out.data_ptr(), arg.data_ptr(), st), "bn_relu_max")
        ctx.save_for_backward(y, gamma, beta, saved, arg)
        ctx.ws_b, ctx.K, ctx.has_bias = ws_b, K, conv_bias is not None
        ctx.ws, ctx.ws_gen = ws, ws.generation
        return out

    @staticmethod
    def backward(ctx, dout):
        y, gamma, beta, saved, arg = ctx.saved_tensors
        dout = dout.contiguous()
        R, C = y.shape
        dy = torch.empty((R, C), dtype=_f32, device=y.device)
        dpar = torch.empty((3, C), dtype=_f32, device=y.device)
        sums = ctx.ws.backward_slice(ctx)
        with torch.cuda.device(y.device):
            _native._check(_lib.pn2x_bn_relu_max_bwd(R // ctx.K, ctx.K, C, dout.data_ptr(), arg.data_ptr(), y.data_ptr(), y.stride(0),
                                                     saved[0].data_ptr(), saved[1].data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                     sums.data_ptr(), dy.data_ptr(), C, dpar[0].data_ptr(), dpar[1].data_ptr(),
                                                     dpar[2].data_ptr(), _native._stream(y)), "bn_relu_max_bwd")
        return dy, None, dpar[0], dpar[1], (dpar[2] if ctx.has_bias else None), None, None, None, None, None, None, None, None


def bn_relu_max(y: torch.Tensor, K: int, bn: torch.nn.Module, ws: Workspace, conv_bias: torch.Tensor = None) -> torch.Tensor:
    """max over each group of K consecutive rows of relu(BatchNorm_train(y + conv_bias)): y (G*K, C) -> (G, C).  The last layer
    of a set-abstraction scale + its neighbourhood reduction without writing the (G*K, C) activations; same statistics /
    running-statistics semantics as bn_relu."""
    R, C = y.shape
    if R % K:
        raise ValueError("bn_relu_max: rows must be a multiple of K")
    track = bn.track_running_stats and bn.running_mean is not None
    n = _lib.pn2x_bn_sums_doubles(C)
    return _BnReluMax.apply(y, K, bn.weight, bn.bias, conv_bias, bn.running_mean if track else None, bn.running_var if track else None,
                            bn.num_batches_tracked if track else None, bn.eps, bn.momentum if bn.momentum is not None else 0.1,
                            ws.take(n), ws.take(n), ws)


def inverse_index(idx: torch.Tensor, n_dst: int):
    """idx (B, L) int32 with values in [0, n_dst) -> (offsets (B, n_dst+1), order (B, L)) int32: the positions of every target."""
    B, L = idx.shape
    offsets = torch.empty((B, n_dst + 1), dtype=torch.int32, device=idx.device)
    order = torch.empty((B, L), dtype=torch.int32, device=idx.device)
    with torch.cuda.device(idx.device):
        _native._check(_lib.pn2x_inverse_index(B, n_dst, L, _native._ptr(idx, "idx", torch.int32, B * L), offsets.data_ptr(), order.data_ptr(),
                                               _native._stream(idx)), "inverse_index")
    return offsets, order


def rows_segment_sum(dout: torch.Tensor, inv, n_dst: int, din: torch.Tensor, weight: torch.Tensor = None, accumulate: bool = False):
    """din[b, i, :] (+)= sum over the entries e of target i of (weight[b, e] *) dout[b, e (// 3), :]  -- no atomics, every row of
    din written once.  dout (B,M,C) rows (contiguous, or a column block of a wider gradient: see _row_block);
    inv = inverse_index(idx.view(B, -1), n_dst); din (B,n_dst,>=C) rows."""
    B, M, C = dout.shape
    offsets, order = inv
    t = 1 if weight is None else 3
    dout = _row_block(dout)
    with torch.cuda.device(dout.device):
        _native._check(_lib.pn2x_rows_segment_sum(B, n_dst, M, t, C, dout.data_ptr(), dout.stride(1) if M > 1 else max(dout.stride(1), C), offsets.data_ptr(),
                                                  order.data_ptr(), _p(weight), din.data_ptr(), din.stride(1), 1 if accumulate else 0,
                                                  _native._stream(dout)), "rows_segment_sum")
    return din


def _row_block(t: torch.Tensor) -> torch.Tensor:
    """t (B,M,C) as the kernels with a row-stride argument can read it: itself when its rows are 16-byte aligned, unit-stride and
    uniformly spaced across the batch (a column block of a wider tensor -- the half of a concatenation's gradient -- costs no copy
    launch then), a contiguous copy otherwise."""
    B, M, C = t.shape
    if (t.dtype == _f32 and t.stride(2) == 1 and t.stride(1) % 4 == 0 and t.stride(1) >= C and t.data_ptr() % 16 == 0
            and (B == 1 or t.stride(0) == M * t.stride(1))):
        return t
    return t.contiguous()


def scatter_add_rows(dout: torch.Tensor, idx: torch.Tensor, din: torch.Tensor) -> torch.Tensor:
    """din[b, idx[b,j], :] += dout[b, j, :]; dout (B,M,C) contiguous, idx (B,M) int32, din (B,N,>=C) rows (may be a column block)."""
    B, M, C = dout.shape
    N = din.shape[1]
    with torch.cuda.device(dout.device):
        _native._check(_lib.pn2x_scatter_add_rows(B, N, M, C, _native._ptr(dout, "dout", _f32, B * M * C), C,
                                                  _native._ptr(idx, "idx", torch.int32, B * M), din.data_ptr(), din.stride(1),
                                                  _native._stream(dout)), "scatter_add_rows")
    return din


_group_list_cache = {}


def _group_lists(B: int, G: int, K: int, dev):
    """(offsets (B, G + 1), order (B, G K)) of the trivial lists "target g <- rows g K ... g K + K - 1": what inverse_index would
    return for idx[b, r] = r // K.  Constant per shape: built once, eagerly (never inside a graph capture) and kept."""
    key = (B, G, K, dev)
    t = _group_list_cache.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("train_ops: group lists of a new shape requested inside a graph capture (run one eager step first)")
        off = (torch.arange(G + 1, dtype=torch.int32, device=dev) * K).unsqueeze(0).expand(B, -1).contiguous()
        order = torch.arange(G * K, dtype=torch.int32, device=dev).unsqueeze(0).expand(B, -1).contiguous()
        t = _group_list_cache[key] = (off, order)
    return t


class _SaLayer1(torch.autograd.Function):
    """One module call = all its scales: a1f (B,N,sum C1) | None, cadd (B,S,sum C1) | None, then per scale (idx_i, wx_i)."""

    @staticmethod
    def forward(ctx, a1f, cadd, xyz, cxyz, n_scales, aux, *rest):
        # aux: dict | None -- a side channel to the stacks that consume the y1s (train_stack.mlp_stack(aux=(aux, i))): this forward
        # leaves the relative coordinates there (aux["rel"][i]); a stack whose first-layer backward has dY_1 in registers anyway
        # leaves d(wx_i) there (aux["dwx"][i]), which this backward then takes instead of a pass of its own over dY_1.
        # rest: idx_i (n_scales), wx_i (n_scales), then optionally the inverted neighbour lists (offsets_i, order_i) per scale
        idxs, wxs, invs = rest[:n_scales], rest[n_scales:2 * n_scales], rest[2 * n_scales:]
        B, N, _ = xyz.shape
        S = cxyz.shape[1]
        outs, rels, sums = [], [], []
        col = 0
        st = _native._stream(xyz)
        ws = aux.get("ws") if aux is not None else None  # the consumer stacks' workspace: BatchNorm statistics taken on the way
        pending = []  # per scale (common args, per-scale args, sums slice): launched alone, or two scales as one pair launch
        for idx, wx in zip(idxs, wxs):
            K, C1 = idx.shape[2], wx.shape[0]
            out = torch.empty((B, S * K, C1), dtype=_f32, device=xyz.device)
            rel = torch.empty((B, S * K, 3), dtype=_f32, device=xyz.device)
            if wx.dim() != 2 or wx.shape[1] != 3 or wx.stride(1) != 1 or wx.dtype != _f32:
                wx = wx.contiguous().float()
            sm = ws.take(_lib.pn2x_bn_sums_doubles(C1)) if ws is not None else None
            # (the (C1, 3) block is read in place: a column block of the layer's weight)
            pending.append((K, C1, None if a1f is None else a1f.data_ptr() + 4 * col, 0 if a1f is None else a1f.stride(1),
                            wx.data_ptr(), wx.stride(0) if C1 > 1 else 3, None if cadd is None else cadd.data_ptr() + 4 * col,
                            0 if cadd is None else cadd.stride(1), _native._ptr(idx, "idx", torch.int32, B * S * K), out.data_ptr(),
                            rel.data_ptr(), sm, wx))
            outs.append(out)
            rels.append(rel)
            sums.append(sm)
            col += C1
        with torch.cuda.device(xyz.device):
            if PAIR_SCALES and len(pending) == 2 and all(p[11] is not None for p in pending):
                pa, pb = pending
                _native._check(_lib.pn2x_sa_layer1_stats_pair(B, N, S, xyz.data_ptr(), cxyz.data_ptr(), *pa[:11], pa[11].data_ptr(),
                                                              *pb[:11], pb[11].data_ptr(), st), "sa_layer1_stats_pair")
            else:
                for (K, C1, pa1f, lda, pwx, ldw, pcadd, ldc, pidx, pout, prel, sm, _wx) in pending:
                    args = (B, N, S, K, C1, pa1f, lda, xyz.data_ptr(), cxyz.data_ptr(), pwx, ldw, pcadd, ldc, pidx, pout, prel)
                    if sm is not None:
                        _native._check(_lib.pn2x_sa_layer1_stats(*args, sm.data_ptr(), st), "sa_layer1_stats")
                    else:
                        _native._check(_lib.pn2x_sa_layer1_ld(*args, st), "sa_layer1")
        ctx.aux = aux
        if aux is not None:
            aux["rel"], aux["dwx"] = list(rels), {}
            aux["sums"] = {i: sm for i, sm in enumerate(sums) if sm is not None}  # layer-1 statistics for mlp_stack(aux=(aux, i))
        ctx.save_for_backward(*idxs, *rels, *invs)
        ctx.meta = (n_scales, None if a1f is None else tuple(a1f.shape), None if cadd is None else tuple(cadd.shape), S)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        n, a1f_shape, cadd_shape, S = ctx.meta
        idxs, rels, invs = ctx.saved_tensors[:n], ctx.saved_tensors[n:2 * n], ctx.saved_tensors[2 * n:]
        dev = douts[0].device
        fits = a1f_shape is not None and a1f_shape[1] <= INVERSE_MAX_ROWS
        d_a1f = None
        share = ctx.aux.get("a1f_share") if (ctx.aux is not None and a1f_shape is not None and fits) else None
        if share is not None:  # this module's column block of the buffer all modules of one per-point product share (linear_dw)
            from .linear_dw import share_grad_block
            blk = share_grad_block(share[0], share[1], dev)
            if blk is not None and blk.shape[0] == a1f_shape[0] * a1f_shape[1] and blk.shape[1] == a1f_shape[2]:
                d_a1f = blk.view(a1f_shape)  # (row stride = the buffer's width: every kernel below takes one)
        if d_a1f is None and a1f_shape is not None:
            d_a1f = (torch.empty if fits else torch.zeros)(a1f_shape, dtype=_f32, device=dev)
        d_cadd = torch.empty(cadd_shape, dtype=_f32, device=dev) if cadd_shape is not None else None
        d_wx = []
        col = 0
        douts = [dy.contiguous() for dy in douts]
        paired = False
        if PAIR_SCALES and n == 2 and d_a1f is not None and fits and invs:
            # the two scales scatter into two column blocks of the same rows: one launch (pn2x_rows_segment_sum_pair)
            (da, db_), Bq = douts, douts[0].shape[0]
            ca, cb = da.shape[2], db_.shape[2]
            blk_a, blk_b = d_a1f[:, :, :ca], d_a1f[:, :, ca:ca + cb]
            with torch.cuda.device(dev):
                _native._check(_lib.pn2x_rows_segment_sum_pair(
                    Bq, a1f_shape[1], da.shape[1], ca, da.data_ptr(), da.stride(1), invs[0].data_ptr(), invs[1].data_ptr(), blk_a.data_ptr(),
                    blk_a.stride(1), db_.shape[1], cb, db_.data_ptr(), db_.stride(1), invs[2].data_ptr(), invs[3].data_ptr(), blk_b.data_ptr(),
                    blk_b.stride(1), 0, _native._stream(da)), "rows_segment_sum_pair")
            paired = True
        # the centre term's gradient = per centroid the sum of its K consecutive rows: for two scales ONE segment-sum launch over
        # trivial lists (group g <- rows g K ... g K + K - 1) instead of two torch reductions
        cadd_done = False
        if d_cadd is not None and PAIR_SCALES and n == 2 and S <= INVERSE_MAX_ROWS:
            (da, db_), Bq = douts, douts[0].shape[0]
            ca, cb = da.shape[2], db_.shape[2]
            ga, gb = _group_lists(Bq, S, da.shape[1] // S, dev), _group_lists(Bq, S, db_.shape[1] // S, dev)
            blk_a, blk_b = d_cadd[:, :, :ca], d_cadd[:, :, ca:ca + cb]
            with torch.cuda.device(dev):
                _native._check(_lib.pn2x_rows_segment_sum_pair(
                    Bq, S, da.shape[1], ca, da.data_ptr(), da.stride(1), ga[0].data_ptr(), ga[1].data_ptr(), blk_a.data_ptr(), blk_a.stride(1),
                    db_.shape[1], cb, db_.data_ptr(), db_.stride(1), gb[0].data_ptr(), gb[1].data_ptr(), blk_b.data_ptr(), blk_b.stride(1), 0,
                    _native._stream(da)), "rows_segment_sum_pair")
            cadd_done = True
        for i, (dy, idx, rel) in enumerate(zip(douts, idxs, rels)):
            B, SK, C1 = dy.shape
            if paired:
                pass
            elif d_a1f is not None and fits:  # owner-computes segment sum over the inverted neighbour lists: no atomics, no pre-zeroing
                inv = (invs[2 * i], invs[2 * i + 1]) if invs else inverse_index(idx.view(B, SK), a1f_shape[1])
                rows_segment_sum(dy, inv, a1f_shape[1], d_a1f[:, :, col:col + C1])
            elif d_a1f is not None:
                scatter_add_rows(dy, idx.view(B, SK), d_a1f[:, :, col:col + C1])
            if d_cadd is not None and not cadd_done:
                torch.sum(dy.view(B, S, SK // S, C1), dim=2, out=d_cadd[:, :, col:col + C1])
            ready = ctx.aux["dwx"].pop(i, None) if ctx.aux is not None else None
            if ready is not None:  # formed by the consumer's first-layer backward (pn2x_bn_bwd_apply_rel)
                d_wx.append(ready)
            elif C1 % 4 == 0 and 256 % (C1 // 4) == 0 and rel.is_contiguous():  # one pass over dy (csrc/train_ops.hip: rows_outer3)
                dwx = torch.empty((C1, 3), dtype=_f32, device=dev)
                scratch = torch.empty(int(_lib.pn2x_rows_outer3_scratch_floats(B * SK, C1)), dtype=_f32, device=dev)
                with torch.cuda.device(dev):
                    _native._check(_lib.pn2x_rows_outer3(B * SK, C1, dy.data_ptr(), C1, rel.data_ptr(), dwx.data_ptr(),
                                                         scratch.data_ptr(), scratch.numel(), _native._stream(dy)), "rows_outer3")
                d_wx.append(dwx)
            else:
                d_wx.append(torch.mm(dy.view(B * SK, C1).t(), rel.view(B * SK, 3)))
            col += C1
        return (d_a1f, d_cadd, None, None, None, None, *([None] * n), *d_wx, *([None] * len(invs)))


def sa_layer1(a1f, cadd, xyz, cxyz, idxs, wxs, invs=None, aux=None, ws=None):
    """Layer-1 pre-activations of every scale of one SA module: list of (B, S*K_i, C1_i).
    a1f (B,N,sum C1) per-point feature terms [scale 0 | scale 1 ...] or None; cadd (B,S,sum C1) per-centroid terms or None;
    xyz (B,N,3), cxyz (B,S,3) (no gradient); idxs[i] (B,S,K_i) int32; wxs[i] (C1_i, 3).
    invs[i] = inverse_index(idxs[i].view(B, -1), N), computed by a caller that feeds the same neighbour lists to several modules
    (the backward inverts them itself otherwise).  aux: an empty dict shared with train_stack.mlp_stack(aux=(aux, i)) -- see
    _SaLayer1.forward.  ws (with aux): the Workspace of the stacks that will consume the outputs -- the kernel then also accumulates
    each output's BatchNorm statistics into a slice of it (aux["sums"][i]) and the stack skips its own statistics pass."""
    if aux is not None and ws is not None:
        aux["ws"] = ws
    extra = [t for inv in invs for t in inv] if invs else []
    return list(_SaLayer1.apply(a1f, cadd, xyz, cxyz, len(idxs), aux, *idxs, *wxs, *extra))


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, idx, inv_off, inv_order):
        from . import ext
        ctx.save_for_backward(inv_off, inv_order)
        ctx.n = src.shape[1]
        return ext.gather_rows(src, idx)

    @staticmethod
    def backward(ctx, dout):
        B, M, C = dout.shape
        dsrc = torch.empty((B, ctx.n, C), dtype=_f32, device=dout.device)
        rows_segment_sum(dout, tuple(ctx.saved_tensors), ctx.n, dsrc)  # every row written once: no zero fill, no atomics
        return dsrc, None, None, None


def gather_rows(src: torch.Tensor, idx: torch.Tensor, inv) -> torch.Tensor:
    """src (B,N,C) contiguous, idx (B,M) int32, inv = inverse_index(idx, N) -> src[b, idx[b, j], :] (B,M,C); the gradient is the
    owner-computes segment sum over the inverted lists (torch.index_select's backward: a zero fill + an atomic index_add)."""
    return _GatherRows.apply(src.contiguous(), idx, inv[0], inv[1])


class _InterpRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx, weight, *inv):
        from . import ext
        B, M, C = points.shape
        n = idx.shape[1]
        out = torch.empty((B, n, C), dtype=_f32, device=points.device)
        ext.three_interpolate_pm(points, idx, weight, out)
        ctx.save_for_backward(idx, weight, *inv)
        ctx.m = M
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, weight, *inv = ctx.saved_tensors
        B, n, C = dout.shape
        none = (None,) * len(inv)
        if ctx.m > INVERSE_MAX_ROWS:
            dout = dout.contiguous()
        if ctx.m <= INVERSE_MAX_ROWS:
            dp = torch.empty((B, ctx.m, C), dtype=_f32, device=dout.device)
            rows_segment_sum(dout, tuple(inv) if inv else inverse_index(idx.view(B, n * 3), ctx.m), ctx.m, dp, weight=weight)
            return (dp, None, None, *none)
        dp = torch.zeros((B, ctx.m, C), dtype=_f32, device=dout.device)
        with torch.cuda.device(dout.device):
            _native._check(_lib.pn2x_three_interpolate_pm_grad(B, C, ctx.m, n, dout.data_ptr(), C, idx.data_ptr(), weight.data_ptr(),
                                                               dp.data_ptr(), C, _native._stream(dout)), "three_interpolate_pm_grad")
        return (dp, None, None, *none)


def interpolate_rows(points: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor, inv=None) -> torch.Tensor:
    """points (B,M,C) contiguous, idx / weight (B,n,3) -> (B,n,C); gradient to points only (reference pointnet2_utils.py:190).
    inv = inverse_index(idx.view(B, 3 n), M) when the caller has it already (it depends on the geometry only: a training loop
    computes it for the NEXT batch beside this batch's dense work); the backward inverts the lists itself otherwise."""
    return _InterpRows.apply(points.contiguous(), idx, weight, *(inv if inv is not None else ()))
