"""A whole train-mode [Conv 1x1 + BatchNorm + ReLU] stack as ONE autograd Function on the fused BatchNorm GEMMs of
csrc/train_gemm.hip (include/pn2_ext.h: pn2x_tg_fwd / pn2x_tg_dgrad / pn2x_tg_wgrad).

    mlp_stack(y1, layers, ws, max_over=K)

`y1` (R, C1) are the PRE-activations of the stack's first layer (a gather-assembled layer-1 of a set-abstraction scale, or a
library GEMM over a concatenated input); layers[0] carries only that layer's BatchNorm, layers[i >= 1] a convolution weight
(C_i, C_{i-1}) and its BatchNorm.  Forward: statistics of y1 (pn2x_bn_stats), then per further layer ONE kernel that
normalises + rectifies its input on load, multiplies on the fp32 matrix cores and accumulates the statistics of its output;
the top of the stack is the streaming kernel pair of train_ops (materialised output, or max over every K consecutive rows).
Backward: one reduction over the top layer, then per layer a weight-gradient and a data-gradient kernel that compute the
pre-activation gradient dY on load; the first layer's dY is materialised by the streaming apply kernel and returned as the
gradient of y1.  Neither the normalised activations nor any dY of a hidden layer is ever written to memory.

Semantics are those of torch.nn.BatchNorm1d/2d in training mode (batch statistics, running-statistics update with the
unbiased variance, num_batches_tracked) -- reference composition: pointnet_utils.py:399-403, :460-462, :504-506, :577-581.
GPU tensors only (no CPU path).
"""
from __future__ import annotations

import ctypes
import os as _os

import torch

from . import pointnet2_hip as _native
from . import train_ops as _t

_lib = _native._lib
_vp, _ci, _cl, _cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
_lib.pn2x_tg_supported.argtypes = [_ci, _ci]
_lib.pn2x_tg_supported.restype = _ci
_lib.pn2x_tg_fwd.argtypes = [_cl, _ci, _ci, _vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _cf, _cf, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
_lib.pn2x_tg_fwd.restype = _ci
_lib.pn2x_tg_fwd2.argtypes = _lib.pn2x_tg_fwd.argtypes
_lib.pn2x_tg_fwd2.restype = _ci
_lib.pn2x_tg_fwd2_supported.argtypes = [_ci, _ci]
_lib.pn2x_tg_fwd2_supported.restype = _ci
FWD2 = True  # 64- / 128-channel inputs: the W-resident forward (csrc/train_fwd.hip); False: pn2x_tg_fwd for every shape (tests)
_DY = [_ci, _vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _vp]  # gmode, g, ldg, arg, kmax, yi, ldyi, mean, invstd, gamma, beta, sums_bwd
_lib.pn2x_tg_dgrad.argtypes = [_cl, _ci, _ci] + _DY + [_vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _vp, _ci, _vp, _vp]
_lib.pn2x_tg_dgrad.restype = _ci
_lib.pn2x_tg_wgrad_partial_floats.argtypes = [_cl, _ci, _ci]
_lib.pn2x_tg_wgrad_partial_floats.restype = _cl
_lib.pn2x_tg_wgrad.argtypes = [_cl, _ci, _ci] + _DY + [_vp, _ci, _vp, _vp, _vp, _vp, _vp, _cl, _vp, _vp, _vp, _vp, _vp]
_lib.pn2x_tg_wgrad.restype = _ci
_lib.pn2x_tg_wgrad2.argtypes = _lib.pn2x_tg_wgrad.argtypes[:-1] + [ctypes.POINTER(_ci), _vp]
_lib.pn2x_tg_wgrad2.restype = _ci
_lib.pn2x_tg_reduce_multi.argtypes = [_ci, ctypes.POINTER(_vp), ctypes.POINTER(_ci), ctypes.POINTER(_ci), ctypes.POINTER(_vp),
                                      ctypes.POINTER(_vp), ctypes.POINTER(_ci), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                      ctypes.POINTER(_vp), _vp]
_lib.pn2x_tg_reduce_multi.restype = _ci
_lib.pn2x_tg_bwd_supported.argtypes = [_ci, _ci]
_lib.pn2x_tg_bwd_supported.restype = _ci
_lib.pn2x_tg_bwd_partials.argtypes = [_cl, _ci, _ci]
_lib.pn2x_tg_bwd_partials.restype = _ci
_lib.pn2x_tg_bwd.argtypes = [_cl, _ci, _ci, _ci, _vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _vp, _ci, _vp, _vp, _vp,
                             _vp, _vp, _ci, _vp, _vp, _cl, _vp, _vp]
_lib.pn2x_tg_bwd_slice.argtypes = [_cl, _ci, _ci, _ci, _vp, _ci, _vp, _ci, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp,
                                   _vp, _vp, _vp, _ci, _vp, _vp, _cl, _vp, _vp, _ci, _ci, _vp]
_lib.pn2x_tg_bwd_slice.restype = _ci
_lib.pn2x_tg_fwd2_pair_supported.argtypes = [_ci, _ci]
_lib.pn2x_tg_fwd2_pair_supported.restype = _ci
_lib.pn2x_tg_bwd_pair_supported.argtypes = [_ci, _ci]
_lib.pn2x_tg_bwd_pair_supported.restype = _ci
_lib.pn2x_tg_fwd2_pair.argtypes = _lib.pn2x_tg_fwd.argtypes[:-1] + [_cl] + _lib.pn2x_tg_fwd.argtypes[3:-1] + [_vp]  # (k, n once)
_lib.pn2x_tg_fwd2_pair.restype = _ci
_lib.pn2x_tg_bwd_slice_pair.argtypes = _lib.pn2x_tg_bwd_slice.argtypes[:-1] * 2 + [ctypes.POINTER(_ci), _vp]
_lib.pn2x_tg_bwd_slice_pair.restype = _ci
PAIR_LAUNCH = _os.environ.get("HOTRACK_STACK_PAIR_LAUNCH", "1") != "0"  # one launch for equal-shaped layers of two sibling stacks
_lib.pn2x_tg_reduce_multi2.argtypes = [_ci, ctypes.POINTER(_vp), ctypes.POINTER(_ci), ctypes.POINTER(_ci), ctypes.POINTER(_vp),
                                       ctypes.POINTER(_vp), ctypes.POINTER(_ci), ctypes.POINTER(_ci), ctypes.POINTER(_vp),
                                       ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp]
_lib.pn2x_tg_reduce_multi2.restype = _ci


def _bwd_slices(c_in: int, c_out: int):
    """Column slices [(offset, width)] of layer i the one-kernel backward runs as (one slice where (c_in, c_out) is instantiated,
    2 - 4 slices of 192 / 128 columns for wider layers), or None."""
    if _lib.pn2x_tg_bwd_supported(c_in, c_out):
        return [(0, c_out)]
    for w in (192, 128):
        if c_out % w == 0 and 2 <= c_out // w <= 4 and _lib.pn2x_tg_bwd_supported(c_in, w):
            return [(j * w, w) for j in range(c_out // w)]
    return None


_lib.pn2x_bn_bwd_reduce_routed.argtypes = [_cl, _ci, _ci, _vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _vp, _vp]
_lib.pn2x_bn_bwd_reduce_routed.restype = _ci
_lib.pn2x_tg_bwd.restype = _ci
_lib.pn2x_tg_bwd_set_variant.argtypes = [_ci]
_lib.pn2x_tg_bwd_set_variant.restype = _ci


def set_bwd_kernel_variant(v2: bool) -> None:
    """True (default): the layer backward with W_i register-resident (round 5) where it is instantiated; False: the round-4 kernel
    for every shape.  Process-wide; switch between whole backward passes only (tests, A/B benches)."""
    _native._check(_lib.pn2x_tg_bwd_set_variant(1 if v2 else 0), "tg_bwd_set_variant")
FUSED_BWD = True  # data + weight gradient of a layer in one kernel (train_bwd.hip); False: the two-kernel backward (tests compare)
ROUTE_ON_LOAD = True  # max-pooled top: sums from the arg-max rows, routed on load
_lib.pn2x_bn_bwd_reduce.argtypes = [_cl, _ci, _vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _ci, _vp, _vp]
_lib.pn2x_bn_bwd_reduce.restype = _ci
_lib.pn2x_bn_bwd_reduce_g.argtypes = [_cl, _ci, _vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _ci, _vp, _vp, _ci, _vp]
_lib.pn2x_bn_bwd_reduce_g.restype = _ci
ROUTE_DENSE = True  # max-routed top gradient materialised once by the reduction (where ROUTE_ON_LOAD does not apply)
# The weight gradients are not read before the optimiser: the backward writes only partial tiles and ONE launch at the end of
# the pass (autograd's final callbacks) sums the tiles of every layer of every stack.  Off (HOTRACK_STACK_DEFER_REDUCE=0, or
# `DEFER_REDUCE = False`) where something reads .grad from inside the pass -- DistributedDataParallel's bucket hooks do.
DEFER_REDUCE = (_os.environ.get("HOTRACK_STACK_DEFER_REDUCE", "1") != "0" and hasattr(torch._C, "_current_graph_task_id")
                and hasattr(torch.autograd.Variable._execution_engine, "queue_callback"))  # (engine hooks of this torch build)
_pending = []
_pending_task = [None]  # the autograd pass the pending items belong to (a pass that raised leaves stale ones behind)
_pending_params = set()  # data pointers of the parameters whose gradients of this pass are still unreduced


def _enter_task():
    """Pending state belongs to ONE autograd pass: entering another one drops what a pass that raised left behind.
    Returns True when this call opened the pass (the end-of-pass callback must then be queued)."""
    task = torch._C._current_graph_task_id()
    if _pending_task[0] == task:
        return False
    _pending.clear()
    _pending_params.clear()
    _pending_task[0] = task
    return True


def _defer(item):
    if _enter_task():
        torch.autograd.Variable._execution_engine.queue_callback(_flush_reductions)
    _pending.append(item)


def _flush_reductions():
    items, _pending[:] = list(_pending), []
    _pending_params.clear()
    _pending_task[0] = None
    _reduce_items(items)


def _flush_now():
    """Reduce what is pending without ending the pass (the queued end-of-pass callback then finds nothing)."""
    items, _pending[:] = list(_pending), []
    _pending_params.clear()
    _reduce_items(items)


def _has_hooks(t) -> bool:
    return bool(getattr(t, "_backward_hooks", None)) or bool(getattr(t, "_post_accumulate_grad_hooks", None))


def _may_defer(params) -> bool:
    """Deferring is sound only where autograd ADOPTS the returned tensor as .grad and nothing reads it inside the pass: every
    parameter a leaf without gradient and without hooks (tensor hooks / post-accumulate hooks run inside the pass), and used
    by ONE stack node of this pass.  A parameter met a second time in a pass (weight tying, a module called twice before one
    backward()) would have its two gradients summed by autograd right when this node returns: the pending reductions --
    the first use's among them -- run now, and this node reduces immediately."""
    if not DEFER_REDUCE:
        return False
    if torch._C._current_graph_task_id() != _pending_task[0]:
        _pending.clear()          # (a pass that raised left these behind)
        _pending_params.clear()
        _pending_task[0] = None
    ptrs = [t.data_ptr() for t in params]
    if any(p in _pending_params for p in ptrs) or len(set(ptrs)) != len(ptrs):
        _flush_now()
        return False
    if not all(t.is_leaf and t.grad is None and not _has_hooks(t) for t in params):
        return False
    return True


def _pending_now(item):
    _reduce_items([item])


# ---- gradient homes (network/trainer.py, dp = flat) ------------------------------------------------------------------------------
# A data-parallel trainer exchanges gradients as a few flat buffers.  Instead of packing every .grad into them after the
# backward (and scattering them back for the optimiser: two passes over 16.7 MB per step), it registers, per parameter, the
# slice of the flat buffer its gradient belongs in; the producers below -- the fused stacks' end-of-pass reduction, the
# grouped weight-gradient launch, the first-layer BatchNorm backward -- then write there directly and hand autograd a FRESH
# view of the slice (adopted as .grad without a clone).  Only while the parameter has no gradient yet: an accumulating pass
# gets an ordinary tensor (autograd would add the home to itself).
_grad_home = {}


def set_grad_homes(pairs) -> None:
    """pairs: iterable of (parameter, tensor of the parameter's shape that is to receive its gradient); replaces the registry."""
    _grad_home.clear()
    for p, v in pairs:
        if v.shape != p.shape or v.dtype != p.dtype or v.device != p.device or not v.is_contiguous():
            raise ValueError("train_stack.set_grad_homes: a home must be a contiguous tensor of its parameter's shape / dtype / device")
        _grad_home[p.data_ptr()] = (p, v)


def clear_grad_homes() -> None:
    _grad_home.clear()


def grad_buffer(param, shape):
    """Where this pass writes `param`'s gradient, as a tensor of `shape` (same element count): a fresh view of the registered
    home, or a new tensor."""
    if param is not None and param.numel():
        h = _grad_home.get(param.data_ptr())
        if h is not None and h[0].numel() == param.numel() and param.is_leaf and param.grad is None:
            return h[1].view(shape)
    return torch.empty(shape, dtype=_f32, device=param.device if param is not None else None)


def _fv(t):
    """A fresh tensor object on t's memory: autograd adopts a returned gradient without a clone only when nothing else holds
    the very object (the deferred-reduction records do hold these)."""
    return t.view(t.shape)


def _dpar(gamma, beta, bias, n, dev):
    """[d gamma, d beta, d conv bias] of one BatchNorm layer, each (n,): homes where registered (the conv bias' is all zeros)."""
    out = []
    for t in (gamma, beta, bias):
        if t is not None and t.numel() == n:
            out.append(grad_buffer(t, (n,)))
        else:
            out.append(torch.empty((n,), dtype=_f32, device=dev))
    return out


_lib.pn2x_wgrad_multi_max.restype = _ci
_lib.pn2x_wgrad_multi_scratch_floats.argtypes = [_ci, ctypes.POINTER(_ci), ctypes.POINTER(_ci), ctypes.POINTER(_ci)]
_lib.pn2x_wgrad_multi_scratch_floats.restype = _cl
_lib.pn2x_wgrad_multi.argtypes = [_ci, ctypes.POINTER(_vp), ctypes.POINTER(_ci), ctypes.POINTER(_vp), ctypes.POINTER(_ci), ctypes.POINTER(_ci),
                                  ctypes.POINTER(_ci), ctypes.POINTER(_ci), ctypes.POINTER(_vp), ctypes.POINTER(_ci), _vp, _cl, _vp]
_lib.pn2x_wgrad_multi.restype = _ci


class WgradItem:
    """dw (N x K, row stride lddw, written in place) = g^T (R x N) . x (R x K): the weight gradient of a plain linear layer,
    recorded by hotrack_amd.linear_dw during the autograd pass and computed with all the others by ONE grouped launch at its
    end (csrc/train_wgrad.hip).  Holds g, x and the tensor dw points into until then."""
    __slots__ = ("g", "x", "dw", "dw_ptr", "lddw", "n", "k", "stream")

    def __init__(self, g, x, dw, dw_ptr, lddw, n, k, stream):
        self.g, self.x, self.dw, self.dw_ptr, self.lddw, self.n, self.k, self.stream = g, x, dw, dw_ptr, lddw, n, k, stream


def wgrad_multi(items):
    """Run the recorded weight-gradient products (list of WgradItem) now."""
    if not items:
        return
    st = items[0].stream
    if any(it.stream != st for it in items):
        raise RuntimeError("train_stack: deferred weight gradients recorded on different streams")
    cap = int(_lib.pn2x_wgrad_multi_max())
    dev = items[0].g.device
    with torch.cuda.device(dev):
        for i0 in range(0, len(items), cap):
            chunk = items[i0:i0 + cap]
            n = len(chunk)
            g, x, dw = (_vp * n)(), (_vp * n)(), (_vp * n)()
            ldg, ldx, rows, nn, kk, lddw = (_ci * n)(), (_ci * n)(), (_ci * n)(), (_ci * n)(), (_ci * n)(), (_ci * n)()
            for j, it in enumerate(chunk):
                g[j], x[j], dw[j] = it.g.data_ptr(), it.x.data_ptr(), it.dw_ptr
                ldg[j], ldx[j], rows[j], nn[j], kk[j], lddw[j] = it.g.stride(0), it.x.stride(0), it.g.shape[0], it.n, it.k, it.lddw
            nf = int(_lib.pn2x_wgrad_multi_scratch_floats(n, rows, nn, kk))
            if nf < 0:
                raise RuntimeError("train_stack: pn2x_wgrad_multi_scratch_floats rejected the problem list")
            scratch = torch.empty(max(nf, 4), dtype=_f32, device=dev)
            _native._check(_lib.pn2x_wgrad_multi(n, g, ldg, x, ldx, rows, nn, kk, dw, lddw, scratch.data_ptr(), scratch.numel(), st),
                           "wgrad_multi")


def _reduce_items(items):
    wg = [it for it in items if isinstance(it, WgradItem)]
    if wg:
        wgrad_multi(wg)
        items = [it for it in items if not isinstance(it, WgradItem)]
    if not items:
        return
    n = len(items)
    arr = lambda: (_vp * n)()
    part, dw, sm, dg, db, dbi = arr(), arr(), arr(), arr(), arr(), arr()
    P, numel, ch, sld = (_ci * n)(), (_ci * n)(), (_ci * n)(), (_ci * n)()
    for j, it in enumerate(items):
        partial, np_, dwt, sums, dpar, _st = it[:6]
        part[j], dw[j], sm[j] = partial.data_ptr(), dwt.data_ptr(), sums.data_ptr()
        dg[j], db[j], dbi[j] = dpar[0].data_ptr(), dpar[1].data_ptr(), dpar[2].data_ptr()
        P[j], numel[j], ch[j] = np_, dwt.numel(), dwt.shape[0]
        sld[j] = it[6] if len(it) > 6 else dwt.shape[0]  # channels of the whole layer when dwt is a column slice of it
    st = items[0][5]
    if any(it[5] != st for it in items):
        raise RuntimeError("train_stack: deferred reductions recorded on different streams")
    with torch.cuda.device(items[0][2].device):
        _native._check(_lib.pn2x_tg_reduce_multi2(n, part, P, numel, dw, sm, ch, sld, dg, db, dbi, st), "tg_reduce_multi")


_lib.pn2x_bn_bwd_apply_rel.argtypes = [_cl, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _ci, _vp, _vp, _ci, _vp, _vp, _vp, _vp, _vp, _cl, _vp, _vp]
_lib.pn2x_bn_bwd_apply_rel.restype = _ci
_lib.pn2x_bn_bwd_reduce_routed_pair.argtypes = _lib.pn2x_bn_bwd_reduce_routed.argtypes[:-1] * 2 + [_vp]
_lib.pn2x_bn_bwd_reduce_routed_pair.restype = _ci
_lib.pn2x_bn_relu_max_pair.argtypes = _t._lib.pn2x_bn_relu_max.argtypes[:-1] * 2 + [_vp]
_lib.pn2x_bn_relu_max_pair.restype = _ci
_MAX_LD = _t._lib.pn2x_bn_relu_max.argtypes[:-2] + [_ci, _vp]  # (..., out, ldo, arg): the output rows ldo floats apart
_lib.pn2x_bn_relu_max_ld.argtypes = _MAX_LD + [_vp]
_lib.pn2x_bn_relu_max_ld.restype = _ci
_lib.pn2x_bn_relu_max_pair_ld.argtypes = _MAX_LD * 2 + [_vp]
_lib.pn2x_bn_relu_max_pair_ld.restype = _ci
_lib.pn2x_bn_bwd_apply_rel_pair.argtypes = _lib.pn2x_bn_bwd_apply_rel.argtypes[:-1] * 2 + [_vp]
_lib.pn2x_bn_bwd_apply_rel_pair.restype = _ci
_lib.pn2x_bn_bwd_apply_rel_scratch_floats.argtypes = [_cl, _ci]
_lib.pn2x_bn_bwd_apply_rel_scratch_floats.restype = _cl
_lib.pn2x_bn_bwd_apply.argtypes = [_cl, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _ci, _vp, _vp, _ci, _vp, _vp, _vp, _vp]
_lib.pn2x_bn_bwd_apply.restype = _ci
_f32 = torch.float32
_p = _t._p


def supported(c_in: int, c_out: int) -> bool:
    """Layer widths the fused kernels cover (forward, dgrad and wgrad): c_in and c_out multiples of 32, c_in <= 512."""
    return bool(_lib.pn2x_tg_supported(int(c_in), int(c_out))) and c_in % 32 == 0


class Layer:
    """One layer of a stack: weight (C_out, C_in[, 1[, 1]]) | None for the first layer, BatchNorm module, the conv bias (or
    None).  Pass the convolution's weight PARAMETER itself (not a view of it) to let its gradient sum be deferred."""

    def __init__(self, weight, bn, conv_bias=None):
        self.weight, self.bn, self.conv_bias = weight, bn, conv_bias


# The forward / backward of ONE stack are generators: wherever a launch could be grouped with the same launch of a sibling stack
# (the W-resident forward and the one-kernel layer backward of csrc/train_fwd.hip / train_bwd.hip), they yield the launch request
# instead of issuing it; _drive() advances one or two stacks in lockstep and issues each request alone or, for two stacks of
# the same layer shape, as one pair launch (pn2x_tg_fwd2_pair / pn2x_tg_bwd_slice_pair: the two neighbourhood sizes of a
# keypoint-query module).  Everything else is launched where it stands.
class _Fwd:
    @staticmethod
    def gen(y1, K, ws, metas, aux, tensors, out_dst=None):
        # out_dst: (G, C) view of a wider buffer the max-pooled top writes into (the scales of a module: no concatenation launch)
        # tensors: per layer (weight | placeholder, gamma, beta, conv bias | placeholder); metas: per layer (running_mean, running_var, nbt, eps, momentum)
        L = len(metas)
        R, C1 = y1.shape
        dev = y1.device
        py, ldy = _t._rows2d(y1, "y1")
        st = _native._stream(y1)
        ws_f = [ws.take(_lib.pn2x_bn_sums_doubles(tensors[4 * i + 1].shape[0])) for i in range(L)]
        ws_b = [ws.take(_lib.pn2x_bn_sums_doubles(tensors[4 * i + 1].shape[0])) for i in range(L)]
        ys, saved = [y1], []
        # the producer of y1 may have taken its statistics already (train_ops.sa_layer1(ws=...)): a slice of THIS workspace generation
        pre = aux[0].get("sums", {}).pop(aux[1], None) if aux is not None else None
        if pre is not None and pre.numel() == ws_f[0].numel() and pre.device == ws_f[0].device:
            ws_f[0] = pre
        else:
            pre = None
        with _guarded_by_drive(dev):
            if pre is None:
                _native._check(_lib.pn2x_bn_stats(R, C1, py, ldy, ws_f[0].data_ptr(), st), "bn_stats")
            for i in range(1, L):
                w, gamma_p, beta_p, bias_p = tensors[4 * i], tensors[4 * (i - 1) + 1], tensors[4 * (i - 1) + 2], tensors[4 * (i - 1) + 3]
                bias_p = bias_p if bias_p.numel() else None
                rm, rv, nbt, eps, mom = metas[i - 1]
                w = w.view(w.shape[0], -1)  # a Conv1d / Conv2d 1x1 weight parameter as (C_out, C_in)
                N, Kc = w.shape
                if w.stride(1) != 1 or w.stride(0) % 4 or w.data_ptr() % 16:
                    w = w.contiguous()
                y = torch.empty((R, N), dtype=_f32, device=dev)
                sv = torch.empty((2, Kc), dtype=_f32, device=dev)
                x = ys[-1]
                fargs = (R, Kc, N, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), y.data_ptr(), N,
                         ws_f[i - 1].data_ptr(), gamma_p.data_ptr(), beta_p.data_ptr(), _p(bias_p), float(eps),
                         float(mom), _p(rm), _p(rv), _p(nbt), sv[0].data_ptr(), sv[1].data_ptr(), ws_f[i].data_ptr())
                if FWD2 and _lib.pn2x_tg_fwd2_supported(Kc, N):
                    yield ("fwd2", fargs, st, dev)  # (issued by _drive: alone, or grouped with a sibling stack's)
                else:
                    _native._check(_lib.pn2x_tg_fwd(*fargs, st), "tg_fwd")
                ys.append(y)
                saved.append(sv)
            # top of the stack: the streaming kernels (they also finalise the last layer's statistics)
            gamma, beta, bias = tensors[4 * (L - 1) + 1], tensors[4 * (L - 1) + 2], tensors[4 * (L - 1) + 3]
            bias = bias if bias.numel() else None
            rm, rv, nbt, eps, mom = metas[L - 1]
            yl = ys[-1]
            C = yl.shape[1]
            sv = torch.empty((2, C), dtype=_f32, device=dev)
            arg = None
            if K:
                G = R // K
                if (out_dst is not None and tuple(out_dst.shape) == (G, C) and out_dst.stride(1) == 1 and out_dst.stride(0) % 4 == 0
                        and out_dst.data_ptr() % 16 == 0 and out_dst.dtype == _f32):
                    out = out_dst
                else:
                    out = torch.empty((G, C), dtype=_f32, device=dev)
                arg = torch.empty((G, C), dtype=torch.int32, device=dev)
                yield ("max", (G, K, C, yl.data_ptr(), yl.stride(0), ws_f[L - 1].data_ptr(), gamma.data_ptr(), beta.data_ptr(), _p(bias),
                               float(eps), float(mom), _p(rm), _p(rv), _p(nbt), sv[0].data_ptr(), sv[1].data_ptr(), out.data_ptr(),
                               out.stride(0) if G > 1 else C, arg.data_ptr()), st, dev)  # (issued by _drive: alone, or with the sibling stack's as one pair launch)
            else:
                out = torch.empty((R, C), dtype=_f32, device=dev)
                _native._check(_lib.pn2x_bn_relu_apply(R, C, yl.data_ptr(), yl.stride(0), ws_f[L - 1].data_ptr(), gamma.data_ptr(),
                                                       beta.data_ptr(), _p(bias), float(eps), float(mom), _p(rm), _p(rv), _p(nbt),
                                                       sv[0].data_ptr(), sv[1].data_ptr(), out.data_ptr(), C, 1, st), "bn_relu_apply")
            saved.append(sv)
        info = _Info()
        info.L, info.K, info.aux = L, K, aux
        info.has_bias = [tensors[4 * i + 3].numel() > 0 for i in range(L)]
        info.ws, info.ws_gen, info.ws_b_all = ws, ws.generation, ws_b
        return out, [*ys, *saved, *([arg] if arg is not None else []), *tensors], info


class _Info:
    """What a stack's backward needs besides its saved tensors (the `ctx` attributes of a single-stack Function)."""
    ws_used = False


class _Bwd:
    @staticmethod
    def gen(ctx, t, dout):
        L, K = ctx.L, ctx.K
        ys, saved = t[:L], t[L:2 * L]
        off = 2 * L
        arg = None
        if K:
            arg = t[off]
            off += 1
        tensors = t[off:]
        dev = dout.device
        # a column block of a wider gradient (the concatenated scales of a query module) is read in place where the kernels
        # take a row stride (the routed path); everything else gets a contiguous copy
        strided_ok = (dout.dim() == 2 and dout.stride(1) == 1 and dout.stride(0) % 4 == 0 and dout.data_ptr() % 16 == 0)
        dout_c = dout if dout.is_contiguous() else None
        R = ys[0].shape[0]
        st = _native._stream(dout)
        # backward accumulators: this forward's slices while they are fresh, zeros otherwise (train_ops.Workspace)
        if ctx.ws_gen == ctx.ws.generation and not getattr(ctx, "ws_used", False):
            ctx.ws_used = True
            sums = ctx.ws_b_all
        else:
            sums = [torch.zeros(s.numel(), dtype=torch.float64, device=dev) for s in ctx.ws_b_all]
        grads = [None] * len(tensors)
        # deferring is only sound when autograd ADOPTS the returned tensors (grad is None); an in-place accumulation into an
        # existing .grad would read them before the reduction ran
        # (nor may anything downstream consume them inside the pass: leaves only)
        # (the conv bias shares dpar with gamma / beta: dpar[2] is written by the same deferred launch)
        dparams = [tensors[j] for i in range(1, L) for j in (4 * i, 4 * i + 1, 4 * i + 2)]
        dparams += [tensors[4 * i + 3] for i in range(1, L) if ctx.has_bias[i]]
        defer = _may_defer(dparams)
        if defer:
            _enter_task() and torch.autograd.Variable._execution_engine.queue_callback(_flush_reductions)
            _pending_params.update(t.data_ptr() for t in dparams)
        gam = lambda i: tensors[4 * i + 1]
        bet = lambda i: tensors[4 * i + 2]
        with _guarded_by_drive(dev):
            yl, svl = ys[L - 1], saved[L - 1]
            Cl = yl.shape[1]
            g, gmode = dout, (2 if K else 1)
            g_dense = None
            wl = tensors[4 * (L - 1)]
            routed = bool(K and L > 1 and FUSED_BWD and ROUTE_ON_LOAD and _bwd_slices(wl.shape[1], wl.shape[0]))
            if not (routed and strided_ok) and dout_c is None:
                dout_c = dout.contiguous()
            if dout_c is not None:
                dout = dout_c
            g, gmode = dout, (2 if K else 1)
            if routed:
                # the routed gradient is non-zero in one row per (group, channel): the sums need the arg-max rows only, and the
                # one-kernel layer backward below routes dout on load (no rows x C gradient tensor at all)
                yield ("routed", (R // K, K, Cl, dout.data_ptr(), dout.stride(0), arg.data_ptr(), Cl, yl.data_ptr(), yl.stride(0),
                                  svl[0].data_ptr(), svl[1].data_ptr(), gam(L - 1).data_ptr(), bet(L - 1).data_ptr(),
                                  sums[L - 1].data_ptr()), st, dev)
            elif K and ROUTE_DENSE:
                # the reduction reads dout / arg / y_L anyway: it also writes the routed, masked gradient once (dense), and the two
                # GEMMs of the top layer read that instead of routing through arg-max on every load (their slowest variant)
                g_dense = torch.empty((R, Cl), dtype=_f32, device=dev)
            if not routed:
                _native._check(_lib.pn2x_bn_bwd_reduce_g(R, Cl, dout.data_ptr(), Cl, _p(arg), K if K else 1, yl.data_ptr(), yl.stride(0),
                                                         svl[0].data_ptr(), svl[1].data_ptr(), gam(L - 1).data_ptr(),
                                                         bet(L - 1).data_ptr(), 1, sums[L - 1].data_ptr(), _p(g_dense), Cl, st),
                               "bn_bwd_reduce")
            if g_dense is not None:
                g, gmode = g_dense, 0
            for i in range(L - 1, 0, -1):
                w_shape = tensors[4 * i].shape
                w = tensors[4 * i].view(w_shape[0], -1)
                N, Kc = w.shape
                wc = w if (w.stride(1) == 1 and w.stride(0) % 4 == 0 and w.data_ptr() % 16 == 0) else w.contiguous()
                yi, yp, svi, svp = ys[i], ys[i - 1], saved[i], saved[i - 1]
                dy_args = (gmode, g.data_ptr(), g.stride(0), _p(arg) if gmode == 2 else None, K if gmode == 2 else 1, yi.data_ptr(),
                           yi.stride(0), svi[0].data_ptr(), svi[1].data_ptr(), gam(i).data_ptr(), bet(i).data_ptr(), sums[i].data_ptr())
                dw = grad_buffer(tensors[4 * i], (N, Kc))
                dpar = _dpar(gam(i), bet(i), tensors[4 * i + 3] if ctx.has_bias[i] else None, N, dev)
                gp = torch.empty((R, Kc), dtype=_f32, device=dev)
                slices = _bwd_slices(Kc, N) if (FUSED_BWD and (gmode != 2 or routed)) else None
                if slices:
                    for j, (off, wd) in enumerate(slices):  # (one slice unless the layer is wider than the instantiated kernels)
                        np_ = int(_lib.pn2x_tg_bwd_partials(R, wd, Kc))
                        partial = torch.empty(np_ * wd * Kc, dtype=_f32, device=dev)
                        o4 = 4 * off
                        bargs = (
                            R, wd, Kc, gmode, g.data_ptr() + o4, g.stride(0), (arg.data_ptr() + o4) if gmode == 2 else None,
                            arg.stride(0) if gmode == 2 else 4, K if gmode == 2 else 1, yi.data_ptr() + o4, yi.stride(0), svi[0].data_ptr() + o4, svi[1].data_ptr() + o4,
                            gam(i).data_ptr() + o4, bet(i).data_ptr() + o4, sums[i].data_ptr() + 8 * off, N,
                            wc.data_ptr() + o4 * wc.stride(0), wc.stride(0), yp.data_ptr(), yp.stride(0), svp[0].data_ptr(),
                            svp[1].data_ptr(), gam(i - 1).data_ptr(), bet(i - 1).data_ptr(), gp.data_ptr(), Kc, sums[i - 1].data_ptr(),
                            partial.data_ptr(), partial.numel(), dw.data_ptr() + o4 * Kc, gp.data_ptr() if j else None, Kc,
                            1 if j + 1 < len(slices) else 0)
                        # issued by _drive; the answer is the number of partial tiles actually written (a pair launch gives this
                        # problem a share of the grid, fewer workgroups than it would get alone)
                        np_ = yield ("bwd", bargs, st, dev, np_)
                        item = (partial, np_, dw[off:off + wd], sums[i][off:], [t_[off:off + wd] for t_ in dpar], st, N)
                        if defer:
                            _defer(item)
                        else:
                            _pending_now(item)
                    grads[4 * i], grads[4 * i + 1], grads[4 * i + 2] = dw.view(w_shape), _fv(dpar[0]), _fv(dpar[1])
                    if ctx.has_bias[i]:
                        grads[4 * i + 3] = _fv(dpar[2])
                    g, gmode = gp, 0
                    continue
                pf = int(_lib.pn2x_tg_wgrad_partial_floats(R, N, Kc))
                partial = torch.empty(pf, dtype=_f32, device=dev)
                np_ = _ci(0)
                _native._check(_lib.pn2x_tg_wgrad2(R, N, Kc, *dy_args, yp.data_ptr(), yp.stride(0), svp[0].data_ptr(), svp[1].data_ptr(),
                                                   gam(i - 1).data_ptr(), bet(i - 1).data_ptr(), partial.data_ptr(), pf, dw.data_ptr(),
                                                   dpar[0].data_ptr(), dpar[1].data_ptr(), dpar[2].data_ptr(),
                                                   ctypes.byref(np_) if defer else None, st), "tg_wgrad")
                if defer:
                    _defer((partial, np_.value, dw, sums[i], dpar, st))
                dw = dw.view(w_shape)  # a fresh view: autograd adopts it (no clone); a late reduction lands in the adopted storage
                grads[4 * i], grads[4 * i + 1], grads[4 * i + 2] = dw, _fv(dpar[0]), _fv(dpar[1])
                if ctx.has_bias[i]:
                    grads[4 * i + 3] = _fv(dpar[2])  # zeros: the bias of a convolution in front of a BatchNorm has no gradient
                _native._check(_lib.pn2x_tg_dgrad(R, N, Kc, *dy_args, wc.data_ptr(), wc.stride(0), yp.data_ptr(), yp.stride(0),
                                                  svp[0].data_ptr(), svp[1].data_ptr(), gam(i - 1).data_ptr(), bet(i - 1).data_ptr(),
                                                  gp.data_ptr(), Kc, sums[i - 1].data_ptr(), st), "tg_dgrad")
                g, gmode = gp, 0
            # first layer: materialise dY_1 (its producer -- a row gather or a library GEMM -- takes it from here)
            y1, sv1 = ys[0], saved[0]
            C1 = y1.shape[1]
            dy1 = torch.empty((R, C1), dtype=_f32, device=dev)
            dpar = _dpar(gam(0), bet(0), tensors[3] if ctx.has_bias[0] else None, C1, dev)
            rel = None
            if ctx.aux is not None:  # (aux dict of train_ops.sa_layer1, scale index): also d(W_xyz) = dY_1^T rel from this pass
                adict, ai = ctx.aux
                rel = adict.get("rel", [None] * (ai + 1))[ai]
                if rel is None or not rel.is_contiguous() or rel.numel() != 3 * R or C1 > 256 or gmode != 0:
                    rel = None
            if rel is not None:
                nf = int(_lib.pn2x_bn_bwd_apply_rel_scratch_floats(R, C1))
                scratch = torch.empty(nf, dtype=_f32, device=dev)
                dwx = torch.empty((C1, 3), dtype=_f32, device=dev)
                # (issued by _drive: alone, or with the sibling stack's as one pair launch)
                yield ("apply_rel", (R, C1, g.data_ptr(), g.stride(0), y1.data_ptr(), y1.stride(0), sv1[0].data_ptr(), sv1[1].data_ptr(),
                                     gam(0).data_ptr(), bet(0).data_ptr(), 0, sums[0].data_ptr(), dy1.data_ptr(), C1, dpar[0].data_ptr(),
                                     dpar[1].data_ptr(), dpar[2].data_ptr(), rel.data_ptr(), scratch.data_ptr(), nf, dwx.data_ptr()), st, dev)
                adict["dwx"][ai] = dwx
            else:
                _native._check(_lib.pn2x_bn_bwd_apply(R, C1, g.data_ptr(), g.stride(0), y1.data_ptr(), y1.stride(0), sv1[0].data_ptr(),
                                                      sv1[1].data_ptr(), gam(0).data_ptr(), bet(0).data_ptr(), 0, sums[0].data_ptr(),
                                                      dy1.data_ptr(), C1, dpar[0].data_ptr(), dpar[1].data_ptr(), dpar[2].data_ptr(), st),
                               "bn_bwd_apply")
            grads[1], grads[2] = _fv(dpar[0]), _fv(dpar[1])
            if ctx.has_bias[0]:
                grads[3] = _fv(dpar[2])
        return dy1, grads


def _pairable(a, b):
    """Two launch requests that one pair launch can serve: same kind, same layer shape (and gradient source), same stream."""
    if a[0] != b[0] or a[2] != b[2] or a[3] != b[3] or not PAIR_LAUNCH:
        return False
    if a[0] == "fwd2":
        return a[1][1:3] == b[1][1:3] and bool(_lib.pn2x_tg_fwd2_pair_supported(a[1][1], a[1][2]))
    if a[0] in ("apply_rel", "max", "routed"):
        return True
    return a[1][1:4] == b[1][1:4] and bool(_lib.pn2x_tg_bwd_pair_supported(a[1][2], a[1][1]))


def _issue(req):
    kind, args, st, dev = req[:4]
    with torch.cuda.device(dev):
        if kind == "fwd2":
            _native._check(_lib.pn2x_tg_fwd2(*args, st), "tg_fwd2")
            return None
        if kind == "apply_rel":
            _native._check(_lib.pn2x_bn_bwd_apply_rel(*args, st), "bn_bwd_apply_rel")
            return None
        if kind == "max":
            _native._check(_lib.pn2x_bn_relu_max_ld(*args, st), "bn_relu_max")
            return None
        if kind == "routed":
            _native._check(_lib.pn2x_bn_bwd_reduce_routed(*args, st), "bn_bwd_reduce_routed")
            return None
        _native._check(_lib.pn2x_tg_bwd_slice(*args, st), "tg_bwd")
        return req[4]


def _issue_pair(a, b):
    kind, st, dev = a[0], a[2], a[3]
    with torch.cuda.device(dev):
        if kind == "fwd2":
            _native._check(_lib.pn2x_tg_fwd2_pair(*a[1], b[1][0], *b[1][3:], st), "tg_fwd2_pair")
            return None, None
        if kind == "apply_rel":
            _native._check(_lib.pn2x_bn_bwd_apply_rel_pair(*a[1], *b[1], st), "bn_bwd_apply_rel_pair")
            return None, None
        if kind == "max":
            _native._check(_lib.pn2x_bn_relu_max_pair_ld(*a[1], *b[1], st), "bn_relu_max_pair")
            return None, None
        if kind == "routed":
            _native._check(_lib.pn2x_bn_bwd_reduce_routed_pair(*a[1], *b[1], st), "bn_bwd_reduce_routed_pair")
            return None, None
        nparts = (_ci * 2)()
        _native._check(_lib.pn2x_tg_bwd_slice_pair(*a[1], *b[1], nparts, st), "tg_bwd_pair")
        return int(nparts[0]), int(nparts[1])


class _guarded_by_drive:
    """What the stack generators wrap their launches in: nothing.  A generator that yields from inside `torch.cuda.device(dev)`
    enters and leaves that guard out of LIFO order when two of them run in lockstep (the one that finishes first restores the
    previous device under the other's feet: its remaining direct launches would go to the wrong device when the model is not on
    the current one).  _drive() holds ONE guard around the whole lockstep loop instead; this context only checks that."""

    def __init__(self, dev):
        self.dev = dev

    def __enter__(self):
        if self.dev.type == "cuda" and torch.cuda.current_device() != (self.dev.index if self.dev.index is not None else torch.cuda.current_device()):
            raise RuntimeError("train_stack: a stack generator must be advanced by _drive() (device guard)")
        return self

    def __exit__(self, *exc):
        return False


def _drive(gens, dev):
    """Run the stack generators in lockstep under ONE device guard; returns their return values.  Whatever happens, no
    generator is left suspended (a generator that outlives an exception of its sibling would run its cleanup at GC time)."""
    n = len(gens)
    results, answer, alive = [None] * n, [None] * n, list(range(n))
    try:
        with torch.cuda.device(dev):
            while alive:
                reqs = {}
                for i in list(alive):
                    try:
                        reqs[i] = gens[i].send(answer[i])
                    except StopIteration as stop:
                        results[i] = stop.value
                        alive.remove(i)
                    answer[i] = None
                if len(reqs) == 2 and _pairable(*reqs.values()):
                    (i, a), (j, b) = reqs.items()
                    answer[i], answer[j] = _issue_pair(a, b)
                else:
                    for i, r in reqs.items():
                        answer[i] = _issue(r)
    finally:
        for g in gens:
            g.close()
    return results


class _Stack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y1, K, ws, metas, aux, *tensors):
        (out, saved, info), = _drive([_Fwd.gen(y1, K, ws, metas, aux, tensors)], y1.device)
        ctx.save_for_backward(*saved)
        ctx.info = info
        return out

    @staticmethod
    def backward(ctx, dout):
        (dy1, grads), = _drive([_Bwd.gen(ctx.info, ctx.saved_tensors, dout)], dout.device)
        return (dy1, None, None, None, None, *grads)


class _StackPair(torch.autograd.Function):
    """Two stacks of the same layer widths (different weights, different row counts) whose fused launches are grouped."""

    @staticmethod
    def forward(ctx, y1a, y1b, Ka, Kb, ws, metas_a, metas_b, aux_a, aux_b, n_a, *tensors):
        ra, rb = _drive([_Fwd.gen(y1a, Ka, ws, metas_a, aux_a, tensors[:n_a]), _Fwd.gen(y1b, Kb, ws, metas_b, aux_b, tensors[n_a:])], y1a.device)
        ctx.save_for_backward(*ra[1], *rb[1])
        ctx.split, ctx.infos = len(ra[1]), (ra[2], rb[2])
        return ra[0], rb[0]

    @staticmethod
    def backward(ctx, da, db):
        t = ctx.saved_tensors
        ra, rb = _drive([_Bwd.gen(ctx.infos[0], t[:ctx.split], da), _Bwd.gen(ctx.infos[1], t[ctx.split:], db)], da.device)
        return (ra[0], rb[0], None, None, None, None, None, None, None, None, *ra[1], *rb[1])


class _StackPairCat(torch.autograd.Function):
    """_StackPair whose two max-pooled tops write the two column blocks of ONE (G, Ca + Cb) tensor -- what a multi-scale module
    returns (reference pointnet_utils.py:405-409, :583-590: `torch.cat(new_points_list, dim=1)`) -- instead of being concatenated by a
    copy launch; the backward reads its two column blocks of the incoming gradient in place."""

    @staticmethod
    def forward(ctx, y1a, y1b, Ka, Kb, ws, metas_a, metas_b, aux_a, aux_b, n_a, *tensors):
        ca, cb = tensors[n_a - 3].shape[0], tensors[-3].shape[0]  # (the last layers' BatchNorm weights)
        both = torch.empty((y1a.shape[0] // Ka, ca + cb), dtype=_f32, device=y1a.device)
        ra, rb = _drive([_Fwd.gen(y1a, Ka, ws, metas_a, aux_a, tensors[:n_a], both[:, :ca]),
                         _Fwd.gen(y1b, Kb, ws, metas_b, aux_b, tensors[n_a:], both[:, ca:])], y1a.device)
        if ra[0].data_ptr() != both.data_ptr() or rb[0].data_ptr() != both.data_ptr() + 4 * ca:  # (a top that could not take the view)
            both = torch.cat([ra[0], rb[0]], dim=1)
        ctx.save_for_backward(*ra[1], *rb[1])
        ctx.split, ctx.infos, ctx.ca = len(ra[1]), (ra[2], rb[2]), ca
        return both

    @staticmethod
    def backward(ctx, dboth):
        t = ctx.saved_tensors
        da, db = dboth[:, :ctx.ca], dboth[:, ctx.ca:]
        ra, rb = _drive([_Bwd.gen(ctx.infos[0], t[:ctx.split], da), _Bwd.gen(ctx.infos[1], t[ctx.split:], db)], dboth.device)
        return (ra[0], rb[0], None, None, None, None, None, None, None, None, *ra[1], *rb[1])


def mlp_stack(y1: torch.Tensor, layers, ws, max_over: int = 0, aux=None) -> torch.Tensor:
    """relu(BN_L(... relu(BN_1(y1)) W_2^T ...)), optionally followed by the max over every `max_over` consecutive rows.
    y1 (R, C_1) pre-activations of layer 1 (without the conv bias: it cancels in the normalisation); layers: list of Layer;
    ws: train_ops.Workspace.  Conv biases get no gradient here (identically zero in front of a BatchNorm): the caller returns
    zeros for them where the reference's `grad is None` mask needs a tensor.
    aux = (dict, i): the side channel of train_ops.sa_layer1 whose i-th output y1 is (its relative coordinates are read from
    dict["rel"][i]; the first-layer backward then also leaves d(W_xyz) in dict["dwx"][i])."""
    R = y1.shape[0]
    if max_over and R % max_over:
        raise ValueError("mlp_stack: rows must be a multiple of max_over")
    if len(layers) == 1:  # nothing to fuse: the streaming kernels
        l = layers[0]
        return _t.bn_relu_max(y1, max_over, l.bn, ws, l.conv_bias) if max_over else _t.bn_relu(y1, l.bn, ws, l.conv_bias)
    metas, tensors = [], []
    for i, l in enumerate(layers):
        bn = l.bn
        track = bn.track_running_stats and bn.running_mean is not None
        metas.append((bn.running_mean if track else None, bn.running_var if track else None,
                      bn.num_batches_tracked if track else None, bn.eps, bn.momentum if bn.momentum is not None else 0.1))
        none = y1.new_empty(0)
        tensors += [l.weight if i else none, bn.weight, bn.bias, l.conv_bias if l.conv_bias is not None else none]
    return _Stack.apply(y1, int(max_over), ws, metas, aux, *tensors)


def _stack_inputs(y1, layers):
    metas, tensors = [], []
    for i, l in enumerate(layers):
        bn = l.bn
        track = bn.track_running_stats and bn.running_mean is not None
        metas.append((bn.running_mean if track else None, bn.running_var if track else None,
                      bn.num_batches_tracked if track else None, bn.eps, bn.momentum if bn.momentum is not None else 0.1))
        none = y1.new_empty(0)
        tensors += [l.weight if i else none, bn.weight, bn.bias, l.conv_bias if l.conv_bias is not None else none]
    return metas, tensors


def mlp_stack_pair(y1a, y1b, layers_a, layers_b, ws, max_over_a: int = 0, max_over_b: int = 0, aux_a=None, aux_b=None, cat: bool = False):
    """mlp_stack for TWO stacks at once: (mlp_stack(y1a, layers_a, ...), mlp_stack(y1b, layers_b, ...)), with the fused launches of
    layers of equal shape grouped into one launch each (forward and backward).  For the neighbourhood sizes of a multi-scale
    module: same widths, different weights, different row counts."""
    if len(layers_a) < 2 or len(layers_b) < 2:
        return mlp_stack(y1a, layers_a, ws, max_over_a, aux_a), mlp_stack(y1b, layers_b, ws, max_over_b, aux_b)
    for y, k in ((y1a, max_over_a), (y1b, max_over_b)):
        if k and y.shape[0] % k:
            raise ValueError("mlp_stack_pair: rows must be a multiple of max_over")
    metas_a, tensors_a = _stack_inputs(y1a, layers_a)
    metas_b, tensors_b = _stack_inputs(y1b, layers_b)
    if (cat and max_over_a and max_over_b and y1a.shape[0] // max_over_a == y1b.shape[0] // max_over_b
            and layers_a[-1].bn.weight.shape[0] % 4 == 0 and layers_b[-1].bn.weight.shape[0] % 4 == 0):
        # cat = True: ONE (G, Ca + Cb) tensor = torch.cat((out_a, out_b), dim=1), written in place by the two tops
        return _StackPairCat.apply(y1a, y1b, int(max_over_a), int(max_over_b), ws, metas_a, metas_b, aux_a, aux_b, len(tensors_a), *tensors_a, *tensors_b)
    if cat:
        oa, ob = _StackPair.apply(y1a, y1b, int(max_over_a), int(max_over_b), ws, metas_a, metas_b, aux_a, aux_b, len(tensors_a), *tensors_a, *tensors_b)
        return torch.cat([oa, ob], dim=1)
    return _StackPair.apply(y1a, y1b, int(max_over_a), int(max_over_b), ws, metas_a, metas_b, aux_a, aux_b, len(tensors_a), *tensors_a, *tensors_b)


def stack_supported(c1: int, widths) -> bool:
    """True when every fused layer (C_{i-1} -> C_i, i >= 2) is covered by the kernels."""
    prev = c1
    for c in widths:
        if not supported(prev, c):
            return False
        prev = c
    return c1 % 4 == 0 and c1 <= 1024
