"""Plain linear layers of the training path with the WEIGHT GRADIENT deferred to the end of the autograd pass.

    linear(x, weight, bias=None)                     torch.nn.functional.linear for 2-D x and a weight PARAMETER
                                                     (C_out, C_in[, 1[, 1]]); same values forward
    linear_blocks(blocks, weight)                    the same for an input that is a column concatenation, without the concatenation
    per_point_first_layer(x, groups, D)              the per-point halves x . W_f^T of the first layers of set-abstraction
                                                     modules whose weights are [feature | xyz | centre] column blocks
                                                     (reference pointnet_utils.py:389-403, :566-581), plus the xyz / centre
                                                     blocks as differentiable views

Nothing reads a weight gradient before the optimiser step, and `grad_out^T . x` reduces 672 ... 32768 rows into a small
matrix -- as separate library GEMMs these were 15 latency-bound launches of a training step.  The backward here computes the
input gradient at once (it is on the critical path) and only RECORDS (grad_out, x, where dW goes); ONE grouped launch at the
end of the pass (train_stack's end-of-pass callback, csrc/train_wgrad.hip: pn2x_wgrad_multi) computes every recorded product,
writing straight into the tensor autograd adopted as `.grad` (for a first-layer weight: into its feature column block).

Deferring is only sound where autograd ADOPTS the returned tensor and nothing reads it inside the pass (train_stack._may_defer:
leaf parameter without gradient, without hooks, met once in the pass; off under DistributedDataParallel).  Otherwise the
product is a library GEMM inside the pass, as before.  GPU tensors only (no CPU path).
"""
from __future__ import annotations

import torch

from . import pointnet2_hip as _native
from . import train_stack as _ts

_f32 = torch.float32


def _rows(t: torch.Tensor) -> torch.Tensor:
    """t as a 2-D operand the grouped kernel takes: unit column stride, row stride >= columns."""
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) < t.shape[1] or t.dtype != _f32:
        t = t.contiguous().float()
    return t


def _defer_ok(params) -> bool:
    ok = _ts._may_defer(params)
    if ok:
        if _ts._enter_task():
            torch.autograd.Variable._execution_engine.queue_callback(_ts._flush_reductions)
        _ts._pending_params.update(p.data_ptr() for p in params)
    return ok


def _record(g, x, dw_owner, dw_ptr, lddw, n, k):
    _ts._pending.append(_ts.WgradItem(g, x, dw_owner, dw_ptr, lddw, n, k, _native._stream(g)))


_consts = {}
_retired = []  # constants replaced by larger ones: captured graphs may still read them (never freed)


def _const(key, rows_needed, make):
    """A process-wide constant tensor, grown on demand -- but never created or replaced inside a graph capture (a tensor from
    the capturing graph's pool is only filled by replays, and replacing one drops memory earlier graphs read): None then, and the
    caller takes its non-deferred path."""
    t = _consts.get(key)
    if t is not None and t.shape[0] >= rows_needed:
        return t
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return None
    if t is not None:
        _retired.append(t)
    t = _consts[key] = make()
    return t


def _ones(rows: int, dev):
    """(rows, 1) of ones: `g^T . ones` = the column sums of g (a bias gradient) as one more problem of the grouped launch."""
    t = _const(("ones", dev), rows, lambda: torch.ones((max(rows, 1024), 1), dtype=_f32, device=dev))
    return None if t is None else t[:rows]


def _eye(n: int, dev):
    """(n, n) identity: `eye^T . block` copies a small gradient block into its columns of a wider weight gradient at the end of
    the pass (instead of a concatenation launch inside it)."""
    t = _const(("eye", dev), n, lambda: torch.eye(max(n, 128), dtype=_f32, device=dev))
    return None if t is None else t[:n, :n]


# ---- a tensor used by TWO consumers: its two input gradients summed by the earlier consumer's product, not by an add launch ------
# autograd sums the gradients a twice-used tensor receives with an element-wise launch (four of them per training step: the
# tail's two residual connections, the two backbone levels that feed both the next level and a skip connection).  The later
# consumer (in forward order; its backward runs first) reads the tensor through tap(x, stash): its gradient is parked in the
# stash instead of being returned; the earlier consumer -- a product of this module given the same stash -- forms its input
# gradient as  parked + g W  (addmm into the parked tensor).  Either order of the two backwards is handled: a product that
# finds nothing parked marks the stash done, and the tap then returns its gradient the ordinary way.
FUSE_GRAD_SUMS = True  # (module attribute: tests set it False to compare with autograd's own sums)


class GradStash:
    __slots__ = ("g", "armed", "done")

    def __init__(self):
        self.g, self.armed, self.done = None, False, False


class _Tap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stash):
        ctx.stash = stash
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        st = ctx.stash
        if g is None or not st.armed or st.done or st.g is not None:
            return g, None  # (nobody will pick it up: the ordinary path)
        st.g = g
        return None, None


def tap(x: torch.Tensor, stash) -> torch.Tensor:
    """x for its LATER consumer (see above); identity when stash is None or the fusion is off."""
    if stash is None or not FUSE_GRAD_SUMS or not x.requires_grad:
        return x
    return _Tap.apply(x, stash)


def _arm(stash, x):
    if stash is not None and FUSE_GRAD_SUMS and x.requires_grad:
        stash.armed = True
        return stash
    return None


def _dx_with_stash(stash, g, w):
    """g . w, plus the gradient parked in `stash` (accumulated into the parked tensor itself)."""
    if stash is not None:
        parked, stash.g, stash.done = stash.g, None, True
        if (parked is not None and parked.is_contiguous() and parked.dtype == g.dtype and parked.device == g.device
                and parked.numel() == g.shape[0] * w.shape[1]):
            return parked.view(g.shape[0], w.shape[1]).addmm_(g, w)
        dx = torch.mm(g, w)
        return dx if parked is None else dx.add_(parked.reshape(dx.shape))
    return torch.mm(g, w)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stash=None):
        ctx.stash = _arm(stash, x)
        w2 = weight.view(weight.shape[0], -1)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.bias = bias if (bias is not None and bias.requires_grad) else None  # (identity only: which parameter receives db)
        return torch.mm(x, w2.t()) if bias is None else torch.addmm(bias, x, w2.t())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        w2 = weight.view(weight.shape[0], -1)
        dx = _dx_with_stash(ctx.stash, g, w2) if ctx.needs_input_grad[0] else None
        want_b = ctx.has_bias and ctx.needs_input_grad[2] and ctx.bias is not None
        dw = db = None
        params = ([weight] if ctx.needs_input_grad[1] else []) + ([ctx.bias] if want_b else [])
        if params and g.is_cuda and _defer_ok(params):
            gg = _rows(g)
            if ctx.needs_input_grad[1]:
                dw = _ts.grad_buffer(weight, w2.shape)  # (the parameter's slice of a flat exchange buffer where one is registered)
                _record(gg, _rows(x), dw, dw.data_ptr(), w2.shape[1], w2.shape[0], w2.shape[1])
                dw = dw.view(weight.shape)  # a fresh view: autograd adopts it (no clone); the late product lands in its storage
            if want_b:  # the column sums of g: one more (N x 1) problem of the grouped launch
                ones = _ones(g.shape[0], g.device)
                if ones is None:  # (a first use inside a graph capture: the plain reduction)
                    db = g.sum(0)
                else:
                    db = _ts.grad_buffer(ctx.bias, (w2.shape[0], 1))
                    _record(gg, ones, db, db.data_ptr(), 1, w2.shape[0], 1)
                    db = db.view(-1)
        else:
            if ctx.needs_input_grad[1]:
                dw = torch.mm(g.t(), x).view(weight.shape)
            if want_b:
                db = g.sum(0)
        return dx, dw, db, None


def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor = None, stash=None) -> torch.Tensor:
    """x (R, C_in) . weight^T (+ bias): `weight` the PARAMETER itself, (C_out, C_in) or a 1x1 convolution's (C_out, C_in, 1[, 1]).
    stash: a GradStash shared with tap(x, stash) at x's later consumer (the two input gradients are then summed by this product)."""
    return _Linear.apply(x, weight, bias, stash)


class _LinearBlocks(torch.autograd.Function):
    """y (R, N) = sum_i X_i . W[:, cols_i]^T for an input that is the column concatenation of the blocks X_i -- without the
    concatenation.  reps[i] = 1: X_i (R, k_i); reps[i] = n > 1: X_i (R / n, k_i) whose every row stands for n consecutive rows of
    the input (a per-cloud feature broadcast over the cloud's points: reference pointnet_utils.py:437-438 repeats it), its product
    is then formed once per source row and added to the n rows it covers.  Input gradients are written contiguous (no slice of a
    wider gradient to be copied); weight gradients per column block at the end of the pass."""

    @staticmethod
    def forward(ctx, weight, reps, stashes, *xs):
        ctx.stashes = tuple(_arm(st, x) for st, x in zip(stashes, xs)) if stashes is not None else (None,) * len(xs)
        w2 = weight.view(weight.shape[0], -1)
        y, col, cols = None, 0, []
        order = sorted(range(len(xs)), key=lambda i: reps[i] != 1)  # a full-rows block first: it creates y
        for i in range(len(xs)):
            cols.append(col)
            col += xs[i].shape[1]
        for i in order:
            x, wb = xs[i], w2[:, cols[i]:cols[i] + xs[i].shape[1]]
            if reps[i] == 1:
                y = torch.mm(x, wb.t()) if y is None else y.addmm_(x, wb.t())
            else:
                t = torch.mm(x, wb.t())  # (R / n, N)
                if y is None:
                    y = t.repeat_interleave(reps[i], dim=0)
                else:
                    y.view(x.shape[0], reps[i], -1).add_(t.unsqueeze(1))
        ctx.save_for_backward(weight, *xs)
        ctx.reps, ctx.cols = tuple(reps), tuple(cols)
        return y

    @staticmethod
    def backward(ctx, g):
        weight, *xs = ctx.saved_tensors
        reps, cols = ctx.reps, ctx.cols
        w2 = weight.view(weight.shape[0], -1)
        N = w2.shape[0]
        gs = {}  # g summed over the rows a broadcast row covers, per repeat factor

        def gsum(n):
            if n not in gs:
                gs[n] = g.reshape(g.shape[0] // n, n, N).sum(1)
            return gs[n]
        dxs = []
        for i, x in enumerate(xs):
            if not ctx.needs_input_grad[3 + i]:
                dxs.append(None)
                continue
            wb = w2[:, cols[i]:cols[i] + x.shape[1]]
            dxs.append(_dx_with_stash(ctx.stashes[i], g if reps[i] == 1 else gsum(reps[i]), wb))
        dw = None
        if ctx.needs_input_grad[0]:
            if g.is_cuda and _defer_ok([weight]):
                gg = _rows(g)
                dw = _ts.grad_buffer(weight, w2.shape)
                for i, x in enumerate(xs):
                    gi = gg if reps[i] == 1 else _rows(gsum(reps[i]))
                    _record(gi, _rows(x), dw, dw.data_ptr() + 4 * cols[i], w2.shape[1], N, x.shape[1])
                dw = dw.view(weight.shape)
            else:
                dw = torch.cat([torch.mm((g if reps[i] == 1 else gsum(reps[i])).t(), x) for i, x in enumerate(xs)], dim=1).view(weight.shape)
        return (dw, None, None, *dxs)


def linear_blocks(blocks, weight: torch.Tensor, stashes=None) -> torch.Tensor:
    """torch.nn.functional.linear(cat(blocks, dim=1), weight) without materialising the concatenation.  blocks: list of X (R, k)
    or (X (R / n, k), n) -- a block whose rows are each repeated n times; weight: the PARAMETER (N, sum k[, 1[, 1]])."""
    xs = [b[0] if isinstance(b, tuple) else b for b in blocks]
    reps = tuple(int(b[1]) if isinstance(b, tuple) else 1 for b in blocks)
    return _LinearBlocks.apply(weight, reps, None if stashes is None else tuple(stashes), *xs)


class _PerPoint(torch.autograd.Function):
    """x (R, D); ws = the first-layer weights (C_s, D + 3 [+ Dc][, 1, 1]) of all scales of `len(sizes)` modules (sizes[m] scales
    each).  Returns per module a1f_m = x . cat_s(w_s[:, :D])^T, then per weight its xyz block (C_s, 3), then per weight its centre
    block (C_s, Dc) (only for weights that have one)."""

    @staticmethod
    def forward(ctx, x, D, sizes, share, stash, *ws):
        ctx.stash = _arm(stash, x)
        w2 = [w.view(w.shape[0], -1) for w in ws]
        blocks = [w[:, :D] for w in w2]
        # ONE product over the feature blocks of every scale of every module (rows of one stacked weight); a module's a1f is its
        # column block of the result (row stride = all columns: the consumers take a row stride)
        wf_all = blocks[0] if len(blocks) == 1 else torch.cat(blocks, dim=0)
        out = torch.mm(x, wf_all.t())
        outs, col, i = [], 0, 0
        for n in sizes:
            width = sum(w.shape[0] for w in w2[i:i + n])
            outs.append(out[:, col:col + width] if len(sizes) > 1 else out)
            col += width
            i += n
        if share is not None:  # where the consumers' backward leaves the gradients of these column blocks: one buffer, see backward
            share["cols"], share["rows"], share["buf"] = col, x.shape[0], None
        outs += [w[:, D:D + 3] for w in w2]
        centre = [w.shape[1] > D + 3 for w in w2]
        outs += [w[:, D + 3:] for w, c in zip(w2, centre) if c]
        ctx.save_for_backward(x, wf_all, *ws)
        ctx.D, ctx.sizes, ctx.centre, ctx.share = D, tuple(sizes), centre, share
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        D, sizes, centre = ctx.D, ctx.sizes, ctx.centre
        x, wf_all, *ws = ctx.saved_tensors
        w2 = [w.view(w.shape[0], -1) for w in ws]
        nm, nw = len(sizes), len(ws)
        ga = grads[:nm]
        gx = grads[nm:nm + nw]
        gc_it = iter(grads[nm + nw:])
        gc = [next(gc_it) if c else None for c in centre]
        dev = x.device
        # input gradient.  When every module's gradient arrived as ITS column block of the shared buffer (train_ops._SaLayer1
        # writes there: `share`), it is one product [g_1 | g_2 ...] . W_all; otherwise the sum over modules of g_m . W_f,m (the
        # first product creates it, the others accumulate: no add pass)
        dx = None
        if ctx.needs_input_grad[0]:
            buf = ctx.share.get("buf") if ctx.share is not None else None
            whole, col = buf is not None and all(g is not None for g in ga), 0
            if whole:
                for g in ga:
                    whole = whole and (g.dim() == 2 and g.stride(1) == 1 and g.stride(0) == buf.stride(0)
                                       and g.data_ptr() == buf.data_ptr() + 4 * col and g.shape[0] == buf.shape[0])
                    col += g.shape[1] if g.dim() == 2 else 0
                whole = whole and col == buf.shape[1]
            if whole:
                dx = _dx_with_stash(ctx.stash, buf, wf_all)
            else:
                row = 0
                for m, n_ in enumerate(sizes):
                    width = sum(w.shape[0] for w in w2[sum(sizes[:m]):sum(sizes[:m]) + n_])
                    if ga[m] is not None:
                        wfm = wf_all[row:row + width]
                        if dx is None:
                            dx = _dx_with_stash(ctx.stash, ga[m], wfm)
                        else:
                            dx.addmm_(ga[m], wfm)
                    row += width
                if dx is None and ctx.stash is not None:  # (no module sent a gradient: what is parked goes on as it is)
                    dx, ctx.stash.g, ctx.stash.done = ctx.stash.g, None, True
            if ctx.share is not None:
                ctx.share["buf"] = None
        live = [w for j, w in enumerate(ws) if ctx.needs_input_grad[5 + j]]
        defer = bool(live) and x.is_cuda and _defer_ok(live)
        xx = _rows(x) if defer else x
        dws, i = [], 0
        for m, n in enumerate(sizes):
            g_m = ga[m]
            if g_m is not None and defer:
                g_m = _rows(g_m)
            c0 = 0
            for j in range(i, i + n):
                w = w2[j]
                C = w.shape[0]
                if not ctx.needs_input_grad[5 + j]:
                    dws.append(None)
                    c0 += C
                    continue
                zx = gx[j] if gx[j] is not None else torch.zeros((C, 3), dtype=_f32, device=dev)
                tail = [zx] + ([gc[j] if gc[j] is not None else torch.zeros((C, w.shape[1] - D - 3), dtype=_f32, device=dev)] if centre[j] else [])
                if g_m is None:
                    full = torch.cat([torch.zeros((C, D), dtype=_f32, device=dev)] + tail, dim=1)
                elif defer and _eye(C, dev) is not None:
                    # [feature block | xyz | centre], every block written at the end of the pass by the grouped launch: the feature
                    # block as g^T x, the two small ones as eye^T . block (a copy as one more problem: no concatenation launch here)
                    full = _ts.grad_buffer(ws[j], (C, w.shape[1]))
                    ld = full.shape[1]
                    _record(g_m[:, c0:c0 + C], xx, full, full.data_ptr(), ld, C, D)
                    eye = _eye(C, dev)
                    col = D
                    for blk in tail:
                        blk = _rows(blk)
                        _record(eye, blk, full, full.data_ptr() + 4 * col, ld, C, blk.shape[1])
                        col += blk.shape[1]
                else:  # (not deferred -- or the identity constant would have had to be created inside a graph capture)
                    full = torch.cat([torch.mm(g_m[:, c0:c0 + C].t(), x)] + tail, dim=1)
                dws.append(full.view(ws[j].shape))
                c0 += C
            i += n
        return (dx, None, None, None, None, *dws)


def per_point_first_layer(x: torch.Tensor, groups, D: int, stash=None):
    """groups: per module the list of its scales' first-layer weight PARAMETERS (C_s, D + 3 [+ Dc][, 1, 1]).
    -> (a1f per module (R, sum_s C_s) -- column blocks of ONE product --, [per module [per scale (xyz block, centre block | None)]],
    share).  `share` (a dict, or None for a single module) lets the consumers' backward write the gradients of the a1f blocks into
    one (R, all columns) buffer -- share_grad_block(share, rows, first column, width) -- so that the input gradient here is one
    product instead of one per module."""
    sizes = tuple(len(g) for g in groups)
    ws = [w for g in groups for w in g]
    share = {} if len(sizes) > 1 else None
    outs = _PerPoint.apply(x, int(D), sizes, share, stash, *ws)
    nm, nw = len(sizes), len(ws)
    a1f = list(outs[:nm])
    wx = list(outs[nm:nm + nw])
    rest = iter(outs[nm + nw:])
    wc = [next(rest) if w.view(w.shape[0], -1).shape[1] > D + 3 else None for w in ws]
    blocks, i = [], 0
    for n in sizes:
        blocks.append([(wx[j], wc[j]) for j in range(i, i + n)])
        i += n
    if share is not None:
        col = 0
        share["first_col"] = []
        for a in a1f:
            share["first_col"].append(col)
            col += a.shape[1]
    return a1f, blocks, share


def share_grad_block(share, module: int, device):
    """The (R, width) column block of the shared gradient buffer that belongs to module `module`'s a1f (allocated on first use in
    a backward pass), or None when there is no shared buffer."""
    if share is None or "cols" not in share:
        return None
    if share.get("buf") is None:
        share["buf"] = torch.empty((share["rows"], share["cols"]), dtype=_f32, device=device)
    c0 = share["first_col"][module]
    c1 = share["first_col"][module + 1] if module + 1 < len(share["first_col"]) else share["cols"]
    return share["buf"][:, c0:c1]
