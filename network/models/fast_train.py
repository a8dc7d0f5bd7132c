"""Point-major TRAINING forward of HandTrackNet's PointNet++ part (backbone + the two keypoint query modules) for MI355X.

Same parameters, same mathematics as the module path (`PointNet2Msg_fast`, `PointNetSetAbstractionMsg_GivenCenterPoints`,
`rearrange_module`; reference backbones.py:114-133, pointnet_utils.py:368-409, :424-464, :484-512, :536-590, blocks.py:226-239)
-- train-mode BatchNorm with batch statistics and running-statistics updates included -- but laid out for the GPU:

  * activations are point-major (rows = B*S*K positions, channels contiguous): a 1x1 convolution is ONE library GEMM over all
    rows (no convolution-library call, no NCHW<->NHWC transposes), and BatchNorm + ReLU between two GEMMs is one stats kernel +
    one apply kernel forward, one reduce + one apply kernel backward (hotrack_amd.train_ops);
  * the grouped inputs [feat_j | xyz_j - c_s | centre feat] are never materialised: layer 1 is linear, so its per-point half
    is a GEMM over the N points and the rest is assembled by pn2x_sa_layer1 (gather + relative coordinates + centre term);
    the backward is a row scatter-add;
  * q1 / q2 share one kNN search (the K=16 list is the prefix of the K=64 list).

The module path (pointnet_utils.py in this directory) remains the general fallback; tests compare the two
(tests/test_gpu_train.py) and both against the reference's golden training step.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F


# ONE switch for "weight gradients of plain linear layers deferred to the grouped end-of-pass launch" (hotrack_amd.linear_dw):
# FastTrain, the rearrange modules and FastTail all read it here (three separate reads of the environment let a
# programmatic change switch the path only partly).  HOTRACK_DEFER_WGRAD=0, or fast_train.DEFER_WGRAD = False before a forward.
DEFER_WGRAD = os.environ.get("HOTRACK_DEFER_WGRAD", "1") != "0"


def _w2d(conv):
    return conv.weight.view(conv.weight.shape[0], -1)


class _SplitCols(torch.autograd.Function):
    """Column blocks of a first-layer weight [feature | xyz | centre-feature] as views, with ONE concatenation as backward.
    Plain slicing costs, per block and step, a zero-fill of the full weight, a copy of the block's gradient into it and an
    accumulation (34 launches of a few microseconds each over the five split weights of the network)."""

    @staticmethod
    def forward(ctx, w, *sizes):
        ctx.sizes = sizes
        ctx.meta = (w.shape[0], w.dtype, w.device)
        return tuple(p for p in w.split(list(sizes), dim=1))

    @staticmethod
    def backward(ctx, *grads):
        rows, dtype, dev = ctx.meta
        parts = [g if g is not None else torch.zeros((rows, n), dtype=dtype, device=dev) for g, n in zip(grads, ctx.sizes)]
        if len(parts) == 1:  # one block = the whole weight (sa1: xyz columns only): its gradient as it is, no copy launch
            return (parts[0].reshape(rows, -1),) + (None,)
        return (torch.cat(parts, dim=1),) + (None,) * len(ctx.sizes)


def _split_first_layer(w, D, has_center):
    """w (C1, D + 3 [+ Dc]) -> (feature block | None, xyz block, centre block | None)."""
    sizes = ([D] if D else []) + [3] + ([w.shape[1] - D - 3] if has_center else [])
    parts = list(_SplitCols.apply(w, *sizes))
    wf = parts.pop(0) if D else None
    wx = parts.pop(0)
    wc = parts.pop(0) if has_center else None
    return wf, wx, wc


class _Linear2Shared(torch.autograd.Function):
    """(x W1^T, x W2^T) for two weight matrices applied to the SAME rows (the per-point halves of layer 1 of the two keypoint
    query modules both read the backbone features).  As two independent linears autograd sums their two input gradients with
    a separate pass over (B*N, C) (23 us at 32 x 1024 x 384); here the second product accumulates into the first (addmm)."""

    @staticmethod
    def forward(ctx, x, w1, w2):
        ctx.save_for_backward(x, w1, w2)
        return torch.mm(x, w1.t()), torch.mm(x, w2.t())

    @staticmethod
    def backward(ctx, g1, g2):
        x, w1, w2 = ctx.saved_tensors
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(g1, w1)
            dx.addmm_(g2, w2)
        dw1 = torch.mm(g1.t(), x) if ctx.needs_input_grad[1] else None
        dw2 = torch.mm(g2.t(), x) if ctx.needs_input_grad[2] else None
        return dx, dw1, dw2


class FastTrain:
    def __init__(self, net):
        self.net = net
        self.ws = None
        import os
        self.use_fused_stacks = os.environ.get("HOTRACK_FUSED_STACKS", "1") != "0"  # 0: round-2 path (library GEMMs + streaming BN)
        self.small_stack_rows = 4096  # stacks with at most this many rows run as unfused layers (8192 lost at batch 64: DESIGN.md 5b)

    @property
    def defer_wgrad(self) -> bool:
        return DEFER_WGRAD  # False: every weight gradient a library GEMM inside the pass

    @staticmethod
    def supported(net) -> bool:
        bh = net.bhand
        mods = (bh.sa1, bh.sa2, net.q1, net.q2)
        if bh.in_dim != 0 or not bh.sa3.group_all or bh.sa1.knn or bh.sa2.knn or not (net.q1.knn and net.q2.knn):
            return False
        if len(bh.sa1.conv_blocks) != 1 or len(bh.sa2.conv_blocks) != 1 or list(net.q1.nsample_list) != list(net.q2.nsample_list):
            return False
        widths = [c.weight.shape[0] for m in mods for convs in m.conv_blocks for c in convs]
        widths += [c.weight.shape[0] for m in (bh.sa3, bh.fp3, bh.fp2, bh.fp1) for c in m.mlp_convs] + [bh.conv1.weight.shape[0]]
        return all(w % 4 == 0 and w <= 1024 for w in widths)

    # ------------------------------------------------------------------------------------------------------------------
    def _stack(self, x2d, convs, bns, first_done=False, max_over=0, aux=None):
        """[1x1 conv + train-mode BatchNorm + ReLU]*; max_over = K > 0: the last layer also takes the max over every K
        consecutive rows (its full-size activations are then never written).  Layers 2.. run as fused BatchNorm GEMMs
        (hotrack_amd.train_stack: normalise + ReLU on load, statistics in the epilogue, dY on load in the backward) when
        their widths are covered; otherwise as library GEMM + streaming BatchNorm kernels (hotrack_amd.train_ops)."""
        from hotrack_amd.train_ops import bn_relu, bn_relu_max
        # few rows (the group-all sa3 and the two coarse feature-propagation stacks: 4096 - 8192 rows at 32 clouds): a fused layer
        # is a latency chain of 32 - 64 workgroups per launch, and since the weight gradients of plain linears left the pass
        # (linear_dw) the unfused form has only the small input-gradient GEMM on the critical path
        small = self.defer_wgrad and x2d.shape[0] <= self.small_stack_rows
        if self.use_fused_stacks and len(convs) > 1 and not small:
            from hotrack_amd import train_stack
            widths = [c.weight.shape[0] for c in convs]
            if train_stack.stack_supported(widths[0], widths[1:]):
                y1 = x2d if first_done else self._linear(x2d, convs[0])
                layers = [train_stack.Layer(None, bns[0], convs[0].bias)]
                layers += [train_stack.Layer(c.weight, bn, c.bias) for c, bn in zip(convs[1:], bns[1:])]
                return train_stack.mlp_stack(y1, layers, self.ws, max_over, aux=aux if first_done else None)
        last = len(convs) - 1
        for i, (conv, bn) in enumerate(zip(convs, bns)):
            y = x2d if (first_done and i == 0) else self._linear(x2d, conv)
            x2d = bn_relu_max(y, max_over, bn, self.ws, conv.bias) if (max_over and i == last) else bn_relu(y, bn, self.ws, conv.bias)
        return x2d

    def _linear(self, x2d, conv, bias=None):
        """x2d . W^T (+ bias) for a Linear / 1x1 convolution module's weight; the weight gradient is computed by the grouped
        end-of-pass launch (hotrack_amd.linear_dw) unless HOTRACK_DEFER_WGRAD=0."""
        if self.defer_wgrad:
            from hotrack_amd.linear_dw import linear
            return linear(x2d, conv.weight, bias)
        return F.linear(x2d, _w2d(conv), bias)

    def _per_point(self, feat2d, mods, D, stash=None):
        """Per module (first-layer blocks [(None, xyz block, centre block | None) per scale], feat2d . W_f^T): the per-point halves of
        the first layers of `mods`, which all read the same rows (one Function: one input gradient, deferred weight gradients)."""
        from hotrack_amd.linear_dw import per_point_first_layer
        groups = [[convs[0].weight for convs in m.conv_blocks] for m in mods]
        a1f, blocks, share = per_point_first_layer(feat2d, groups, D, stash=stash)
        # (share, m): where module m's backward leaves the gradient of its a1f block -- one buffer for all modules, so that the
        # input gradient of the per-point product is one GEMM (hotrack_amd.linear_dw._PerPoint)
        return [([(None, wx, wc) for wx, wc in b], a, (share, m)) for m, (b, a) in enumerate(zip(blocks, a1f))]

    @staticmethod
    def _first_layer_blocks(mod, D, has_center):
        """Per scale (feature block | None, xyz block, centre block | None) of the first-layer weights, and the feature blocks of
        all scales stacked (the weight of the per-point GEMM)."""
        w1 = [_split_first_layer(_w2d(convs[0]), D, has_center) for convs in mod.conv_blocks]
        wf = None
        if D:
            wf = w1[0][0] if len(w1) == 1 else torch.cat([w[0] for w in w1], dim=0)
        return w1, wf

    def _sa_scales(self, mod, xyz, cxyz, feat2d, idxs, center2d=None, pre=None, invs=None, feat_stash=None):
        """All scales of one SA module.  xyz (B,N,3), cxyz (B,S,3), feat2d (B*N, D)|None, center2d (B*S, D2)|None ->
        (B, S, sum C3) point-major.  pre = (w1, a1f2d): the first-layer blocks and the per-point product feat2d wf^T computed by
        the caller (_Linear2Shared)."""
        from hotrack_amd.train_ops import sa_layer1
        B, N, _ = xyz.shape
        S = cxyz.shape[1]
        D = 0 if feat2d is None else feat2d.shape[1]
        a1f = cadd = None
        a1f_share = None
        if pre is not None:
            w1, a1f2d = pre[:2]
            a1f_share = pre[2] if len(pre) > 2 else None
            a1f = a1f2d.view(B, N, -1)
        elif D and self.defer_wgrad:
            (w1, a1f2d, _), = self._per_point(feat2d, [mod], D, stash=feat_stash)
            a1f = a1f2d.view(B, N, -1)
        else:
            w1, wf = self._first_layer_blocks(mod, D, center2d is not None)
            if D:
                a1f = F.linear(feat2d, wf).view(B, N, -1)
        if center2d is not None:
            wc = w1[0][2] if len(w1) == 1 else torch.cat([w[2] for w in w1], dim=0)
            cadd = F.linear(center2d, wc).view(B, S, -1)
        aux = {} if self.use_fused_stacks else None  # relative coordinates -> the stacks, d(W_xyz) <- the stacks (train_ops.sa_layer1)
        if aux is not None and a1f_share is not None and a1f_share[0] is not None:
            aux["a1f_share"] = a1f_share
        y1s = sa_layer1(a1f, cadd, xyz, cxyz, idxs, [w[1] for w in w1], invs=invs, aux=aux,
                        ws=self.ws)
        outs = []
        pair = self._pair_stacks(mod, y1s, idxs, aux) if len(y1s) == 2 else None
        if pair is not None:  # both neighbourhood sizes layer by layer, equal-shaped fused launches grouped (train_stack.mlp_stack_pair);
            return pair.view(B, S, -1)  # their tops wrote the two halves of the module's output (no concatenation launch)
        else:
            for i, y1 in enumerate(y1s):
                K = idxs[i].shape[2]
                h = self._stack(y1.view(B * S * K, -1), mod.conv_blocks[i], mod.bn_blocks[i], first_done=True, max_over=K,
                                aux=None if aux is None else (aux, i))
                outs.append(h.view(B, S, -1))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=2)

    def _pair_stacks(self, mod, y1s, idxs, aux):
        """The two scales of a module as ONE pair of fused stacks when their layer widths agree and the kernels cover them."""
        if not self.use_fused_stacks:
            return None
        from hotrack_amd import train_stack
        widths = [[c.weight.shape[0] for c in convs] for convs in mod.conv_blocks]
        if widths[0] != widths[1] or len(widths[0]) < 2 or not train_stack.stack_supported(widths[0][0], widths[0][1:]):
            return None
        stacks = []
        for i, y1 in enumerate(y1s):
            convs, bns = mod.conv_blocks[i], mod.bn_blocks[i]
            layers = [train_stack.Layer(None, bns[0], convs[0].bias)]
            layers += [train_stack.Layer(c.weight, bn, c.bias) for c, bn in zip(convs[1:], bns[1:])]
            stacks.append((y1.view(-1, y1.shape[-1]), layers, idxs[i].shape[2]))
        (ya, la, ka), (yb, lb, kb) = stacks
        return train_stack.mlp_stack_pair(ya, yb, la, lb, self.ws, ka, kb, aux_a=None if aux is None else (aux, 0),
                                          aux_b=None if aux is None else (aux, 1), cat=True)

    def _fp(self, mod, xyz1, xyz2, points1, points2, extra=None, nn3=None):
        """xyz1 (B,N,3), xyz2 (B,S,3), points1 (B,N,D1)|None, points2 (B,S,D2) -> (B*N, D') rows.
        extra = (conv, bn): a further Conv1d + BatchNorm + ReLU appended to the module's stack (backbone conv1 / bn1).
        nn3 = (weights, indices, inverted lists | None) of the three-NN search when the geometry stage has run it."""
        from hotrack_amd import ext
        from hotrack_amd.train_ops import interpolate_rows
        B, N, _ = xyz1.shape
        if xyz2.shape[1] == 1:
            interp = points2.expand(B, N, points2.shape[2])
        else:
            w, i3, inv = nn3 if nn3 is not None else (*ext.three_nn_weights(xyz1, xyz2), None)
            interp = interpolate_rows(points2, i3, w, inv=inv)
        convs, bns = list(mod.mlp_convs), list(mod.mlp_bns)
        if extra is not None:
            convs.append(extra[0])
            bns.append(extra[1])
        if self.defer_wgrad and self.use_fused_stacks:
            # layer 1 over [skip | interpolated] as the sum of its column blocks' products: no concatenation forward, no copy of a
            # gradient slice backward; a broadcast per-cloud feature (fp3: the global feature) is multiplied once per cloud
            from hotrack_amd.linear_dw import linear_blocks
            blocks = [] if points1 is None else [points1.reshape(B * N, -1)]
            blocks.append((points2.reshape(B, -1), N) if xyz2.shape[1] == 1 else interp.reshape(B * N, -1))
            return self._stack(linear_blocks(blocks, convs[0].weight), convs, bns, first_done=True)
        x = interp if points1 is None else torch.cat([points1, interp], dim=2)
        return self._stack(x.reshape(B * N, -1), convs, bns)

    # ------------------------------------------------------------------------------------------------------------------
    def geometry(self, xyz: torch.Tensor, kp: torch.Tensor, with_inverse: bool = True) -> dict:
        """Everything of a training step that depends on the COORDINATES only (no parameter, no gradient): the two sampling levels
        (furthest point sampling + ball query: reference pointnet_utils.py:368-388), the three-NN searches and weights of the
        feature propagation (:440-449), the keypoints' kNN lists (:551-556) and, for the backward row scatters, the inverted
        neighbour lists.  xyz (B,N,3) hand-frame cloud, kp (B,J,3) hand-frame keypoints, point-major.  A training loop runs this
        stage for batch t+1 on its own stream while batch t's dense work runs (network/trainer.py: update(next_data=...))."""
        from hotrack_amd import ext
        from hotrack_amd import pointnet2_utils as ops
        from hotrack_amd.train_ops import INVERSE_MAX_ROWS, inverse_index
        net, bh = self.net, self.net.bhand
        B, N, _ = xyz.shape
        S1, S2 = bh.sa1.npoint, bh.sa2.npoint
        inv = (lambda i, n: inverse_index(i.view(B, -1), n)) if (with_inverse and N <= INVERSE_MAX_ROWS) else (lambda i, n: None)
        g = {}
        g["l1_xyz"] = ext.gather_rows(xyz, ops.furthest_point_sample(xyz, S1))
        g["idx1"] = ops.ball_query(bh.sa1.radius_list[0], bh.sa1.nsample_list[0], xyz, g["l1_xyz"])
        g["l2_xyz"] = ext.gather_rows(g["l1_xyz"], ops.furthest_point_sample(g["l1_xyz"], S2))
        g["idx2"] = ops.ball_query(bh.sa2.radius_list[0], bh.sa2.nsample_list[0], g["l1_xyz"], g["l2_xyz"])
        g["inv2"] = inv(g["idx2"], S1)
        w, i3 = ext.three_nn_weights(g["l1_xyz"], g["l2_xyz"])      # fp2: level-1 points from level 2
        g["fp2"] = (w, i3, inv(i3, S2))
        w, i3 = ext.three_nn_weights(xyz, g["l1_xyz"])              # fp1: all points from level 1
        g["fp1"] = (w, i3, inv(i3, S1))
        Ks = list(net.q1.nsample_list)
        kmax = max(Ks)
        if len(set(Ks)) == 2 and len(Ks) == 2:  # one search: the smaller list is the prefix of the larger
            gi, gi_small = ext.knn_indices(kmax, kp, xyz, k2=min(Ks))
            idxs = [gi if K == kmax else gi_small for K in Ks]
        else:
            idxs = [ops.knn(K, kp, xyz)[1] for K in Ks]
        g["knn"] = idxs
        g["knn_inv"] = [inv(i, N) for i in idxs]
        if g["knn_inv"][0] is None:
            g["knn_inv"] = None
        return g

    def forward(self, xyz2_cm: torch.Tensor, xyz1_cm: torch.Tensor, geo: dict = None):
        """xyz2_cm (B,3,N) hand-frame cloud, xyz1_cm (B,3,J) hand-frame keypoints (no gradient flows into coordinates:
        they derive from the inputs only) -> f14 (B,C,J) channel-major as `r2` returns it, src2 (B,N,C) point-major.
        geo: the result of geometry() for these coordinates when the caller has it already (computed here otherwise)."""
        from hotrack_amd.train_ops import Workspace
        net, bh = self.net, self.net.bhand
        dev = xyz2_cm.device
        if self.ws is None or self.ws.buf.device != dev:
            self.ws = Workspace(dev)
        self.ws.reset()  # one fill launch: the fp64 accumulators of every BatchNorm reduction of this step, both directions
        if geo is not None and geo.get("frame") is not None:  # the geometry stage's own point-major buffers (no copies here)
            xyz, kp = geo["frame"][2], geo["frame"][3]
        else:
            xyz = xyz2_cm.detach().transpose(1, 2).contiguous()   # (B,N,3)
            kp = xyz1_cm.detach().transpose(1, 2).contiguous()    # (B,J,3)
        B, N, _ = xyz.shape
        J = kp.shape[1]
        if geo is None:
            geo = self.geometry(xyz, kp, with_inverse=torch.is_grad_enabled())

        # ---- backbone: sa1, sa2, sa3 (group-all), fp3, fp2, fp1, conv1 -------------------------------------------------
        S1, S2 = bh.sa1.npoint, bh.sa2.npoint
        l1_xyz, l2_xyz = geo["l1_xyz"], geo["l2_xyz"]
        from hotrack_amd.linear_dw import GradStash, tap
        # l1_feat / l2_feat each feed the next level AND a skip connection: the later consumer reads them through tap(), the
        # earlier one's product sums both input gradients (linear_dw: no element-wise add launch in the backward)
        st1, st2 = (GradStash(), GradStash()) if (self.defer_wgrad and self.use_fused_stacks) else (None, None)
        l1_feat = self._sa_scales(bh.sa1, xyz, l1_xyz, None, [geo["idx1"]])                            # (B,S1,64)
        l2_feat = self._sa_scales(bh.sa2, l1_xyz, l2_xyz, l1_feat.reshape(B * S1, -1), [geo["idx2"]],
                                  invs=None if geo["inv2"] is None else [geo["inv2"]], feat_stash=st1)  # (B,S2,128)
        # group-all: [xyz | feat], centre = origin (not subtracted)
        if self.defer_wgrad and self.use_fused_stacks:
            from hotrack_amd.linear_dw import linear_blocks
            y1 = linear_blocks([l2_xyz.reshape(B * S2, 3), l2_feat.reshape(B * S2, -1)], bh.sa3.mlp_convs[0].weight, stashes=[None, st2])
            l3 = self._stack(y1, bh.sa3.mlp_convs, bh.sa3.mlp_bns, first_done=True, max_over=S2).view(B, 1, -1)  # (B,1,512)
        else:
            x = torch.cat([l2_xyz, l2_feat], dim=2).view(B * S2, -1)
            l3 = self._stack(x, bh.sa3.mlp_convs, bh.sa3.mlp_bns, max_over=S2).view(B, 1, -1)
        l2_out = self._fp(bh.fp3, l2_xyz, l2_xyz[:, :1], tap(l2_feat, st2), l3).view(B, S2, -1)
        l1_out = self._fp(bh.fp2, l1_xyz, l2_xyz, tap(l1_feat, st1), l2_out, nn3=geo["fp2"]).view(B, S1, -1)
        # fp1 (skip = xyz) and the backbone's conv1 / bn1 as one stack [131 -> 128 -> 128 -> C]
        src2 = self._fp(bh.fp1, xyz, l1_xyz, xyz, l1_out, extra=(bh.conv1, bh.bn1), nn3=geo["fp1"])    # (B*N, C)
        src2 = net.cut_after_backbone(src2)  # (a fresh leaf when the trainer runs the backward in two segments: hand_network.py)
        C = src2.shape[1]

        # ---- q1 -> r1 -> q2 -> r2 around the J keypoints; one kNN search for both neighbourhood sizes ------------------
        idxs, invs = geo["knn"], geo["knn_inv"]
        # the per-point halves of both modules' first layers read src2: one Function, one input gradient (_Linear2Shared)
        if self.defer_wgrad:
            (w1_q1, a1f_q1, sh_q1), (w1_q2, a1f_q2, sh_q2) = self._per_point(src2, [net.q1, net.q2], C)
        else:
            sh_q1 = sh_q2 = None
            w1_q1, wf_q1 = self._first_layer_blocks(net.q1, C, False)
            w1_q2, wf_q2 = self._first_layer_blocks(net.q2, C, True)
            a1f_q1, a1f_q2 = _Linear2Shared.apply(src2, wf_q1, wf_q2)
        # both modules gather through the same neighbour lists: inverted once (geometry) for the two backward scatters
        f11 = self._sa_scales(net.q1, xyz, kp, src2, idxs, pre=(w1_q1, a1f_q1, sh_q1), invs=invs)             # (B,J,C)
        f12 = self._rearrange(net.r1, f11)                                                                 # (B*J, C)
        f13 = self._sa_scales(net.q2, xyz, kp, src2, idxs, center2d=f12, pre=(w1_q2, a1f_q2, sh_q2), invs=invs)
        f14 = self._rearrange(net.r2, f13).view(B, J, C)
        self.last_token_rows = f14.view(B * J, C)  # token-major rows for FastTail (the transposed view below is what `r2` returns)
        return f14.transpose(1, 2), src2.view(B, N, C)

    @staticmethod
    def _rearrange(mod, tok):
        """rearrange_module on token-major features tok (B,J,C) -> (B*J, C) (blocks.py: fast formula)."""
        from hotrack_amd.train_ops import INVERSE_MAX_ROWS, gather_rows, inverse_index
        B, J, C = tok.shape
        cache = getattr(mod, "_perm_rows", None)  # (B, J*re) token indices + their inverted lists: static, built once per batch size
        if cache is None or cache[0].shape[0] != B or cache[0].device != tok.device:
            idx = mod._perm.t().reshape(1, -1).to(device=tok.device, dtype=torch.int32).expand(B, -1).contiguous()
            cache = (idx, inverse_index(idx, J) if J <= INVERSE_MAX_ROWS else None)
            mod._perm_rows = cache
        if cache[1] is not None:
            g = gather_rows(tok, cache[0], cache[1])  # (B, J*re, C); backward: one segment-sum launch
        else:
            g = tok.index_select(1, cache[0][0].long())
        if DEFER_WGRAD:
            from hotrack_amd.linear_dw import linear
            return linear(g.view(B * J, mod.re * C), mod.linear.weight, mod.linear.bias)
        return F.linear(g.view(B * J, mod.re * C), mod.linear.weight.squeeze(-1), mod.linear.bias)


class FastTail:
    """The 21-token tail in training mode on the GPU (TransT.s11 -> c11 -> c3 with attn=False, final_mlp, residual on the initial
    keypoints, de-canonicalisation; reference transformer.py:65-67,72-82, hand_network.py:139-147): token-major rows, library
    GEMMs without bias, and every element-wise run between two GEMMs as one launch per direction (hotrack_amd.tail_train:
    LayerNorm pairs, bias + ReLU + dropout, dropout + residual + LayerNorm(s)).  Same parameters, same mathematics; only the
    dropout random stream differs from torch's (statistically equivalent; identical when p = 0)."""

    def __init__(self, net):
        self.net = net
        self.seed = None   # device counter, created at the first forward (see _seed_tensor)
        self._restore = None

    # The dropout masks are a hash of (counter, site, element); the counter lives on the device (capture-safe) and advances by
    # one per forward.  It STARTS from torch's seed mixed with the data-parallel rank -- torch.manual_seed() selects the stream,
    # ranks draw different masks (like the CUDA generator the reference's torch dropout uses) -- and is part of a checkpoint
    # (Trainer.save / resume via dropout_state()), so a resumed run continues the stream instead of replaying it.
    def _seed_tensor(self, dev):
        if self.seed is None or self.seed.device != dev:
            if self._restore is not None:
                start, self._restore = int(self._restore), None
            else:
                import torch.distributed as dist
                rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
                start = ((int(torch.initial_seed()) * 0x9E3779B97F4A7C15) ^ ((rank + 1) * 0xD1B54A32D192ED03)) & ((1 << 62) - 1)
            self._start = start
            self.seed = torch.full((1,), start, dtype=torch.int64, device=dev)
        return self.seed

    def rewind_dropout_counter(self, snapshot=None):
        """Put the device counter back (in place: captured graphs hold its address): to `snapshot` (a clone taken earlier), or to
        the value it was created with when no snapshot exists (the counter was created after the caller's snapshot point).  The
        trainer's capture warm-up runs forwards whose effects must not be seen by the step that follows."""
        if self.seed is None:
            return
        if snapshot is not None:
            self.seed.copy_(snapshot)
        else:
            self.seed.fill_(int(self._start))

    def dropout_state(self) -> int:
        """The counter's current value (a host read: checkpoint time only); None before the first forward."""
        if self.seed is None:
            return self._restore
        return int(self.seed.item())

    def set_dropout_state(self, value):
        if value is None:
            return
        if self.seed is None:
            self._restore = int(value)
        else:
            self.seed.fill_(int(value))

    @staticmethod
    def supported(net) -> bool:
        t, c3 = net.transt, net.c3
        mods = (t.s11, t.c11, c3)
        if any(m.concat for m in mods) or not t.s11.no_linear or t.c11.no_linear or c3.no_linear:
            return False
        C = t.c11.norm1.normalized_shape[0]
        H = t.c11.linear1.out_features
        ok_act = all(m.activation is F.relu for m in (t.c11, c3))
        return ok_act and C % 4 == 0 and C <= 512 and H % 4 == 0 and net.final_mlp[0].weight.shape[0] % 4 == 0

    def forward(self, rows, xyz1, canon_pose):
        """rows (B*J, C) token-major output of r2; xyz1 (B,3,J) hand-frame keypoints -> (pred_kp_handframe (B,3,J), pred_kp (B,J,3))."""
        from hotrack_amd import tail_train as T
        from .hand_utils import decanonicalize
        net = self.net
        s11, c11, c3 = net.transt.s11, net.transt.c11, net.c3
        dev = rows.device
        B, _, J = xyz1.shape
        C, H, Hf = rows.shape[1], c11.linear1.out_features, net.final_mlp[0].weight.shape[0]
        self._seed_tensor(dev)
        seed_used = torch.empty(1, dtype=torch.int64, device=dev)
        grads = T.TailGrads(dev, 14 * C + 2 * H + Hf + 3 * Hf + 3)
        pd = lambda m: float(m.p) if m.training else 0.0
        if DEFER_WGRAD:
            from hotrack_amd.linear_dw import GradStash, linear as lin, tap
            sa, sb = GradStash(), GradStash()  # h / h2 feed their block's first product AND its residual: one input gradient each
        else:
            lin = lambda x, w, stash=None: F.linear(x, w.view(w.shape[0], -1))
            tap = lambda x, stash: x
            sa = sb = None
        h = T.ln(rows, s11.norm1, c11.norm1, grads, seed_dev=self.seed, seed_out=seed_used)
        d = T.relu_dropout(lin(h, c11.linear1.weight, stash=sa), c11.linear1.bias, pd(c11.dropout2), 1, seed_used, grads)
        h2 = T.ln(tap(h, sa), c11.norm2, c3.norm1, grads, y=lin(d, c11.linear2.weight), bias=c11.linear2.bias, p=pd(c11.dropout3), site=2, seed_in=seed_used)
        d2 = T.relu_dropout(lin(h2, c3.linear1.weight, stash=sb), c3.linear1.bias, pd(c3.dropout2), 3, seed_used, grads)
        h3 = T.ln(tap(h2, sb), c3.norm2, None, grads, y=lin(d2, c3.linear2.weight), bias=c3.linear2.bias, p=pd(c3.dropout3), site=4, seed_in=seed_used)
        hf = T.relu_dropout(lin(h3, net.final_mlp[0].weight), net.final_mlp[0].bias, 0.0, 0, None, grads)
        # last Conv1d + residual on the initial keypoints + de-canonicalisation: one launch per direction
        return T.pose_head(hf, net.final_mlp[2].weight.squeeze(-1), net.final_mlp[2].bias, xyz1, canon_pose["rotation"],
                           canon_pose["translation"], canon_pose["scale"], grads)
