"""Point-major eval forward of HandTrackNet for MI355X (inference fast path).

Same network, same parameters, same arithmetic up to fp32 re-association as
`HandTrackNet.forward` (reference hand_network.py:78-157) -- but laid out for the GPU:

  * activations are POINT-MAJOR (B, N, C): every 1x1 convolution becomes ONE 2-D GEMM over all
    B*N points with bias+ReLU in the GEMM epilogue (BatchNorm folded), instead of B small
    batched GEMMs + separate bias / BN / ReLU passes on channel-major tensors;
  * each set-abstraction scale is one fused kernel (pn2x_sa_mlp_max): the per-point half of
    its first layer is a dense GEMM over the N points (for q1/q2 all four scale GEMMs are ONE
    (B*N x 384) x (384 x 512) GEMM), the gather / relative-xyz / centre terms, layers 2-3 (fp32
    MFMA) and the max over K happen in-kernel and write straight into the consumer's input
    buffer (no torch.cat);
  * feature propagation = three_nn with the interpolation weights computed in its epilogue +
    a point-major interpolate that also writes into the consumer's buffer;
  * the discarded attention of the "TransT" blocks is not computed (see transformer.py).

It is selected by HandTrackNet.forward when the fused backend is enabled, the module is in
eval mode, grad is disabled and the input is on the GPU.  Parity: tests/test_gpu_fused.py
(vs the unfused operator path and vs the golden vectors captured from the imported reference).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F



def _lin_relu(x2d: torch.Tensor, W: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """relu(x W^T + b): small problems (the B = 1 / B = 8 tracking loop) through pn2x_linear_small, one workgroup per 32 x 32
    output block; otherwise one library GEMM with a bias+ReLU epilogue (hipBLASLt).  hotrack_amd.ext.linear decides."""
    from hotrack_amd import ext
    return ext.linear(x2d, W, b, relu=True)


def _lin(x2d: torch.Tensor, W: torch.Tensor, b: torch.Tensor = None) -> torch.Tensor:
    """x W^T (+ b), same dispatch."""
    from hotrack_amd import ext
    return ext.linear(x2d, W, b)


class FastEval:
    MAX_POINTS = 16384  # the register-resident FPS with the prefix shortcut covers clouds up to 1024 x 16 points

    def __init__(self, net):
        self.net = net
        self._shapes_seen = set()  # (B, N) of the forwards so far (gemm_tuning.scope(tune=...))
        self._key = None
        self.P = None
        self._consts = {}
        self._idents = {}
        self.two_level_fps = True  # level-2 sampling via the prefix property (ext.fps_two_level)

    def _palm_idx(self, device):
        key = ("palm", str(device))
        if key not in self._consts:
            self._consts[key] = torch.tensor([0, 1, 5, 9, 13, 17], dtype=torch.int32, device=device)  # hand_utils.handkp2palmkp
        return self._consts[key]

    def _scale(self, device):
        key = ("scale", str(device))
        if key not in self._consts:
            self._consts[key] = 0.2 * torch.ones(1, dtype=torch.float32, device=device)
        return self._consts[key]

    # ------------------------------------------------------------------------------------
    def _versions(self):
        """Cache key of the folded weights: in-place edits bump `_version`; replacing a Parameter object, `.to(device)` /
        `.float()` (which assign `param.data`) change `data_ptr()` / the device instead."""
        return tuple((t._version, t.data_ptr(), t.device) for t in list(self.net.parameters()) + list(self.net.buffers()))

    @staticmethod
    def supported(net) -> bool:
        """True when this path covers the network's configuration: single-scale sa1/sa2, two-scale kNN q1/q2, three-layer
        grouped MLPs whose (K, widths) the fused set-abstraction kernel implements, 'kp' hand frame.  Anything else (e.g.
        --network/backbone_out_dim overrides, multi-scale pointnet YAMLs) runs the module path, which falls back to the
        unfused HIP operators per scale."""
        from hotrack_amd import ext
        bh = net.bhand
        if net.handframe != "kp" or not net.elide_dead_attention:
            return False

        def scales_ok(mod, n_scales):
            if len(mod.conv_blocks) != n_scales or len(mod.nsample_list) != n_scales:
                return False
            for convs, K in zip(mod.conv_blocks, mod.nsample_list):
                if len(convs) != 3 or not ext.sa_mlp_max_supported(int(K), *(int(c.weight.shape[0]) for c in convs)):
                    return False
            return True

        if not (scales_ok(bh.sa1, 1) and scales_ok(bh.sa2, 1) and scales_ok(net.q1, 2) and scales_ok(net.q2, 2)):
            return False
        if getattr(bh.sa1, "knn", False) or getattr(bh.sa2, "knn", False) or not (net.q1.knn and net.q2.knn):
            return False
        if bh.in_dim != 0 or not bh.sa3.group_all:
            return False
        # q1 / q2 share one kNN search and the layer-1 GEMM: same neighbourhood sizes and widths in both modules
        if list(net.q1.nsample_list) != list(net.q2.nsample_list):
            return False
        w = [tuple(int(c.weight.shape[0]) for c in convs) for mod in (net.q1, net.q2) for convs in mod.conv_blocks]
        return len(set(w)) == 1

    def prepare(self, force: bool = False):
        """Fold BatchNorm into the convolutions and pre-arrange the weights (cached; refreshed when any
        parameter / buffer changed in place)."""
        key = self._versions()
        if not force and self._key == key:
            return self.P
        # NaN contract of this path (tests/test_gpu_fused.py::test_fast_path_nan_contract): the fused kernels drop NaNs in
        # their ReLU / max-pool maxima, so (a) non-finite WEIGHTS (a diverged checkpoint) are detected here, once per weight
        # change, and the network then runs the module path, which propagates them like torch / the reference;
        # (b) non-finite INPUT frames are flagged on the device by the hand-frame kernel and get NaN keypoints from the head.
        with torch.no_grad():
            ts = [t for t in list(self.net.parameters()) + list(self.net.buffers()) if t.is_floating_point()]
            self.finite_weights = bool(torch.stack([torch.isfinite(t).all() for t in ts]).all().item()) if ts else True
        if not self.finite_weights:
            self.P, self._key = None, key
            return None
        from hotrack_amd import gemm_tuning
        from hotrack_amd.fused import fold_conv_bn as fold
        gemm_tuning.enable()  # load the gfx950 solution table for the library GEMMs (no-op if absent / disabled)
        net, bh = self.net, self.net.bhand
        P = {}

        def msg(mod, i):
            return [fold(c, n) for c, n in zip(mod.conv_blocks[i], mod.bn_blocks[i])]

        (W1, b1), l2, l3 = msg(bh.sa1, 0)
        P["sa1"] = dict(wx=W1.contiguous(), b1=b1, l2=l2, l3=l3)
        (W1, b1), l2, l3 = msg(bh.sa2, 0)
        D = W1.shape[1] - 3
        P["sa2"] = dict(w1f=W1[:, :D].contiguous(), wx=W1[:, D:].contiguous(), b1=b1, l2=l2, l3=l3)
        sa3 = [fold(c, n) for c, n in zip(bh.sa3.mlp_convs, bh.sa3.mlp_bns)]  # reference input = [xyz | feat]
        W = sa3[0][0]
        sa3[0] = (torch.cat([W[:, 3:], W[:, :3]], dim=1).contiguous(), sa3[0][1])  # -> [feat | xyz] (aligned block first)
        P["sa3"] = sa3
        fp3 = [fold(c, n) for c, n in zip(bh.fp3.mlp_convs, bh.fp3.mlp_bns)]
        c_l2 = bh.sa2.out_channel
        P["fp3"] = dict(wa=fp3[0][0][:, :c_l2].contiguous(), wb=fp3[0][0][:, c_l2:].contiguous(), b=fp3[0][1], rest=fp3[1:])
        P["fp2"] = [fold(c, n) for c, n in zip(bh.fp2.mlp_convs, bh.fp2.mlp_bns)]  # in = [l1_feat | interp]
        fp1 = [fold(c, n) for c, n in zip(bh.fp1.mlp_convs, bh.fp1.mlp_bns)]  # in = [xyz | interp]
        W = fp1[0][0]
        fp1[0] = (torch.cat([W[:, 3:], W[:, :3]], dim=1).contiguous(), fp1[0][1])  # -> [interp | xyz] (aligned block first)
        P["fp1"] = fp1
        # Two consecutive per-point layers whose weights fit the register file run as ONE launch (pn2x_mlp2_rows: the
        # intermediate activation stays on chip): fp1 = [interp | xyz] -> c -> c.  Measured on MI355X against the two tuned library
        # GEMMs: 64-cloud step 1.031 -> 1.022 ms single-stream, 0.735 -> 0.733 ms with four batches in flight (with only two
        # streams sharing the chip it lost 0.7 %: the persistent kernel holds every CU's LDS for its whole run).  HOTRACK_MLP2=0
        # restores the GEMM pair.
        import os
        from hotrack_amd import ext
        P["fp1_fused"] = None
        c_in = fp1[0][0].shape[1] - 3
        if os.environ.get("HOTRACK_MLP2", "1") != "0" and len(fp1) == 2 and c_in == P["fp2"][-1][0].shape[0] and ext.mlp2_rows_supported(c_in, fp1[0][0].shape[0], fp1[1][0].shape[0]):
            P["fp1_fused"] = dict(w2=fp1[0][0][:, :c_in].contiguous(), w2e=fp1[0][0][:, c_in:].contiguous(), b2=fp1[0][1],
                                  w3=fp1[1][0].contiguous(), b3=fp1[1][1])
        P["conv1"] = fold(bh.conv1, bh.bn1)
        C = bh.out_dim
        wq, q = [], {}
        for name, mod in (("q1", net.q1), ("q2", net.q2)):
            for i in range(2):
                (W1, b1), l2, l3 = msg(mod, i)
                wq.append(W1[:, :C])
                q[(name, i)] = dict(wx=W1[:, C:C + 3].contiguous(), b1=b1, l2=l2, l3=l3, K=mod.nsample_list[i],
                                    wc=W1[:, C + 3:] if W1.shape[1] > C + 3 else None)
        P["q"] = q
        # Layer 1 of a keypoint branch is linear in the per-point feature, so it is a GEMM over POINTS for branches
        # whose J*K neighbour slots outnumber the points (K = 64: 1344 slots > 1024 points) and a GEMM over the
        # GATHERED slots for the small ones (K = 16: 336 rows instead of 1024).  Branch order inside each block:
        # (q1, i), (q2, i) for the i's of that kind.
        P["wq"] = wq  # per branch, order q1s0, q1s1, q2s0, q2s1
        P["wc2"] = torch.cat([q[("q2", 0)]["wc"], q[("q2", 1)]["wc"]], dim=0).contiguous()  # (2*128, C)
        P["head_w"] = net.final_mlp[2].weight.detach().squeeze(-1).contiguous()  # (3, 256)
        P["r1"] = (net.r1.linear.weight.detach().squeeze(-1), net.r1.linear.bias.detach(), net.r1._perm.t().contiguous())
        P["r2"] = (net.r2.linear.weight.detach().squeeze(-1), net.r2.linear.bias.detach(), net.r2._perm.t().contiguous())
        self.P, self._key = P, key
        return P

    def _wcat(self, i):
        """Cache of the per-scale layer-1 feature weights [q1 scale i | q2 scale i] (lives in P: rebuilt with the parameters)."""
        c = self.P.setdefault("_wcat", {})
        if i not in c:
            if isinstance(i, tuple):  # several scales stacked (one GEMM over the points for all of them)
                c[i] = torch.cat([self._wcat(j) for j in i], dim=0).contiguous()
            else:
                c[i] = torch.cat([self.P["wq"][i], self.P["wq"][2 + i]], dim=0).contiguous()
        return c[i]

    def _perm_idx(self, perm, B):
        """(B, J*re) int32 row indices of rearrange_module's token gather (perm (J, re)), cached per batch size."""
        key = ("perm", B, str(perm.device))
        if key not in self._idents:
            self._idents[key] = perm.reshape(1, -1).to(torch.int32).expand(B, -1).contiguous()
        return self._idents[key]

    def _ident(self, B, J, K, dev):
        """(B, J, K) int32 identity neighbour index for slot-major gathered rows (row j*K + k of each cloud)."""
        key = (B, J, K, str(dev))
        if key not in self._idents:
            self._idents[key] = torch.arange(J * K, dtype=torch.int32, device=dev).view(1, J, K).expand(B, J, K).contiguous()
        return self._idents[key]

    @staticmethod
    def _q_scales(ext, name, plan, q, xyz1, c1q, a_col, cadd, out, c_q):
        """The scales of one keypoint-query module: both in ONE persistent launch when there are two (ext.sa_mlp_max_pair,
        which itself falls back to one launch per scale for combinations its kernel does not cover)."""
        probs = []
        for i, (idx, a, nb_xyz) in enumerate(plan):
            p = q[(name, i)]
            probs.append(dict(idx=idx, w2=p["l2"][0], b2=p["l2"][1], w3=p["l3"][0], b3=p["l3"][1], a1f=a[:, :, a_col:a_col + c1q],
                              xyz=nb_xyz, cxyz=xyz1, wx=p["wx"], b1=p["b1"],
                              cadd=None if cadd is None else cadd[:, :, i * c1q:(i + 1) * c1q], out=out[:, :, i * c_q:(i + 1) * c_q]))
        if len(probs) == 2:
            ext.sa_mlp_max_pair(*probs)
        else:
            for p in probs:
                ext.sa_mlp_max(p.pop("idx"), p.pop("w2"), p.pop("b2"), p.pop("w3"), p.pop("b3"), **p)

    # ------------------------------------------------------------------------------------
    def forward(self, input, flag_dict):
        from hotrack_amd import gemm_tuning
        # recorded GEMM solutions for this forward only; the FIRST eager forward of a batch shape may tune the shapes the shipped table
        # does not hold when the user asks for it (HOTRACK_TUNE_GEMMS=1; a forward under stream capture never tunes)
        shape = tuple(input["hand_points"].shape[:2])
        first = shape not in self._shapes_seen
        self._shapes_seen.add(shape)
        with gemm_tuning.scope(tune=first):
            return self._forward(input, flag_dict)

    def _forward(self, input, flag_dict):
        G = self._geometry(input, flag_dict)
        return None if G is None else self._dense(G, flag_dict)

    def _geometry(self, input, flag_dict):
        from hotrack_amd import ext
        from hotrack_amd import pointnet2_utils as ops
        net = self.net
        P = self.prepare()
        if P is None:
            return None
        dev = net.device
        if flag_dict["track_flag"]:
            palm = input["pred_palm_template"]
        else:
            palm = input["gt_hand_pose"]["palm_template"]
        palm = palm.to(dev).float()
        kp = input["jittered_hand_kp"].to(dev).float()  # (B,21,3)
        pts = input["hand_points"].to(dev).float()  # (B,N,3)
        B, N, _ = pts.shape
        J = kp.shape[1]
        f32 = dict(dtype=torch.float32, device=pts.device)

        # ---- hand frame: R^T (x - t) / s, written for row vectors ------------------------------
        if net.handframe != "kp":
            raise NotImplementedError("fast path covers handframe='kp' (HandTrackNet's configuration)")
        if kp.shape[1] != 21:
            raise NotImplementedError("fast path covers the 21-keypoint hand")
        palm_idx = self._palm_idx(pts.device)
        # Kabsch (palm template -> jittered palm keypoints) + canonicalisation of cloud and keypoints: one launch
        # fp1's input rows [interp(l1 -> l0) | xyz | pad] exist from the start: the hand-frame kernel drops its xyz copy there
        c_i = P["fp2"][-1][0].shape[0]
        fp1_in = torch.empty((B, N, c_i + 4), **f32)
        nonfinite = torch.empty(B, dtype=torch.int32, device=pts.device)  # per-frame NaN / Inf flag (see prepare())
        R, t, xyz2, xyz1 = ext.hand_frame(palm.contiguous(), kp.contiguous(), palm_idx, pts.contiguous(), 0.2,
                                          xyz2_copy=fp1_in[:, :, c_i:c_i + 3], nonfinite=nonfinite)
        scale = self._scale(pts.device)
        canon = {"scale": scale, "rotation": R, "translation": t}
        tt = t.transpose(1, 2)

        bh = net.bhand
        # ---- sa1: 1024 -> 256 centroids, r = 0.1, K = 32, MLP [3 -> 32 -> 32 -> 64] ------------------
        p = P["sa1"]
        S1, K1 = bh.sa1.npoint, bh.sa1.nsample_list[0]
        # both sampling levels at once: level 2 (FPS over level 1's samples) is level 1's prefix unless an arg-max tied
        if self.two_level_fps:
            # sampling level 1 -> ball query level 1 (which also emits the centroids' coordinates) -> tie check ->
            # sampling level 2 (a no-op launch unless an arg-max tied): no gather launches
            # ... and the keypoints' kNN lists (sorted by (distance, index): the K = 16 list is the prefix of the K = 64 list) ride
            # in the launch of sampling level 1, which keeps one compute unit per cloud busy and nothing else
            qK = [P["q"][("q1", i)]["K"] for i in range(2)]
            knn_req = (xyz1, max(qK), min(qK) if min(qK) < max(qK) else 0)
            _, l1_xyz, i_l2, idx1, knn_lists = ext.fps_two_level(xyz2, S1, bh.sa2.npoint, query=(bh.sa1.radius_list[0], K1), knn=knn_req)
        else:
            knn_lists = None
            l1_xyz = ext.gather_rows(xyz2, ops.furthest_point_sample(xyz2, S1))
            i_l2 = ops.furthest_point_sample(l1_xyz, bh.sa2.npoint)
            idx1 = ops.ball_query(bh.sa1.radius_list[0], K1, xyz2, l1_xyz)
        return dict(P=P, pts=pts, B=B, N=N, J=J, c_i=c_i, fp1_in=fp1_in, nonfinite=nonfinite, R=R, t=t, xyz2=xyz2, xyz1=xyz1,
                    canon=canon, S1=S1, K1=K1, l1_xyz=l1_xyz, i_l2=i_l2, idx1=idx1, knn_lists=knn_lists)

    def _dense(self, G, flag_dict):
        from hotrack_amd import ext
        from hotrack_amd import pointnet2_utils as ops
        net, bh = self.net, self.net.bhand
        P, pts, B, N, J, c_i, fp1_in, nonfinite = G["P"], G["pts"], G["B"], G["N"], G["J"], G["c_i"], G["fp1_in"], G["nonfinite"]
        R, t, xyz2, xyz1, canon, S1, K1 = G["R"], G["t"], G["xyz2"], G["xyz1"], G["canon"], G["S1"], G["K1"]
        l1_xyz, i_l2, idx1, knn_lists = G["l1_xyz"], G["i_l2"], G["idx1"], G["knn_lists"]
        f32 = dict(dtype=torch.float32, device=pts.device)
        dev = net.device
        p = P["sa1"]
        c_l1 = p["l3"][0].shape[0]
        fp2_w = P["fp2"][0][0].shape[1]
        fp2_in = torch.empty((B, S1, fp2_w), **f32)  # [l1_feat | interp(l2 -> l1)]
        l1_feat = fp2_in[:, :, :c_l1]
        ext.sa_mlp_max(idx1, *p["l2"], *p["l3"], xyz=xyz2, cxyz=l1_xyz, wx=p["wx"], b1=p["b1"], out=l1_feat)

        # ---- sa2: 256 -> 128, r = 0.2, K = 32, MLP [64+3 -> 64 -> 64 -> 128] ---------------------------
        p = P["sa2"]
        S2, K2 = bh.sa2.npoint, bh.sa2.nsample_list[0]
        c_l2 = p["l3"][0].shape[0]
        sa3_in = torch.empty((B, S2, c_l2 + 4), **f32)  # [l2_feat | l2_xyz | pad]: sa3's group-all input, no torch.cat
        idx2, l2_xyz = ext.ball_query_picks(bh.sa2.radius_list[0], K2, l1_xyz, i_l2, xyz_copy=sa3_in[:, :, c_l2:c_l2 + 3])
        a1f = _lin(l1_feat.reshape(B * S1, c_l1), p["w1f"]).view(B, S1, -1)
        l2_feat = sa3_in[:, :, :c_l2]
        ext.sa_mlp_max(idx2, *p["l2"], *p["l3"], a1f=a1f, xyz=l1_xyz, cxyz=l2_xyz, wx=p["wx"], b1=p["b1"], out=l2_feat)

        # ---- sa3: group-all [feat | xyz] (weights permuted to match) -> MLP -> max over the 128 points -----------
        x = sa3_in.view(B * S2, c_l2 + 4)[:, :c_l2 + 3]
        for W, b in P["sa3"]:
            x = _lin_relu(x, W, b)
        l3 = ext.max_rows(x.view(B, S2, -1))  # (B,512)

        # ---- fp3: S == 1 -> the global feature is broadcast; first layer split so it is applied once per cloud
        p = P["fp3"]
        g = _lin(l3, p["wb"], p["b"])  # (B,256) per-cloud half, bias included
        if B == 1:  # one cloud: the per-cloud half IS the layer's bias vector (one launch less in the tracking loop; same operations)
            h = _lin_relu(sa3_in.view(S2, c_l2 + 4)[:, :c_l2], p["wa"], g.view(-1)).view(B, S2, -1)
        else:
            h = _lin(sa3_in.view(B * S2, c_l2 + 4)[:, :c_l2], p["wa"]).view(B, S2, -1)
            ext.bias_act_pm_(h, g, rows_per_bias=S2, relu=True)
        x = h.view(B * S2, -1)
        for W, b in p["rest"]:
            x = _lin_relu(x, W, b)
        l2_out = x.view(B, S2, -1)

        # ---- fp2: interpolate l2 -> l1, [l1_feat | interp] -> MLP ---------------------------------------
        ext.three_nn_interpolate_pm(l1_xyz, l2_xyz, l2_out, fp2_in[:, :, c_l1:])  # search + blend: one launch
        x = fp2_in.view(B * S1, fp2_w)
        for W, b in P["fp2"]:
            x = _lin_relu(x, W, b)
        l1_out = x.view(B, S1, -1)

        # ---- fp1: interpolate l1 -> l0, [interp | xyz] (weights permuted to match) -> MLP; conv1 ----------
        assert c_i == l1_out.shape[2]
        ext.three_nn_interpolate_pm(xyz2, l1_xyz, l1_out, fp1_in[:, :, :c_i])
        f = P["fp1_fused"]
        if f is not None:  # both fp1 layers in one launch (pn2x_mlp2_rows): rows [interp | xyz | pad] -> 128 -> 128
            x = ext.mlp2_rows(fp1_in.view(B * N, c_i + 4), f["w2"], f["b2"], f["w3"], f["b3"], w2e=f["w2e"])
        else:
            x = fp1_in.view(B * N, c_i + 4)[:, :c_i + 3]
            for W, b in P["fp1"]:
                x = _lin_relu(x, W, b)
        src2 = _lin_relu(x, *P["conv1"])  # (B*N, C) per-point backbone features
        C = src2.shape[1]

        # ---- q1 / q2: kNN (16 / 64) neighbourhoods of the 21 keypoints ------------------------------------
        q = P["q"]
        # kNN lists are sorted by (distance, index): the K=16 list is the prefix of the K=64 list -> one search
        Ks = [q[("q1", i)]["K"] for i in range(2)]
        kmin, kmax = min(Ks), max(Ks)
        if knn_lists is not None:
            gi, gi_small = knn_lists
        else:
            gi, gi_small = ext.knn_indices(kmax, xyz1, xyz2, k2=kmin) if kmin < kmax else (ext.knn_indices(kmax, xyz1, xyz2), None)
        c_q = q[("q1", 0)]["l3"][0].shape[0]
        c1q = q[("q1", 0)]["l2"][0].shape[1]
        src3 = src2.view(B, N, C)
        # per scale i: neighbour index, feature rows the layer-1 GEMM runs over, the coordinates that go with them
        plan = []
        # few slots and a batch large enough that the GEMM, not the launch count, is what costs (measured: pays from B ~ 32)
        gathered = [J * K * 2 <= N and B * N >= 32768 for K in Ks]
        over_points = [i for i, gth in enumerate(gathered) if not gth]
        a_all = None
        if len(over_points) > 1:  # small batch: the scales' per-point GEMMs read the same rows -> one GEMM, column blocks per scale
            a_all = _lin(src2, self._wcat(tuple(over_points))).view(B, N, -1)
        for i, K in enumerate(Ks):
            W = self._wcat(i)  # (2*c1q, C): layer-1 feature weights of q1 | q2 at this scale
            if gathered[i]:  # gather the J*K feature rows, then the GEMM (slot-major rows, identity index)
                flat = (gi_small if K == kmin and gi_small is not None else gi[:, :, :K].contiguous()).view(B, J * K)
                a = _lin(ext.gather_rows(src3, flat).view(B * J * K, C), W).view(B, J * K, -1)
                plan.append((self._ident(B, J, K, dev), a, ext.gather_rows(xyz2, flat)))
            else:
                idx = gi if K == kmax else (gi_small if K == kmin and gi_small is not None else gi[:, :, :K].contiguous())
                if a_all is not None:
                    j = over_points.index(i)
                    a = a_all[:, :, j * W.shape[0]:(j + 1) * W.shape[0]]
                else:
                    a = _lin(src2, W).view(B, N, -1)
                plan.append((idx, a, xyz2))
        f11 = torch.empty((B, J, 2 * c_q), **f32)
        self._q_scales(ext, "q1", plan, q, xyz1, c1q, 0, None, f11, c_q)
        Wr, br, perm = P["r1"]
        f12 = _lin(ext.gather_rows(f11, self._perm_idx(perm, B)).view(B * J, -1), Wr, br)  # (B*J, C)
        cadd = _lin(f12, P["wc2"]).view(B, J, -1)
        f13 = torch.empty((B, J, 2 * c_q), **f32)
        self._q_scales(ext, "q2", plan, q, xyz1, c1q, c1q, cadd, f13, c_q)
        Wr, br, perm = P["r2"]
        f14 = _lin(ext.gather_rows(f13, self._perm_idx(perm, B)).view(B * J, -1), Wr, br)

        # ---- "TransT" with attn=False: only LayerNorms and FFNs are live; the element-wise runs between the GEMMs
        # (residual add, bias, one or two LayerNorms) are one launch each --------------------------------------
        s11, c11, c3 = net.transt.s11, net.transt.c11, net.c3
        wf, bf = net.final_mlp[0].weight.squeeze(-1), net.final_mlp[0].bias
        if ext.ln_linear_supported(f14.shape[0], f14.shape[1]):
            # few tokens (the tracking loop, B <= 12): every launch here sits at the floor of a graph node, so each LayerNorm launch rides in
            # the Linear that consumes it (pn2x_ln_linear_small: same bits as the two launches) -- 9 launches -> 6
            x, hdn = ext.ln_linear(f14, s11.norm1, c11.linear1.weight, c11.linear1.bias, relu=True, ln2=c11.norm1)
            y = _lin(hdn, c11.linear2.weight)
            x, hdn = ext.ln_linear(x, c11.norm2, c3.linear1.weight, c3.linear1.bias, relu=True, y=y, ybias=c11.linear2.bias, ln2=c3.norm1)
            y = _lin(hdn, c3.linear2.weight)
            x, hdn = ext.ln_linear(x, c3.norm2, wf, bf, relu=True, y=y, ybias=c3.linear2.bias)
        else:
            x = ext.add_layernorm(f14, s11.norm1, ln2=c11.norm1)
            for blk, nxt in ((c11, c3.norm1), (c3, None)):
                hdn = _lin_relu(x, blk.linear1.weight, blk.linear1.bias)
                x = ext.add_layernorm(x, blk.norm2, y=_lin(hdn, blk.linear2.weight), bias=blk.linear2.bias, ln2=nxt)
            hdn = _lin_relu(x, wf, bf)
        # head: last 1x1 conv + residual on the initial keypoints + back to the camera frame, one launch
        pred_hf, pred_kp = ext.pose_head(hdn, P["head_w"], net.final_mlp[2].bias, xyz1, R, t, 0.2, nonfinite=nonfinite)

        ret = {"canon_pose": canon}
        ret["pred_kp_handframe"] = pred_hf.transpose(1, 2)
        ret["init_kp_handframe"] = xyz1.transpose(1, 2)
        ret["points_handframe"] = xyz2.transpose(1, 2)
        ret["pred_kp"] = pred_kp
        if flag_dict.get("IKNet_flag", False):
            d4, _ = ops.knn(4, ret["pred_kp"].contiguous(), pts.contiguous())
            d4 = d4.mean(dim=-1)
            d4[:, 0] -= 0.01
            d4[:, 1] -= 0.01
            ret["pred_kp_vis_mask"] = (d4 < 0.02).bool()
        return ret
