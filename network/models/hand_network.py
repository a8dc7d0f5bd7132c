"""HandTrackNet on MI355X (counterpart of the reference's hand_network.py:40-221).

Same constructor (`HandTrackNet(cfg)`), same input / output dictionaries, same parameter names
(bhand, r1, r2, q1, q2, transt, c3, final_mlp) so reference checkpoints load unchanged.
What differs from the reference:
  * the point operators run on the hand-written gfx950 kernels (via models/pointnet_utils.py);
  * the palm alignment (Kabsch) runs on the device (no CPU SVD hop per forward);
  * `elide_dead_attention` (default True): the multi-head attention results that the
    reference computes and then discards (attn=False, transformer.py:72-82), the sine position
    embedding that only feeds them, and TransT.s12 / TransT.c12 whose outputs are never read
    are skipped.  Eval outputs are bit-identical either way (tests/test_network.py);
  * no MANO / IKNet (needs licensed assets; out of scope).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .backbones import PointNet2Msg_fast
from .blocks import rearrange_module
from .hand_utils import canonicalize, decanonicalize, handkp2palmkp, ransac_rt
from . import pointnet_utils
from .pointnet_utils import PointNetSetAbstractionMsg_GivenCenterPoints, knn_point
from .transformer import PositionEmbeddingSine, TransT, attn_module


def L2_loss(x, y, mask=None):
    """Mean Euclidean error over (B,3,K) tensors (optionally masked (B,1,K))."""
    assert x.shape[1] == 3 and y.shape[1] == 3
    if mask is None:
        return (x - y).norm(dim=1).mean()
    assert mask.shape[1] == 1
    return (((x - y) * mask).norm(dim=1).sum(dim=-1) / torch.clamp(mask.sum(dim=-1), min=1).squeeze()).mean()


def L1_loss(x, y, mask=None, check_dim_in=3):
    """Mean absolute error over (B,D,K) tensors (optionally masked (B,1,K))."""
    assert x.shape[1] == check_dim_in and y.shape[1] == check_dim_in
    if mask is None:
        return (x - y).abs().mean()
    assert mask.shape[1] == 1
    return (((x - y) * mask).abs().mean(dim=1).sum(dim=-1) / torch.clamp(mask.sum(dim=-1), min=1).squeeze()).mean()


def _conv1x1(conv: nn.Conv1d, x: torch.Tensor, fast: bool = True) -> torch.Tensor:
    """Conv1d(kernel 1) on (B,C,J) with J = 21 keypoints as one token-major GEMM with the bias in its
    epilogue (F.linear) instead of a convolution-library call; same arithmetic, differentiable."""
    return F.linear(x.transpose(1, 2), conv.weight.squeeze(-1), conv.bias).transpose(1, 2)


def _rot_angle_deg(R: torch.Tensor) -> torch.Tensor:
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    return torch.mean(torch.acos(torch.clamp((tr - 1) / 2, min=-1, max=1))) * 180 / math.pi


_CONSTS = {}


def _palm_idx(device):
    key = ("palm", str(device))
    if key not in _CONSTS:
        _CONSTS[key] = torch.tensor([0, 1, 5, 9, 13, 17], dtype=torch.int32, device=device)  # hand_utils.handkp2palmkp
    return _CONSTS[key]


def _scale_const(device):
    key = ("scale", str(device))
    if key not in _CONSTS:
        _CONSTS[key] = 0.2 * torch.ones(1, device=device)
    return _CONSTS[key]


class LossDict(dict):
    """The loss dictionary, plus (attribute, not an entry) the (9,) tensor all its values are views of when they come from the
    fused loss kernel -- Trainer.summarize_losses then forms the weighted total with one dot product."""
    fused_values = None
    fused_total = None          # sum_i w_i value_i formed inside the loss kernel (when the Trainer handed its weights to the model)
    fused_total_weights = None  # ... and the (9,) weight tensor it was formed with


class HandTrackNet(nn.Module):
    def __init__(self, cfg, elide_dead_attention: bool = True):
        super().__init__()
        self.device = cfg["device"]
        self.handframe = cfg["network"]["handframe"]
        C = cfg["network"]["backbone_out_dim"]
        assert C % 6 == 0
        self.elide_dead_attention = elide_dead_attention
        self.use_fast_eval = True  # eval + fused backend + GPU -> models/fast_eval.py (set False to force this file's path)
        self._fast = None
        self.use_fast_train = True  # training on the GPU -> models/fast_train.py (point-major GEMM + fused BatchNorm/ReLU kernels)
        self.use_fused_losses = True  # GPU: compute_loss's dictionary and its gradient as two launches (hotrack_amd.ext.HandLosses)
        self.use_fast_tail = True  # training on the GPU: the 21-token tail with fused element-wise runs (fast_train.FastTail)
        self._ftail = None
        self._ftrain = None
        # Segmented backward (network/trainer.py, dp = flat): when set, a TRAINING forward detaches the backbone's output
        # features (the only tensor that joins the backbone to everything after it) and leaves (backbone output, detached leaf)
        # in `backward_cut`; the caller then runs loss.backward() -- which stops at the leaf: gradients of q1 / r1 / q2 / r2 /
        # tail are complete -- and later backbone_output.backward(leaf.grad).  Same gradients as one pass; the exchange of the
        # first segment's gradients can travel while the backbone's backward runs.
        self.cut_backbone_grad = False
        self.backward_cut = None
        self.bhand = PointNet2Msg_fast(cfg, C)
        self.r1 = rearrange_module(channel=C)
        self.r2 = rearrange_module(channel=C)
        self.positionEmbedding = PositionEmbeddingSine(num_pos_feats=C // 6)
        widths = [[128, 128, C // 2], [128, 128, C // 2]]
        # radius_list is ignored because knn=True (16 / 64 nearest points of each keypoint)
        self.q1 = PointNetSetAbstractionMsg_GivenCenterPoints([0.2, 0.2], [16, 64], widths, in_channel=C + 3, knn=True)
        self.q2 = PointNetSetAbstractionMsg_GivenCenterPoints([0.2, 0.2], [16, 64], widths, in_channel=2 * C + 3, knn=True)
        self.transt = TransT(d_model=C)
        self.c3 = attn_module(d_model=C)
        self.final_mlp = nn.Sequential(nn.Conv1d(C, 256, 1), nn.ReLU(inplace=True), nn.Conv1d(256, 3, 1))

    # ------------------------------------------------------------------------------------
    def _hand_frame(self, palm_template, jittered_kp, hand_points):
        dev = hand_points.device
        scale = 0.2 * torch.ones(1, device=dev)
        if self.handframe == "kp":
            R, t, _, _, _ = ransac_rt(palm_template, handkp2palmkp(jittered_kp))
        elif self.handframe == "camera":
            b = hand_points.shape[0]
            R = torch.eye(3, device=dev).unsqueeze(0).repeat(b, 1, 1)
            t = torch.zeros((b, 3, 1), device=dev)
        else:
            raise NotImplementedError(self.handframe)
        return {"scale": scale, "rotation": R, "translation": t}

    def _fast_train_frame_ok(self, hand_points, palm_template, kp_num, use_ft) -> bool:
        return bool(self.training and use_ft and self.handframe == "kp" and self.elide_dead_attention and hand_points.is_cuda
                    and kp_num == 21 and palm_template.shape[-2] == 6 and pointnet_utils.hip_backend_active())

    def cut_after_backbone(self, feat):
        """The backbone's output features, detached into a fresh leaf when a segmented backward was asked for (see __init__)."""
        self.backward_cut = None
        if self.cut_backbone_grad and self.training and torch.is_grad_enabled() and feat.requires_grad:
            leaf = feat.detach().requires_grad_(True)
            self.backward_cut = (feat, leaf)
            return leaf
        return feat

    def segment_upstream_parameters(self):
        """Parameters whose gradients are complete only after the SECOND segment of a cut backward (the backbone's)."""
        return list(self.bhand.parameters())

    def precompute_geometry(self, input, flag_dict):
        """The part of a TRAINING step on the GPU that depends on the batch only, not on the parameters: the hand frame (Kabsch of
        the palm template, canonicalised cloud and keypoints: hand_utils.py:30-66) and FastTrain.geometry (sampling, ball query,
        three-NN, kNN, inverted neighbour lists).  Returns a dict to hand back as input["_geometry"] with the same batch, or None
        when this configuration does not run the point-major training path.  No gradient, no parameter, no state: a training
        loop runs it for batch t+1 on a second stream while batch t's dense work occupies the matrix cores
        (network/trainer.py)."""
        use_ft = getattr(self, "_force_fast_train", self.use_fast_train)
        dev = self.device
        palm_template = (input["pred_palm_template"] if flag_dict["track_flag"] else input["gt_hand_pose"]["palm_template"]).to(dev).float()
        jittered_kp = input["jittered_hand_kp"].to(dev).float()
        hand_points = input["hand_points"].to(dev).float()
        if not self._fast_train_frame_ok(hand_points, palm_template, jittered_kp.shape[1], use_ft):
            return None
        if self._ftrain is None:
            from .fast_train import FastTrain
            self._ftrain = FastTrain(self) if FastTrain.supported(self) else False
        if not self._ftrain:
            return None
        from hotrack_amd import ext
        with torch.no_grad():
            frame = ext.hand_frame(palm_template.contiguous(), jittered_kp.contiguous(), _palm_idx(hand_points.device),
                                   hand_points.contiguous(), 0.2)
            geo = self._ftrain.geometry(frame[2], frame[3], with_inverse=True)
        geo["frame"] = tuple(frame)
        # the keypoints channel-major (B,3,kp), contiguous: what the pose head and the loss kernel read -- a batch-only copy
        # that otherwise sits in the dense step (one 5 us launch in front of the tail)
        geo["xyz1_cm"] = frame[3].transpose(1, 2).contiguous()
        return geo

    def forward(self, input, flag_dict):
        """input: hand_points (B,N,3), jittered_hand_kp (B,21,3), palm template (gt_hand_pose.palm_template
        or pred_palm_template when tracking).  Returns the reference's ret_dict (pred_kp (B,21,3), ...)."""
        if (self.use_fast_eval and pointnet_utils.fused_backend() is not None and not self.training
                and not torch.is_grad_enabled() and self.elide_dead_attention and self.handframe == "kp"
                and torch.device(self.device).type == "cuda"):
            if self._fast is None:
                from .fast_eval import FastEval
                self._fast = FastEval(self) if FastEval.supported(self) else False
            # configurations / sizes the point-major path does not cover use the module path below (same results)
            if (self._fast is not False and input["jittered_hand_kp"].shape[1] == 21
                    and input["hand_points"].shape[1] <= self._fast.MAX_POINTS):
                out = self._fast.forward(input, flag_dict)
                if out is not None:
                    return out
                # None: non-finite weights (diverged checkpoint).  The fused kernels drop NaNs in their maxima, so this forward
                # runs the unfused module path, which propagates them like torch / the reference.
                saved = pointnet_utils.fused_backend()
                pointnet_utils.set_fused_backend(None)
                try:
                    return self.forward(input, flag_dict)
                finally:
                    pointnet_utils.set_fused_backend(saved)
        dev = self.device
        if flag_dict["track_flag"]:
            palm_template = input["pred_palm_template"]
        else:
            palm_template = input["gt_hand_pose"]["palm_template"]
        palm_template = palm_template.to(dev).float()
        jittered_kp = input["jittered_hand_kp"].to(dev).float()
        hand_points = input["hand_points"].to(dev).float()
        ret = {}

        kp_num = jittered_kp.shape[1]
        elide = self.elide_dead_attention
        cam = None
        use_ft = getattr(self, "_force_fast_train", self.use_fast_train)  # class-level override: tests compare the two paths
        geo = input.get("_geometry") if self.training and use_ft else None  # precompute_geometry() of THIS batch (trainer prefetch)
        if geo is not None:
            R, t, xyz2_pm, xyz1_pm = geo["frame"]
            canon_pose = {"scale": _scale_const(hand_points.device), "rotation": R, "translation": t}
            xyz2, xyz1 = xyz2_pm.transpose(1, 2), xyz1_pm.transpose(1, 2)
            if geo.get("xyz1_cm") is not None:
                xyz1 = geo["xyz1_cm"]  # (already contiguous: the copy below is then a no-op)
        elif self._fast_train_frame_ok(hand_points, palm_template, kp_num, use_ft):
            # training on the GPU: the hand frame (Kabsch of the palm template + canonicalisation of cloud and keypoints) as ONE
            # launch, as in the inference path -- no gradient flows through it (inputs only); the torch composition below is
            # ~12 launches (gather, device Kabsch, cat, transposes, subtract, matmul, divide, two copies)
            from hotrack_amd import ext
            R, t, xyz2_pm, xyz1_pm = ext.hand_frame(palm_template.contiguous(), jittered_kp.contiguous(), _palm_idx(hand_points.device),
                                                    hand_points.contiguous(), 0.2)
            canon_pose = {"scale": _scale_const(hand_points.device), "rotation": R, "translation": t}
            xyz2, xyz1 = xyz2_pm.transpose(1, 2), xyz1_pm.transpose(1, 2)  # (B,3,N) / (B,3,kp) views of point-major buffers
        else:
            if self.handframe == "OBB":
                canon_pose = {k: v.to(dev).float() for k, v in input["OBB_pose"].items()}
            else:
                canon_pose = self._hand_frame(palm_template, jittered_kp, hand_points)
            cam = canonicalize(torch.cat([hand_points, jittered_kp], dim=1).transpose(2, 1), canon_pose)  # (B,3,N+kp)
            xyz2 = cam[..., :-kp_num].contiguous()  # hand points
            xyz1 = cam[..., -kp_num:].contiguous()  # keypoints
        ret["canon_pose"] = canon_pose

        pos1 = pos2 = None
        if not elide:
            pe = self.positionEmbedding(cam)
            pos2, pos1 = pe[..., :-kp_num], pe[..., -kp_num:]

        fast = (pointnet_utils.fused_backend() is not None and xyz2.is_cuda and not self.training
                and not torch.is_grad_enabled())
        ftrain = None
        if self.training and use_ft and xyz2.is_cuda and pointnet_utils.hip_backend_active():
            if self._ftrain is None:
                from .fast_train import FastTrain
                self._ftrain = FastTrain(self) if FastTrain.supported(self) else False
            ftrain = self._ftrain or None
        if ftrain is not None:  # point-major training path, same mathematics (tests/test_gpu_train.py)
            f14, src2_pm = ftrain.forward(xyz2, xyz1, geo=geo)
            src2 = None if elide else src2_pm.transpose(1, 2)
        else:
            xyz2, xyz1 = xyz2.contiguous(), xyz1.contiguous()
            src2 = self.cut_after_backbone(self.bhand(xyz2))  # (B,C,N)
            f11, group_idx = self.q1(xyz2, src2, xyz1, None, return_group_idx=True)
            f12 = self.r1(f11, True)
            f13 = self.q2(xyz2, src2, xyz1, f12, pre_group_idx=group_idx)
            f14 = self.r2(f13, True)
        ftail = None
        if ftrain is not None and elide and self.use_fast_tail:
            if self._ftail is None:
                from .fast_train import FastTail
                self._ftail = FastTail(self) if FastTail.supported(self) else False
            ftail = self._ftail or None
        if ftail is not None:  # the 21-token tail, token-major, fused element-wise runs (fast_train.FastTail)
            xyz1 = xyz1.contiguous()  # (B,3,kp) once: the pose head and the loss kernel both read it channel-major
            ret["pred_kp_handframe"], ret["pred_kp"] = ftail.forward(ftrain.last_token_rows, xyz1, canon_pose)
        else:
            f15, f251 = self.transt(src1=f14, pos1=pos1, src2=src2, pos2=pos2, attn=False, elide_dead=elide,
                                    need_result2=not elide)
            fused = self.c3(f15, pos1, f251, pos2, attn=False, elide_dead=elide)
            delta = _conv1x1(self.final_mlp[2], F.relu(_conv1x1(self.final_mlp[0], fused)))
            ret["pred_kp_handframe"] = delta + xyz1  # (B,3,kp)
            ret["pred_kp"] = decanonicalize(ret["pred_kp_handframe"], canon_pose).transpose(2, 1)
        ret["init_kp_handframe"] = xyz1
        ret["points_handframe"] = xyz2

        if flag_dict.get("IKNet_flag", False):
            d4, _ = knn_point(4, ret["pred_kp"].contiguous(), hand_points.contiguous())
            d4 = d4.mean(dim=-1)
            d4[:, 0] -= 0.01
            d4[:, 1] -= 0.01
            ret["pred_kp_vis_mask"] = (d4 < 0.02).bool()
        return ret

    # ------------------------------------------------------------------------------------
    def compute_loss(self, input, ret_dict, flag_dict):
        """Loss / metric dictionary of the reference (hand_network.py:159-221), minus the MANO term."""
        dev = self.device
        canon_pose = ret_dict["canon_pose"]
        if (self.use_fused_losses and self.handframe != "OBB" and "global_pose" not in ret_dict and not flag_dict["track_flag"]
                and ret_dict["pred_kp"].is_cuda and ret_dict["pred_kp"].shape[1] == 21 and pointnet_utils.hip_backend_active()):
            # the whole dictionary in one launch, its gradient in another (hotrack_amd.ext.HandLosses, csrc/kabsch.hip): the torch
            # composition below is ~75 launches of 4-5 us inside a captured training step
            from hotrack_amd import ext
            gt = input["gt_hand_kp"].to(dev).float()
            palm = input["gt_hand_pose"]["palm_template"].to(dev).float()
            s = float(0.2)  # _hand_frame / fast paths: the constant hand-frame scale (hand_network.py:95)
            # fused_loss_weights (9,), set by the Trainer: the weighted total comes out of the same launch
            wts = getattr(self, "fused_loss_weights", None)
            if wts is not None and (wts.device != ret_dict["pred_kp"].device or wts.numel() != 9):
                wts = None
            res = ext.HandLosses.apply(ret_dict["pred_kp_handframe"], ret_dict["init_kp_handframe"], gt, ret_dict["pred_kp"], canon_pose["rotation"],
                                       canon_pose["translation"].reshape(-1, 3), s, palm, wts)
            vals, total = res if wts is not None else (res, None)
            ret_dict["gt_kp_handframe"] = canonicalize(gt.transpose(-1, -2), canon_pose) if flag_dict.get("save_flag") else None
            loss = {k: vals[i] for i, k in enumerate(ext.HAND_LOSS_NAMES)}
            order = ["hand_pred_kp_loss", "hand_pred_kp_diff", "hand_init_kp_diff", "hand_pred_r_loss", "hand_pred_t_loss", "hand_init_r_diff",
                     "hand_init_t_diff", "hand_pred_r_diff", "hand_pred_t_diff"]
            loss = LossDict((k, loss[k]) for k in order)
            loss.fused_values = vals  # (9,) in ext.HAND_LOSS_NAMES order: lets the trainer form the weighted total in one go
            loss.fused_total, loss.fused_total_weights = total, wts
            return loss, ret_dict
        gt_kp = input["gt_hand_kp"].to(dev).float().transpose(-1, -2)  # (B,3,kp)
        pred_kp = ret_dict["pred_kp"].transpose(-1, -2)
        s = canon_pose["scale"][:, None, None]
        ret_dict["gt_kp_handframe"] = canonicalize(gt_kp, canon_pose)
        init_s = ret_dict["init_kp_handframe"] * s
        pred_s = ret_dict["pred_kp_handframe"] * s
        gt_s = ret_dict["gt_kp_handframe"] * s

        # Terms the optimiser differentiates first (cfg loss_weight: kp / r / t losses), then the metric-only terms.
        loss = {"hand_pred_kp_loss": L1_loss(pred_s, gt_s)}
        pose = None
        if self.handframe != "OBB":
            if "global_pose" in ret_dict:
                gt_R = input["gt_hand_pose"]["rotation"].to(dev).float().reshape(-1, 3, 3)
                gt_t = input["gt_hand_pose"]["translation"].to(dev).float().reshape(-1, 3, 1)
                R = ret_dict["global_pose"]["rotation"].reshape(-1, 3, 3)
                t = ret_dict["global_pose"]["translation"].reshape(-1, 3, 1)
            else:
                palm = input["gt_hand_pose"]["palm_template"].to(dev).float()
                gt_R, gt_t, _, _, _ = ransac_rt(palm, handkp2palmkp(gt_s.transpose(-1, -2)).contiguous())
                R, t, _, _, _ = ransac_rt(palm, handkp2palmkp(pred_s.transpose(-1, -2)).contiguous())
            loss["hand_pred_r_loss"] = L1_loss(R, gt_R)
            loss["hand_pred_t_loss"] = L1_loss(t, gt_t)
            pose = (R, t, gt_R, gt_t)

        def metrics():
            m = {"hand_pred_kp_diff": L2_loss(pred_kp, gt_kp), "hand_init_kp_diff": L2_loss(init_s, gt_s)}
            if pose is not None:
                R, t, gt_R, gt_t = pose
                if "global_pose" not in ret_dict:
                    m["hand_init_r_diff"] = _rot_angle_deg(gt_R)
                    m["hand_init_t_diff"] = gt_t.norm(dim=1).mean()
                m["hand_pred_r_diff"] = _rot_angle_deg(torch.matmul(R.transpose(-1, -2), gt_R))
                m["hand_pred_t_diff"] = L2_loss(t, gt_t)
            if flag_dict["track_flag"] and "rotation" in input.get("gt_hand_pose", {}):
                g_R = input["gt_hand_pose"]["rotation"].to(dev).float().reshape(-1, 3, 3)
                g_t = input["gt_hand_pose"]["translation"].to(dev).float().reshape(-1, 3, 1)
                cR = canon_pose["rotation"].reshape(-1, 3, 3)
                ct = canon_pose["translation"].reshape(-1, 3, 1)
                m["hand_canon_r_diff"] = _rot_angle_deg(torch.matmul(cR.transpose(-1, -2), g_R))
                m["hand_canon_t_diff"] = L2_loss(g_t, ct)
            return m

        with torch.no_grad():  # nothing differentiates them (and a side stream for them made the captured step slower: 5.02 -> 5.37 ms)
            m = metrics()
        # key order of the reference's dictionary
        order = ["hand_pred_kp_loss", "hand_pred_kp_diff", "hand_init_kp_diff", "hand_pred_r_loss", "hand_pred_t_loss", "hand_init_r_diff",
                 "hand_init_t_diff", "hand_pred_r_diff", "hand_pred_t_diff", "hand_canon_r_diff", "hand_canon_t_diff"]
        loss.update(m)
        loss = {k: loss[k] for k in order if k in loss}
        return loss, ret_dict
