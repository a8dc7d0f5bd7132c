"""Per-sequence hand tracking with HandTrackNet (counterpart of HandTrackModel.forward,
reference track_network.py:139-226, HandTrackNet-only branch :214-217).

Frame t is initialised from frame t-1: the previous prediction, expressed relative to the previous
cloud's centroid, is re-attached to the current cloud's centroid ("important for fast motion",
:163,:217).  The palm template comes from the sequence's first frame (the reference builds it from a
MANO layer, which needs licensed assets; the tracking logic is otherwise the same)."""
from __future__ import annotations

import torch
import torch.nn as nn


class HandTrackModel(nn.Module):
    def __init__(self, cfg, handnet, IKnet=None):
        super().__init__()
        if IKnet is not None or cfg.get("use_optimization", False):
            raise NotImplementedError("IKNet / particle optimisation need MANO + DeepSDF assets (out of scope)")
        self.device = cfg["device"]
        self.handnet = handnet(cfg)

    def forward(self, input, flag_dict):
        flag_dict["track_flag"] = True
        assert flag_dict["test_flag"]
        flag_dict["opt_flag"] = False
        palm_template = input[0]["gt_hand_pose"]["palm_template"].to(self.device).float()
        last_kp = None
        rets = []
        for data in input:
            data["pred_palm_template"] = palm_template
            centre = data["hand_points"].mean(dim=-2, keepdim=True).to(self.device).float()
            if last_kp is not None:
                data["jittered_hand_kp"] = last_kp + centre
            ret = self.handnet(data, flag_dict)
            last_kp = (ret["pred_kp"] - centre).clone()
            rets.append(ret)
        return rets

    def compute_loss(self, input, ret_dict_lst, flag_dict):
        total = {}
        for data, ret in zip(input, ret_dict_lst):
            loss, _ = self.handnet.compute_loss(data, ret, flag_dict)
            for k, v in loss.items():
                total[k] = total.get(k, 0.0) + float(v) / len(input)
        return total, ret_dict_lst
