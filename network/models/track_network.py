"""Per-sequence hand tracking with HandTrackNet (counterpart of HandTrackModel.forward,
reference track_network.py:139-226, HandTrackNet-only branch :214-217).

Frame t is initialised from frame t-1: the previous prediction, expressed relative to the previous
cloud's centroid, is re-attached to the current cloud's centroid ("important for fast motion",
:163,:217).  The palm template comes from the sequence's first frame (the reference builds it from a
MANO layer, which needs licensed assets; the tracking logic is otherwise the same)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import pointnet_utils


class HandTrackModel(nn.Module):
    def __init__(self, cfg, handnet, IKnet=None):
        super().__init__()
        if IKnet is not None or cfg.get("use_optimization", False):
            raise NotImplementedError("IKNet / particle optimisation need MANO + DeepSDF assets (out of scope)")
        self.device = cfg["device"]
        self.handnet = handnet(cfg)
        self.use_graph = True  # GPU + fused backend: one captured HIP graph per (N, keypoints) shape, replayed per frame
        self._graphs = {}

    # A captured graph bakes in the pointers of the BN-folded weights FastEval built at capture time: anything that can
    # change the weights (checkpoint load, fine-tuning, .to()/.float()) drops the captured graphs.
    def train(self, mode: bool = True):
        self._graphs.clear()
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self._graphs.clear()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._graphs.clear()
        return super().load_state_dict(*a, **k)

    def invalidate_graphs(self):
        """Call after editing parameters in place outside train() / load_state_dict() / .to()."""
        self._graphs.clear()

    # ------------------------------------------------------------------------------------
    def _graph_step(self, points, kp_init, palm_template, flag_dict):
        """One tracking step as a replay of a captured HIP graph (static input / output buffers): a frame is
        ~75 short kernels, so per-launch host cost would otherwise dominate the frame latency."""
        key = (tuple(points.shape), tuple(kp_init.shape), tuple(palm_template.shape), bool(flag_dict.get("IKNet_flag", False)))
        g = self._graphs.get(key)
        if g is None:
            buf = {"hand_points": points.clone(), "jittered_hand_kp": kp_init.clone(), "pred_palm_template": palm_template.clone()}
            with torch.no_grad():
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):  # warm-up: library handles, folded-weight caches, constant tensors
                        self.handnet(buf, flag_dict)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = self.handnet(buf, flag_dict)
            g = self._graphs[key] = (graph, buf, out)
        graph, buf, out = g
        buf["hand_points"].copy_(points, non_blocking=True)
        buf["jittered_hand_kp"].copy_(kp_init, non_blocking=True)
        buf["pred_palm_template"].copy_(palm_template, non_blocking=True)
        graph.replay()
        return {k: (v.clone() if torch.is_tensor(v) else {kk: vv.clone() for kk, vv in v.items()}) for k, v in out.items()}

    def forward(self, input, flag_dict):
        flag_dict["track_flag"] = True
        assert flag_dict["test_flag"]
        flag_dict["opt_flag"] = False
        palm_template = input[0]["gt_hand_pose"]["palm_template"].to(self.device).float()
        last_kp = None
        rets = []
        graph_ok = (self.use_graph and pointnet_utils.fused_backend() is not None and not self.training
                    and not torch.is_grad_enabled() and torch.device(self.device).type == "cuda")
        for data in input:
            data["pred_palm_template"] = palm_template
            points = data["hand_points"].to(self.device, non_blocking=True).float()
            centre = points.mean(dim=-2, keepdim=True)
            if last_kp is not None:
                data["jittered_hand_kp"] = last_kp + centre  # stays on the device: no host sync between frames
            if graph_ok:
                ret = self._graph_step(points, data["jittered_hand_kp"].to(self.device).float(), palm_template, flag_dict)
            else:
                ret = self.handnet(data, flag_dict)
            last_kp = (ret["pred_kp"] - centre).clone()
            rets.append(ret)
        return rets

    def compute_loss(self, input, ret_dict_lst, flag_dict):
        total = {}
        for data, ret in zip(input, ret_dict_lst):
            loss, _ = self.handnet.compute_loss(data, ret, flag_dict)
            for k, v in loss.items():
                total[k] = total[k] + v if k in total else v  # stays on the device: one host sync per sequence
        return {k: float(v) / len(input) for k, v in total.items()}, ret_dict_lst
