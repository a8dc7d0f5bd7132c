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
    """hand_model (models/hand_model.HandModel, e.g. a MANO layer with the reference's call signature): enables the
    hand-pose particle optimisation of the reference's `use_optimization` branch (track_network.py:142-156, :203-211) on
    top of the HandTrackNet tracking loop.  In the reference that branch sits behind IKNet, which supplies the initial
    MANO pose code and global pose; IKNet needs the MANO assets and its checkpoint, so here those two inputs come from a
    stand-in with the same role (`_pose_init`): the pose code of the previous frame's optimum and the rigid fit of the hand
    model's keypoints to HandTrackNet's prediction (device Kabsch).  Everything downstream -- visibility mask, candidate
    evaluation, update rule, what is fed to the next frame -- is the reference's."""

    def __init__(self, cfg, handnet, IKnet=None, hand_model=None):
        super().__init__()
        if IKnet is not None:
            raise NotImplementedError("IKNet needs the MANO assets and its checkpoint (out of scope)")
        self.device = cfg["device"]
        self.handnet = handnet(cfg)
        self.use_graph = True  # GPU + fused backend: one captured HIP graph per (N, keypoints) shape, replayed per frame
        self._graphs = {}
        self.use_optimization = bool(cfg.get("use_optimization", False)) and hand_model is not None
        self.use_pred_obj_pose = bool(cfg.get("use_pred_obj_pose", False))
        self.optimizer = None
        if self.use_optimization:
            from .optimization_hand import gf_optimize_hand_pose
            self.optimizer = gf_optimize_hand_pose(cfg, hand_model=hand_model, particle_size=int(cfg.get("hand_particles", 5120)))

    # A captured graph bakes in the pointers of the BN-folded weights FastEval built at capture time: anything that can
    # change the weights (checkpoint load, fine-tuning, .to()/.float()) drops the captured graphs.
    def train(self, mode: bool = True):
        if mode or mode != self.training:  # eval() on a model already in eval mode (Trainer.test per sequence) keeps the graphs
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

    def _pose_init(self, pred_kp, prev_theta):
        """Stand-in for IKNet's two outputs (see the class docstring): MANO_theta (1,45) and global_pose."""
        from hotrack_amd import ext
        hm = self.optimizer.mano_layer_right
        theta = prev_theta if prev_theta is not None else torch.zeros((1, hm.num_pose), device=pred_kp.device)
        with torch.no_grad():
            _, kp0 = hm.forward(th_pose_coeffs=torch.cat([torch.zeros((1, 3), device=pred_kp.device), theta], dim=1),
                                th_trans=torch.zeros((1, 3), device=pred_kp.device))
            R, t = ext.kabsch(kp0.contiguous(), pred_kp.contiguous())  # pred ~ R kp0 + t
        return theta, {"rotation": R.reshape(1, 3, 3), "translation": t.reshape(1, 3, 1)}

    def forward(self, input, flag_dict):
        flag_dict["track_flag"] = True
        assert flag_dict["test_flag"]
        flag_dict["opt_flag"] = self.use_optimization
        palm_template = input[0]["gt_hand_pose"]["palm_template"].to(self.device).float()
        last_kp = None
        rets = []
        graph_ok = (self.use_graph and pointnet_utils.fused_backend() is not None and not self.training
                    and not torch.is_grad_enabled() and torch.device(self.device).type == "cuda")
        if self.use_optimization:
            flag_dict["IKNet_flag"] = True  # HandTrackNet also returns the keypoint visibility mask (hand_network.py:149-155)
            if "sdf_volume" in input[0]:
                self.optimizer.load_volume(input[0]["sdf_volume"], input[0].get("voxel_scale"))
            elif self.optimizer.sdf_volume is None:
                raise RuntimeError("use_optimization: no SDF volume (decoding it from a DeepSDF latent needs the checkpoints); "
                                   "put 'sdf_volume' / 'voxel_scale' into the sequence's first frame")
        prev_theta = None
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
            if self.use_optimization:  # track_network.py:142-156 (IKNet's role: _pose_init), :203-211
                ret["baseline_pred_kp"] = ret["pred_kp"].clone()
                theta0, pose0 = self._pose_init(ret["baseline_pred_kp"], prev_theta)
                obj_pose = data["pred_obj_pose"] if (self.use_pred_obj_pose and "pred_obj_pose" in data) else data["gt_obj_pose"]
                kp, theta, rot, trans = self.optimizer.optimize(theta0, pose0, ret["baseline_pred_kp"], last_kp, ret["pred_kp_vis_mask"],
                                                                obj_pose, data.get("pred_beta"), data["projection"], data["background_mask"])
                ret["pred_kp"], ret["MANO_theta"] = kp, theta
                ret["global_pose"] = {"rotation": rot.unsqueeze(0), "translation": trans.unsqueeze(-1)}
                prev_theta = theta
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


class ObjTrackModel_Optimization(nn.Module):
    """Per-sequence object-pose tracking by gradient-free particle optimisation against the object's SDF volume
    (counterpart of the reference's ObjTrackModel_Optimization, track_network.py:322-383; BASELINE configs[3], stage 1).

    Frame 0 starts from the jittered pose; frame t starts from frame t-1's result, which also becomes the `prev_*` entries
    of the pose dict (:353-370).  Each frame is one `gf_optimize_obj.optimize` call = 10 iterations x 2048 candidate
    poses evaluated by the fused SDF-lookup kernels with the pose update on the device (hotrack_amd/csrc/sdf.hip): the loop
    never synchronises with the host.  The reference decodes the volume from a DeepSDF latent per sequence
    (`load_obj_for_opt` + `optimizer.load_obj`, :342-346 -- needs the checkpoints); here the sequence hands the volume
    over (`input[0]['sdf_volume']`, `['voxel_scale']`)."""

    def __init__(self, cfg):
        super().__init__()
        from .optimization_obj import gf_optimize_obj
        self.device = cfg["device"]
        self.dataset_name = cfg["data_cfg"]["dataset_name"]
        self.sdf_code_source = cfg.get("sdf_code_source", "pred")
        self.num_parts = cfg.get("num_parts", 1)
        self.sym = cfg.get("obj_sym", -1)
        self.optimizer = gf_optimize_obj(cfg)

    def forward(self, input, flag_dict):
        flag_dict["track_flag"] = True
        assert flag_dict["test_flag"]
        if "sdf_volume" in input[0]:
            self.optimizer.load_volume(input[0]["sdf_volume"], input[0].get("voxel_scale"))
        elif self.optimizer.sdf_volume is None:
            raise RuntimeError("no SDF volume: decoding it from a DeepSDF latent needs the checkpoints (out of scope); "
                               "put 'sdf_volume' / 'voxel_scale' into the sequence's first frame")
        last = None
        rets = []
        for data in input:
            if last is not None:
                data["jittered_obj_pose"] = last
            else:
                jp = data["jittered_obj_pose"]
                jp["translation"] = jp["translation"].float().reshape(1, 3, 1).to(self.device)
                jp["rotation"] = jp["rotation"].float().reshape(1, 3, 3).to(self.device)
                jp["prev_translation"], jp["prev_rotation"] = jp["translation"], jp["rotation"]
                last = {"translation": jp["translation"], "rotation": jp["rotation"]}
            ret = self.optimizer.optimize(data["obj_points"], data["jittered_obj_pose"], data["category"][0],
                                          data["file_name"][0], data.get("projection"))
            last["prev_translation"], last["prev_rotation"] = last["translation"], last["rotation"]  # last frame's pose
            last["translation"], last["rotation"] = ret["translation"], ret["rotation"]            # current frame's pose
            rets.append(ret)
        return rets

    def compute_loss(self, input, ret_dict_lst, flag_dict):
        """Mean rotation / symmetry-axis (degrees) and translation (metres) error against gt_obj_pose.  (The reference
        evaluates through pose_utils.part_dof_utils.eval_part_full plus a chamfer term on the reconstructed mesh, :385-440 --
        mesh assets.)"""
        r_err = t_err = a_err = 0.0
        for data, ret in zip(input, ret_dict_lst):
            gR = data["gt_obj_pose"]["rotation"].float().reshape(3, 3).to(self.device)
            gt = data["gt_obj_pose"]["translation"].float().reshape(3).to(self.device)
            R, t = ret["rotation"].reshape(3, 3), ret["translation"].reshape(3)
            cos = ((R.t() @ gR).diagonal().sum() - 1) / 2
            r_err = r_err + torch.rad2deg(torch.arccos(cos.clamp(-1, 1)))
            a_err = a_err + torch.rad2deg(torch.arccos((R[:, 2] * gR[:, 2]).sum().clamp(-1, 1)))  # object z axis (revolution axis)
            t_err = t_err + (t - gt).norm()
        n = max(len(input), 1)
        return {"obj_pred_r_diff": float(r_err) / n, "obj_pred_axis_diff": float(a_err) / n, "obj_pred_t_diff": float(t_err) / n}, ret_dict_lst
