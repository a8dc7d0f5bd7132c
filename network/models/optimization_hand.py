"""Hand-pose particle optimiser (counterpart of the reference's gf_optimize_hand_pose,
network/models/optimization_hand.py:138-394): a gradient-free search over global rotation, translation and PCA pose
coefficients of the hand -- 5120 candidate hands per iteration, 5 iterations per frame -- that trades keypoint fidelity
against hand-object penetration, silhouette, temporal smoothness and fingertip attraction.

What is kept: every energy term (`get_silhouette_loss`, `get_penetration_loss`, `get_regularization_loss`,
`get_temporal_smooth_loss`, `get_attraction_loss`, :230-275), `evaluate` (:277-293), `update_seach_size` (:295-298), the
`optimize` loop with its weighted-mean update, re-projection onto SO(3) and search-size schedule (:335-394) -- same
arithmetic, same epsilons (tests/test_hand_opt.py compares a multi-frame run with the IMPORTED reference class driving the
same hand model, tests/golden/hand_opt_sequence.npz).

What is different:
  * the hand model is an interface (models/hand_model.py): the reference hard-wires a MANO layer (licensed assets); any
    module with OurManoLayer's call signature plugs in, `SyntheticLBSHand` is the stand-in for tests and synthetic data;
  * the SDF lookup + penetration maximum of all candidates is ONE kernel launch (hotrack_amd.sdf.query_sdf with
    with_penetration, csrc/sdf.hip) instead of ~20 torch kernels over (P x 778) temporaries;
  * no host synchronisation inside the loop: the reference branches on `torch.any(better_mask)` and on
    `penetrate_sum_loss[0] != 0` every iteration (:359, :284); here both are torch.where selections on the device, so a frame
    is a fixed launch sequence (graph-capturable);
  * the silhouette mask is handed over by the caller (the reference reads a PNG per frame from the dataset folder, :316-331);
  * the object volume is handed over (`load_volume`) instead of being decoded from a DeepSDF latent (`load_obj`, :186-213:
    needs the checkpoints).
"""
from __future__ import annotations

import torch

from .rotations import matrix_to_unit_quaternion, quaternion_to_axis_angle, rotation_from_ortho6d, unit_quaternion_to_matrix


def world2point2D(xyz, fx, fy, cx, cy):
    """(B,N,3) -> (B,N,2) pixel (row, column), optimization_hand.py:13-21."""
    x = xyz[..., 0] / xyz[..., 2] * fx + cx
    y = xyz[..., 1] / xyz[..., 2] * fy + cy
    return torch.stack([y, x], dim=-1).float()


class gf_optimize_hand_pose:
    def __init__(self, cfg=None, hand_model=None, device="cuda", particle_size=5120, seed=None):
        cfg = cfg or {}
        self.ncomps = 10                                   # :146
        self.optimize_dim = 6 + self.ncomps
        self.particle_size = particle_size                 # :151
        self.iteration = 5
        self.energy_weight = dict((cfg.get("opt") or {}).get("energy_weight") or
                                  {"penetrate_sum_loss": 1, "sil_loss": 0.1, "attraction_loss": 0.05, "vis_regu_loss": 10,
                                   "invis_regu_loss": 0, "temporal_smooth": 1})
        self.device = torch.device(cfg.get("device", device))
        self.theta_scale = 30
        self.beta = 0.9
        self.scaling_coefficient2 = 0.1
        self.volume_size = 151
        self.voxel_scale = 0.003
        self.initial_scale = torch.ones(self.optimize_dim, device=self.device) * 0.005
        self.sdf_volume = self.obj_r = self.obj_t = None
        self.last_frame_kp = None
        self.mano_layer_right = None
        if hand_model is not None:
            self.set_hand_model(hand_model)
        # pre-sampled particles: N(0, I) with the first one at the origin (= the current estimate), :158-162
        g = torch.Generator().manual_seed(0 if seed is None else seed)
        pre = torch.randn(self.particle_size, self.optimize_dim, generator=g)
        pre[0] = 0
        self.pre_sampled_particle = pre.to(self.device)

    # ---- pluggable pieces ------------------------------------------------------------------------------------------------
    def set_hand_model(self, hand_model):
        self.mano_layer_right = hand_model.to(self.device)
        zones = hand_model.contact_zones
        self.tips_region, self.finger_mask = [], []        # :164-168
        for i in range(5):
            prev = len(self.tips_region)
            self.tips_region.extend(zones[i + 1])
            self.finger_mask.append(list(range(prev, len(self.tips_region))))
        self._tips = torch.tensor(self.tips_region, dtype=torch.long, device=self.device)
        self._finger_idx = [torch.tensor(m, dtype=torch.long, device=self.device) for m in self.finger_mask]

    def load_volume(self, sdf_volume: torch.Tensor, voxel_scale: float | None = None):
        V = sdf_volume.shape[0]
        assert sdf_volume.dim() == 3 and tuple(sdf_volume.shape) == (V, V, V) and V % 2 == 1
        self.volume_size = V
        if voxel_scale is not None:
            self.voxel_scale = float(voxel_scale)
        self.sdf_volume = sdf_volume.to(self.device).contiguous()

    def set_obj_pose(self, init_obj_pose):  # the two lines of set_init_para that concern the object (:312-313)
        self.obj_r = init_obj_pose["rotation"].to(self.device).reshape(3, 3).float()
        self.obj_t = init_obj_pose["translation"].to(self.device).reshape(1, 1, 3).float()

    # ---- SDF part (one launch for lookup + penetration) --------------------------------------------------------------------
    sdf_lookup = None  # test hook: callable(optimiser, hand) -> (queried_sdf, penetration); None = hotrack_amd.sdf (GPU only)

    def query_sdf(self, hand):
        from hotrack_amd import sdf as _sdf
        return _sdf.query_sdf(hand.float(), self.obj_r, self.obj_t, self.sdf_volume, self.voxel_scale)

    def get_penetration_loss(self, queried_sdf, threshold=0):
        abs_distance = queried_sdf.abs()
        penetrate_mask = (queried_sdf < -threshold).bool()
        return torch.max(abs_distance * penetrate_mask, dim=-1)[0]

    def query_sdf_and_penetration(self, hand):
        """One launch for both (threshold 0): returns (queried_sdf (B,N), penetrate_max (B,))."""
        from hotrack_amd import sdf as _sdf
        return _sdf.query_sdf(hand.float(), self.obj_r, self.obj_t, self.sdf_volume, self.voxel_scale, with_penetration=True)

    # ---- the other energy terms ---------------------------------------------------------------------------------------------
    def get_kp_from_delta(self, delta):
        """delta (B, 1 + 3 + 3 + ncomps) = [quaternion | translation | pose coefficients] -> candidate hands, :215-229."""
        sampled_r = torch.matmul(self.curr_r, unit_quaternion_to_matrix(delta[:, :4]))
        sampled_t = self.curr_t + delta[:, 4:7, None]
        sampled_theta = self.curr_theta + self.mano_layer_right.pca_comps2pose(self.ncomps, delta[:, 7:]) * self.theta_scale
        sampled_axisangle = quaternion_to_axis_angle(matrix_to_unit_quaternion(sampled_r))
        return self.mano_layer_right.forward(th_pose_coeffs=torch.cat([sampled_axisangle, sampled_theta], dim=-1),
                                             th_trans=sampled_t.squeeze(-1), use_registed_beta=True)

    def get_regularization_loss(self, kp):
        error = (kp - self.pred_kp).norm(dim=-1)
        vis = torch.sum(error * self.vis_mask, dim=-1) / torch.clamp(torch.sum(self.vis_mask, dim=-1), 1)
        invis = torch.sum(error * (~self.vis_mask), dim=-1) / torch.clamp(torch.sum(~self.vis_mask, dim=-1), 1)
        return vis, invis

    def get_silhouette_loss(self, hand):
        pred_2D = world2point2D(hand, self.proj["fx"], self.proj["fy"], self.proj["cx"], self.proj["cy"])
        index1 = torch.clamp(pred_2D[..., 0].long(), 0, self.h - 1)
        index2 = torch.clamp(pred_2D[..., 1].long(), 0, self.w - 1)
        return self.gt_background_mask[index1, index2].sum(dim=-1) / pred_2D.shape[1]

    def get_attraction_loss(self, queried_sdf, threshold=0):
        """Sum over the fingers whose tip keypoint is invisible of the smallest positive tip-region distance, :237-246 --
        with the per-finger `if` turned into a mask (no host read of vis_mask)."""
        invis_finger = ~self.vis_mask[0, [8, 12, 16, 20, 4]]
        tips_sdf = queried_sdf[:, self._tips]
        tips_dis = tips_sdf * (tips_sdf > threshold)
        total = torch.zeros(queried_sdf.shape[0], dtype=queried_sdf.dtype, device=queried_sdf.device)
        for i in range(5):  # accumulated in the volume's dtype, finger by finger, like the reference's sum() over its list
            total = total + torch.min(tips_dis[:, self._finger_idx[i]], dim=-1)[0] * invis_finger[i]
        return total

    def get_temporal_smooth_loss(self, kp):
        if self.last_frame_kp is None:
            return 0
        return torch.norm(kp - self.last_frame_kp, dim=-1).mean(dim=1)

    def evaluate(self, hand, kp):
        """Energy of every candidate (B,), :277-293."""
        # queried_sdf / pen stay in the volume's dtype (fp16 in the reference): the penetration and attraction terms are formed
        # in that precision there, and the energies are compared with the reference's to 1e-5
        # (sdf_lookup: None in the product -- the fused HIP lookup, which raises for CPU tensors; CPU-side tests inject the
        # reference's torch composition, oracle/sdf_torch.py, the way they inject the operator backend)
        queried_sdf, pen = self.sdf_lookup(self, hand) if self.sdf_lookup is not None else self.query_sdf_and_penetration(hand)
        loss = {"sil_loss": self.get_silhouette_loss(hand), "penetrate_sum_loss": pen}
        loss["vis_regu_loss"], loss["invis_regu_loss"] = self.get_regularization_loss(kp)
        loss["temporal_smooth"] = self.get_temporal_smooth_loss(kp)
        attr = self.get_attraction_loss(queried_sdf)
        loss["attraction_loss"] = torch.where(pen[0] != 0, attr, torch.zeros_like(attr))  # :284-287 without the host branch
        energy = 0
        for key in ("sil_loss", "penetrate_sum_loss", "vis_regu_loss", "invis_regu_loss", "temporal_smooth", "attraction_loss"):
            energy = energy + loss[key] * self.energy_weight[key]
        return energy

    def update_seach_size(self, energy, mean_transform):
        s = mean_transform.abs() + 1e-3
        return energy * self.scaling_coefficient2 * s / s.norm() + 1e-3

    # ---- per frame ------------------------------------------------------------------------------------------------------------
    def set_init_para(self, init_mano, init_hand_pose, init_kp, last_frame_kp, vis_mask, init_obj_pose, hand_shape, projection,
                      background_mask):
        """:300-333; `background_mask` (h, w) bool replaces the PNG the reference reads from the dataset folder."""
        if hand_shape is not None:
            self.mano_layer_right.register_beta(torch.as_tensor(hand_shape, dtype=torch.float32, device=self.device).reshape(1, -1))
        self.pred_kp = init_kp
        self.last_frame_kp = last_frame_kp
        self.vis_mask = vis_mask
        self.proj = {k: float(torch.as_tensor(v).reshape(-1)[0]) for k, v in projection.items()}
        self.w, self.h = int(self.proj["w"]), int(self.proj["h"])
        self.curr_t = init_hand_pose["translation"].reshape(1, 3, 1).to(self.device)
        self.curr_r = init_hand_pose["rotation"].to(self.device)
        self.curr_theta = init_mano
        self.set_obj_pose(init_obj_pose)
        self.gt_background_mask = torch.as_tensor(background_mask).to(self.device)

    def optimize(self, init_mano, init_hand_pose, init_kp, last_frame_kp, vis_mask, init_obj_pose, hand_shape=None, projection=None,
                 background_mask=None):
        """One frame: returns (final keypoints (1,21,3), MANO pose (1,45), rotation (3,3), translation (1,3)), :335-394."""
        self.set_init_para(init_mano, init_hand_pose, init_kp, last_frame_kp, vis_mask, init_obj_pose, hand_shape, projection, background_mask)
        dev = self.device
        search_size = self.initial_scale
        prev_search_size = search_size
        prev_success = torch.ones((), dtype=torch.bool, device=dev)
        for _ in range(self.iteration):
            sample_part = self.pre_sampled_particle * search_size
            sample_qw = torch.sqrt(1 - sample_part[:, 0] ** 2 - sample_part[:, 1] ** 2 - sample_part[:, 2] ** 2).unsqueeze(1)
            sample = torch.cat([sample_qw, sample_part], dim=1)
            hand, kp = self.get_kp_from_delta(sample)
            energy = self.evaluate(hand, kp)

            origin_energy = energy[0]
            better_mask = energy < origin_energy
            weight = (origin_energy - energy) * better_mask
            weight_sum = weight.sum()
            success = better_mask.any()
            mean_energy = torch.where(success, (energy * weight).sum() / weight_sum, energy[0])

            mt = (sample * weight.unsqueeze(1)).sum(dim=0, keepdim=True) / weight_sum               # (1, 7 + ncomps); NaN when no success
            q = mt[:, :4] / mt[:, :4].norm()
            mt = torch.cat([q, mt[:, 4:]], dim=1)
            new_r = torch.matmul(self.curr_r, unit_quaternion_to_matrix(mt[:, :4]))
            # re-projection onto SO(3): accumulated products drift (:377-378)
            new_r = rotation_from_ortho6d(new_r.reshape(-1, 9)[:, :6]).transpose(-1, -2)
            new_t = self.curr_t + mt[:, 4:7, None]
            new_theta = self.curr_theta + self.mano_layer_right.pca_comps2pose(self.ncomps, mt[:, 7:]) * self.theta_scale
            self.curr_r = torch.where(success, new_r, self.curr_r)
            self.curr_t = torch.where(success, new_t, self.curr_t)
            self.curr_theta = torch.where(success, new_theta, self.curr_theta)
            mean_transform = torch.where(success, mt, torch.zeros_like(mt))

            search_size = self.update_seach_size(mean_energy, mean_transform[:, 1:])
            both = prev_success & success
            search_size = torch.where(both, self.beta * search_size + (1 - self.beta) * prev_search_size, search_size)
            prev_search_size = torch.where(success, search_size, prev_search_size)
            prev_success = success

        curr_axisangle = quaternion_to_axis_angle(matrix_to_unit_quaternion(self.curr_r))
        _, final_kp = self.mano_layer_right.forward(th_pose_coeffs=torch.cat([curr_axisangle, self.curr_theta], dim=-1),
                                                    th_trans=self.curr_t.squeeze(-1), use_registed_beta=True)
        return final_kp, self.curr_theta, self.curr_r.squeeze(0), self.curr_t.squeeze(-1)
