"""The hand model behind the hand-pose particle optimiser, as an interface.

The reference evaluates 5120 candidate hands per iteration through a MANO layer (third_party/mano/our_mano.py, called at
optimization_hand.py:216-229, :386-387) -- licensed assets this repository cannot ship.  The optimiser only needs
`vertices, keypoints = f(pose, translation)`, a PCA basis for the pose coefficients and the fingertip contact zones, so that
is the interface (`HandModel`); `SyntheticLBSHand` is a deterministic linear-blend-skinning hand with MANO's sizes (778
vertices, 21 keypoints, 45 pose dimensions, MANO keypoint order) for tests, golden vectors and the synthetic sequences.  A
MANO layer with the reference's call signature plugs in unchanged."""
from __future__ import annotations

import math

import torch
import torch.nn as nn


def rodrigues(aa: torch.Tensor) -> torch.Tensor:
    """(…,3) axis-angle -> (…,3,3)."""
    theta = aa.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    k = aa / theta
    kx, ky, kz = k.unbind(-1)
    z = torch.zeros_like(kx)
    K = torch.stack((z, -kz, ky, kz, z, -kx, -ky, kx, z), dim=-1).view(aa.shape[:-1] + (3, 3))
    s, c = torch.sin(theta)[..., None], torch.cos(theta)[..., None]
    eye = torch.eye(3, dtype=aa.dtype, device=aa.device).expand(aa.shape[:-1] + (3, 3))
    return eye + s * K + (1 - c) * (K @ K)


class HandModel(nn.Module):
    """vertices (P, V, 3), keypoints (P, 21, 3) = forward(th_pose_coeffs (P, 3 + num_pose), th_trans (P, 3)).
    Call signature of the reference's OurManoLayer.forward (keyword arguments th_pose_coeffs / th_trans / th_betas /
    use_registed_beta), `pca_comps2pose(ncomps, coeffs)` (our_mano.py:208-209), `register_beta` (:211-216), and
    `contact_zones` = {1..5: vertex indices of the index / middle / ring / pinky / thumb tip regions} (the obman contact zones
    the reference loads at optimization_hand.py:160-168)."""

    num_verts: int
    num_pose: int
    contact_zones: dict

    def pca_comps2pose(self, ncomps: int, pca: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def register_beta(self, th_betas=None):
        return None


class SyntheticLBSHand(HandModel):
    FINGERS = ((1, 2, 3, 4), (5, 6, 7, 8), (9, 10, 11, 12), (13, 14, 15, 16), (17, 18, 19, 20))  # thumb, index, middle, ring, pinky

    def __init__(self, num_verts: int = 778, seed: int = 0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.num_verts, self.num_pose = num_verts, 45
        # ---- rest skeleton (metres), wrist at the origin, fingers along +y, palm in the xy plane -------------------------
        rest = torch.zeros(21, 3)
        base_x = (-0.035, -0.02, 0.0, 0.018, 0.034)
        base_y = (0.02, 0.085, 0.09, 0.085, 0.075)
        seg = ((0.038, 0.03, 0.026), (0.04, 0.025, 0.02), (0.044, 0.028, 0.022), (0.04, 0.026, 0.02), (0.032, 0.02, 0.018))
        for f, chain in enumerate(self.FINGERS):
            p = torch.tensor([base_x[f], base_y[f], 0.0])
            direction = torch.tensor([-0.6, 0.8, 0.0]) if f == 0 else torch.tensor([0.05 * (f - 2), 1.0, 0.0])
            direction = direction / direction.norm()
            rest[chain[0]] = p
            for j in range(3):
                p = p + direction * seg[f][j]
                rest[chain[j + 1]] = p
        parents = [0] * 21
        for chain in self.FINGERS:
            parents[chain[0]] = 0
            for a, b in zip(chain[:-1], chain[1:]):
                parents[b] = a
        self.parents = parents
        # articulated joints (15): the first three of every finger chain; pose block j drives joint ART[j]
        self.art = [c for chain in self.FINGERS for c in chain[:3]]
        # ---- vertices: cylinders around the 20 bones + a palm slab, skinned to the bone's two end joints -------------------
        bones = [(parents[j], j) for j in range(1, 21)]
        per = num_verts // 22
        verts, w_idx, w_val, bone_of = [], [], [], []
        for b, (pa, ch) in enumerate(bones):
            n = per
            t = torch.rand(n, generator=g)
            ang = torch.rand(n, generator=g) * 2 * math.pi
            axis = rest[ch] - rest[pa]
            ax = axis / axis.norm()
            u = torch.linalg.cross(ax, torch.tensor([0.0, 0.0, 1.0]))
            u = u / u.norm()
            v = torch.linalg.cross(ax, u)
            radius = 0.009 if pa != 0 else 0.012
            pts = rest[pa] + t[:, None] * axis + radius * (torch.cos(ang)[:, None] * u + torch.sin(ang)[:, None] * v)
            verts.append(pts)
            w_idx.append(torch.tensor([[pa, ch]]).expand(n, 2))
            w_val.append(torch.stack((1 - t, t), dim=1))
            bone_of += [b] * n
        n_palm = num_verts - per * 20
        palm = torch.stack((torch.rand(n_palm, generator=g) * 0.08 - 0.04, torch.rand(n_palm, generator=g) * 0.08,
                            (torch.rand(n_palm, generator=g) - 0.5) * 0.02), dim=1)
        verts.append(palm)
        w_idx.append(torch.zeros(n_palm, 2, dtype=torch.long))
        w_val.append(torch.tensor([[1.0, 0.0]]).expand(n_palm, 2))
        bone_of += [-1] * n_palm
        self.register_buffer("rest_joints", rest)
        self.register_buffer("rest_verts", torch.cat(verts))
        self.register_buffer("skin_idx", torch.cat(w_idx).long())
        self.register_buffer("skin_w", torch.cat(w_val).float())
        # PCA basis of the pose space: orthonormal rows, scaled like MANO's components (a few hundredths of a radian per unit)
        q, _ = torch.linalg.qr(torch.randn(45, 45, generator=g))
        self.register_buffer("th_comps", (q * 0.02).contiguous())
        # fingertip contact zones, numbered as the reference uses them (optimization_hand.py:164-168 with the finger order of
        # get_attraction_loss, :240: keypoints 8, 12, 16, 20, 4 = index, middle, ring, pinky, thumb)
        bone_of = torch.tensor(bone_of)
        tip_bone = {kp: bones.index((parents[kp], kp)) for kp in (8, 12, 16, 20, 4)}
        self.contact_zones = {i + 1: torch.nonzero(bone_of == tip_bone[kp]).flatten().tolist() for i, kp in enumerate((8, 12, 16, 20, 4))}

    def pca_comps2pose(self, ncomps: int, pca: torch.Tensor) -> torch.Tensor:
        return pca.mm(self.th_comps[:ncomps])

    def forward(self, th_pose_coeffs, th_betas=None, th_trans=None, use_registed_beta=False, **_):
        P = th_pose_coeffs.shape[0]
        dev, dt = th_pose_coeffs.device, th_pose_coeffs.dtype
        Rg = rodrigues(th_pose_coeffs[:, :3])                                  # (P,3,3)
        Rl = rodrigues(th_pose_coeffs[:, 3:].reshape(P, 15, 3))                # (P,15,3,3)
        rest = self.rest_joints.to(dt)
        R_w = [None] * 21
        t_w = [None] * 21
        R_w[0] = Rg
        t_w[0] = torch.zeros(P, 3, dtype=dt, device=dev)
        art_of = {j: a for a, j in enumerate(self.art)}
        for chain in self.FINGERS:
            for j in chain:
                pa = self.parents[j]
                off = (rest[j] - rest[pa]).view(1, 3, 1)
                t_w[j] = t_w[pa] + (R_w[pa] @ off).squeeze(-1)
                R_w[j] = R_w[pa] @ Rl[:, art_of[j]] if j in art_of else R_w[pa]
        R_w = torch.stack(R_w, dim=1)                                          # (P,21,3,3)
        t_w = torch.stack(t_w, dim=1)                                          # (P,21,3)
        # linear blend skinning: v = sum_k w_k (R_k (v_rest - j_k) + t_k)
        rel = (self.rest_verts.to(dt)[:, None, :] - rest[self.skin_idx])       # (V,2,3)
        Rk = R_w[:, self.skin_idx]                                             # (P,V,2,3,3)
        vk = (Rk @ rel[None, :, :, :, None]).squeeze(-1) + t_w[:, self.skin_idx]
        verts = (vk * self.skin_w.to(dt)[None, :, :, None]).sum(dim=2)         # (P,V,3)
        joints = t_w
        if th_trans is not None:
            verts = verts + th_trans[:, None, :]
            joints = joints + th_trans[:, None, :]
        return verts, joints
