"""Object-pose particle optimiser -- counterpart of the reference's gf_optimize_obj
(network/models/optimization_obj.py:81-301) for the part that is data-parallel hot path: the SDF-volume
evaluation of 2048 candidate poses per iteration.

Same attribute / method names as the reference (`Distance`, `evaluate`, `update_seach_size`, `optimize`,
`sdf_volume`, `volume_size`, `voxel_scale`, `pre_sampled_particle`, ...), so `track_network`-style callers work
unchanged.  Out of scope here (and raising if requested): decoding the volume from a DeepSDF latent code
(`load_obj`, :106-161 -- needs the checkpoints) and the online shape update (`update_shape`, :345-403); the
volume is handed over with `load_volume`.
"""
from __future__ import annotations

import numpy as np
import torch

from hotrack_amd import sdf as _sdf


class gf_optimize_obj:
    def __init__(self, cfg=None, device="cuda", seed=None):
        self.particle_size = 2048  # :85-91
        self.iteration = 10
        self.scaling_coefficient1 = 0.02
        self.scaling_coefficient2 = 2
        self.volume_size = 201
        self.voxel_scale = 0.002
        self.beta = 0.9
        cfg = cfg or {}
        self.update_shape_flag = bool(cfg.get("opt", {}).get("updateobjshape", False))
        if self.update_shape_flag:
            raise NotImplementedError("online DeepSDF shape update (optimization_obj.py:345-403) is out of scope")
        self.device = torch.device(cfg.get("device", device))
        # pre-sampled particles (:103-107): N(0, I_6), particle 0 is the current pose
        rng = np.random.default_rng(seed) if seed is not None else np.random
        pre = rng.multivariate_normal(np.zeros(6), np.eye(6), self.particle_size)
        pre[0, :] = 0
        self.pre_sampled_particle = torch.tensor(pre, dtype=torch.float32, device=self.device)
        self.sdf_volume = None
        self._work = None

    def load_volume(self, sdf_volume: torch.Tensor, voxel_scale: float | None = None):
        """Install a (V,V,V) fp16/fp32 SDF volume (what load_obj produces at :139-149)."""
        V = sdf_volume.shape[0]
        assert sdf_volume.dim() == 3 and tuple(sdf_volume.shape) == (V, V, V)
        self.volume_size = V
        if voxel_scale is not None:
            self.voxel_scale = float(voxel_scale)
        self.sdf_volume = sdf_volume.to(self.device).contiguous()
        self._corners = _sdf.CornerVolume(self.sdf_volume)  # lookup layout (8x memory, same results), built once per object

    def Distance(self, V):  # noqa: N802 (reference name)
        return _sdf.distance(V.float(), self._corners, self.voxel_scale)

    def evaluate(self, pcld, r, t):
        """pcld (1,N,3), r (P,3,3), t (P,3,1) -> (energy, sdf_energy), each (P,)."""
        sdf_energy = _sdf.particle_energy(pcld.float(), r.float(), t.float(), self._corners, self.voxel_scale)
        return sdf_energy * 500, sdf_energy

    def update_seach_size(self, tsdf, mean_transform):  # reference spelling
        s = mean_transform.abs() + 1e-3
        return tsdf * self.scaling_coefficient2 * s / s.norm() + 1e-3

    def optimize(self, pcld, init_obj_pose, category=None, file_name=None, projection=None):
        """Ten particle iterations around init_obj_pose; returns {'rotation' (1,3,3), 'translation' (1,3,1)}.
        One call enqueues 11 kernels and never synchronises with the host."""
        rotation = init_obj_pose["rotation"].float().to(self.device)
        translation = init_obj_pose["translation"].float().to(self.device)
        pcld = pcld.float().to(self.device)
        need = 16 + self.pre_sampled_particle.shape[0]
        if self._work is None or self._work.numel() < need:
            self._work = torch.empty(need, dtype=torch.float32, device=self.device)
        R, t = _sdf.obj_optimize(pcld, rotation, translation, self.pre_sampled_particle, self._corners, self.voxel_scale,
                                 iterations=self.iteration, scaling_coefficient1=self.scaling_coefficient1,
                                 scaling_coefficient2=float(self.scaling_coefficient2), beta=self.beta, work=self._work)
        return {"rotation": R, "translation": t.reshape(1, 3, 1)}
