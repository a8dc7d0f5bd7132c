"""Python binding of include/pn2_sdf.h -- the particle optimisers' SDF-volume lookups.  GPU tensors only.

Mirrors, function for function, the reference methods it replaces:
  distance          gf_optimize_obj.Distance                     network/models/optimization_obj.py:184-228
  particle_energy   gf_optimize_obj.evaluate                     :230-237
  obj_optimize      the particle loop of gf_optimize_obj.optimize :253-301
  query_sdf         gf_optimize_hand_pose.query_sdf              network/models/optimization_hand.py:252-262
  penetration_loss  ... .get_penetration_loss(query_sdf(hand))   :264-268 (fused with the lookup)
"""
from __future__ import annotations

import ctypes

import torch

from . import pointnet2_hip as _native

_lib = _native._lib
_vp, _ci, _cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_lib.pn2s_trilinear.argtypes = [_ci, _vp, _vp, _ci, _ci, _cf, _cf, _cf, _cf, _vp, _vp]
_lib.pn2s_particle_energy.argtypes = [_ci, _ci, _vp, _vp, _vp, _vp, _ci, _ci, _cf, _cf, _cf, _cf, _vp, _vp]
_lib.pn2s_obj_optimize.argtypes = [_ci, _ci, _ci, _vp, _vp, _vp, _ci, _ci, _cf, _cf, _cf, _cf, _cf, _cf, _cf, _vp, _vp, _vp]
_lib.pn2s_obj_optimize_work_floats.argtypes = [_ci]
_lib.pn2s_nearest.argtypes = [_ci, _ci, _vp, _vp, _vp, _vp, _ci, _ci, _cf, _vp, _vp, _vp, _vp]
for _n in ("pn2s_trilinear", "pn2s_particle_energy", "pn2s_obj_optimize", "pn2s_obj_optimize_work_floats", "pn2s_nearest"):
    getattr(_lib, _n).restype = _ci

_lib.pn2s_build_corner_volume.argtypes = [_vp, _ci, _ci, _vp, _vp]
_lib.pn2s_build_corner_volume.restype = _ci
_lib.pn2s_corner_volume_elems.argtypes = [_ci]
_lib.pn2s_corner_volume_elems.restype = ctypes.c_long

BBOX_MIN = -0.2  # optimization_obj.py:186
CLAMP = (-0.05, 0.05)  # optimization_obj.py:227
_f32 = torch.float32


class CornerVolume:
    """Corner-layout copy of an SDF volume for the trilinear entries (include/pn2_sdf.h: pn2s_build_corner_volume):
    cell i holds the eight corner values Distance() fetches for base index i, so a lookup is one 16-byte load.
    Bit-identical results, ~3x faster lookups, 8x the memory; build once per object and pass it wherever a
    `sdf_volume` is accepted by distance / particle_energy / obj_optimize."""

    def __init__(self, sdf_volume: torch.Tensor):
        pv, f16, res = _linear_volume(sdf_volume)
        self.res, self.f16, self.source = res, f16, sdf_volume
        self.data = torch.empty((res ** 3, 8), dtype=sdf_volume.dtype, device=sdf_volume.device)
        assert _lib.pn2s_corner_volume_elems(res) == self.data.numel()
        with torch.cuda.device(sdf_volume.device):
            rc = _lib.pn2s_build_corner_volume(pv, f16, res, self.data.data_ptr(), _native._stream(sdf_volume))
        _native._check(rc, "sdf.CornerVolume")


def _volume(vol):
    """-> (device pointer, vol_fmt, res) for a linear volume tensor or a CornerVolume."""
    if isinstance(vol, CornerVolume):
        return vol.data.data_ptr(), 2 + vol.f16, vol.res
    return _linear_volume(vol)


def _linear_volume(vol: torch.Tensor):
    """Validate an SDF volume: (res,res,res) or flat res^3, fp16 or fp32, contiguous, on the GPU."""
    if not isinstance(vol, torch.Tensor) or not vol.is_cuda:
        raise RuntimeError("sdf volume must be a GPU (HIP) tensor -- hotrack_amd has no CPU path")
    if vol.dtype not in (torch.float16, torch.float32):
        raise TypeError(f"sdf volume must be float16 or float32, got {vol.dtype}")
    if not vol.is_contiguous():
        raise ValueError("sdf volume must be contiguous")
    res = vol.shape[0] if vol.dim() == 3 else round(vol.numel() ** (1.0 / 3.0))
    if res ** 3 != vol.numel() or (vol.dim() == 3 and tuple(vol.shape) != (res, res, res)):
        raise ValueError(f"sdf volume must hold res^3 elements, got shape {tuple(vol.shape)}")
    return vol.data_ptr(), int(vol.dtype == torch.float16), res


def distance(V: torch.Tensor, sdf_volume: torch.Tensor, voxel_scale: float, bbox_min: float = BBOX_MIN, clamp=CLAMP) -> torch.Tensor:
    """Trilinear SDF at V (M,3) object-frame points -> (M,) fp32, clamped (== gf_optimize_obj.Distance)."""
    pv, f16, res = _volume(sdf_volume)
    if V.dim() != 2 or V.shape[1] != 3:
        raise ValueError(f"V must be (M,3), got {tuple(V.shape)}")
    V = V.contiguous()
    m = V.shape[0]
    pV = _native._ptr(V, "V", _f32, m * 3)
    out = torch.empty((m,), dtype=_f32, device=V.device)
    with torch.cuda.device(V.device):
        rc = _lib.pn2s_trilinear(m, pV, pv, f16, res, bbox_min, voxel_scale, clamp[0], clamp[1],
                                 out.data_ptr(), _native._stream(V))
    _native._check(rc, "sdf.distance")
    return out


def particle_energy(pcld: torch.Tensor, r: torch.Tensor, t: torch.Tensor, sdf_volume: torch.Tensor, voxel_scale: float,
                    bbox_min: float = BBOX_MIN, clamp=CLAMP) -> torch.Tensor:
    """sdf_energy (P,) = mean_n |Distance((pcld - t_p) @ r_p)|  (== the second output of gf_optimize_obj.evaluate;
    the first is 500x this).  pcld (N,3) or (1,N,3); r (P,3,3); t (P,3) or (P,3,1)."""
    pv, f16, res = _volume(sdf_volume)
    pcld = pcld.reshape(-1, 3).contiguous()
    P = r.shape[0]
    r = r.contiguous()
    t = t.reshape(P, 3).contiguous()
    n = pcld.shape[0]
    ptrs = (_native._ptr(pcld, "pcld", _f32, n * 3), _native._ptr(r, "r", _f32, P * 9), _native._ptr(t, "t", _f32, P * 3))
    out = torch.empty((P,), dtype=_f32, device=pcld.device)
    with torch.cuda.device(pcld.device):
        rc = _lib.pn2s_particle_energy(P, n, *ptrs, pv, f16, res, bbox_min, voxel_scale, clamp[0], clamp[1],
                                       out.data_ptr(), _native._stream(pcld))
    _native._check(rc, "sdf.particle_energy")
    return out


def obj_optimize(pcld: torch.Tensor, rotation: torch.Tensor, translation: torch.Tensor, pre_sampled_particle: torch.Tensor,
                 sdf_volume: torch.Tensor, voxel_scale: float, iterations: int = 10, scaling_coefficient1: float = 0.02,
                 scaling_coefficient2: float = 2.0, beta: float = 0.9, bbox_min: float = BBOX_MIN, clamp=CLAMP, work=None):
    """The particle loop of gf_optimize_obj.optimize on the device, no host synchronisation.
    pcld (N,3)|(1,N,3); rotation (3,3)|(1,3,3); translation (3,)|(1,3,1); pre_sampled_particle (P,6), row 0 == 0.
    Returns (rotation (1,3,3), translation (1,3,1)) as new tensors."""
    pv, f16, res = _volume(sdf_volume)
    pcld = pcld.reshape(-1, 3).contiguous()
    n = pcld.shape[0]
    pre = pre_sampled_particle.contiguous()
    P = pre.shape[0]
    if pre.dim() != 2 or pre.shape[1] != 6:
        raise ValueError("pre_sampled_particle must be (P,6)")
    ptrs = (_native._ptr(pcld, "pcld", _f32, n * 3), _native._ptr(pre, "pre_sampled_particle", _f32, P * 6))
    pose = torch.cat([rotation.reshape(9).to(_f32), translation.reshape(3).to(_f32)]).contiguous()
    _native._ptr(pose, "rotation/translation", _f32, 12)
    need = _lib.pn2s_obj_optimize_work_floats(P)
    if work is None:
        work = torch.empty((need,), dtype=_f32, device=pcld.device)
    elif work.numel() < need or work.dtype != _f32 or not work.is_cuda or not work.is_contiguous():
        raise ValueError(f"work must be a contiguous float32 GPU tensor of >= {need} elements")
    with torch.cuda.device(pcld.device):
        rc = _lib.pn2s_obj_optimize(P, n, iterations, *ptrs, pv, f16, res, bbox_min, voxel_scale, clamp[0], clamp[1], scaling_coefficient1, scaling_coefficient2,
                                    beta, pose.data_ptr(), work.data_ptr(), _native._stream(pcld))
    _native._check(rc, "sdf.obj_optimize")
    return pose[:9].view(1, 3, 3), pose[9:].view(1, 3, 1)


def query_sdf(hand: torch.Tensor, obj_r: torch.Tensor, obj_t: torch.Tensor, sdf_volume: torch.Tensor, voxel_scale: float,
              with_penetration: bool = False, with_index: bool = False):
    """Nearest-voxel SDF of hand (B,N,3) in the object frame (== gf_optimize_hand_pose.query_sdf), dtype of the
    volume.  with_penetration: also the fused get_penetration_loss (B,).  with_index: also the flat voxel index."""
    pv, f16, res = _linear_volume(sdf_volume)
    if hand.dim() != 3 or hand.shape[2] != 3:
        raise ValueError(f"hand must be (B,N,3), got {tuple(hand.shape)}")
    hand = hand.contiguous()
    B, N, _ = hand.shape
    obj_r = obj_r.reshape(3, 3).to(_f32).contiguous()
    obj_t = obj_t.reshape(3).to(_f32).contiguous()
    ptrs = (_native._ptr(hand, "hand", _f32, B * N * 3), _native._ptr(obj_r, "obj_r", _f32, 9), _native._ptr(obj_t, "obj_t", _f32, 3))
    sdf = torch.empty((B, N), dtype=sdf_volume.dtype, device=hand.device)
    pen = torch.empty((B,), dtype=sdf_volume.dtype, device=hand.device) if with_penetration else None
    idx = torch.empty((B, N), dtype=torch.int32, device=hand.device) if with_index else None
    with torch.cuda.device(hand.device):
        rc = _lib.pn2s_nearest(B, N, *ptrs, pv, f16, res, voxel_scale,
                               None if idx is None else idx.data_ptr(), sdf.data_ptr(), None if pen is None else pen.data_ptr(),
                               _native._stream(hand))
    _native._check(rc, "sdf.query_sdf")
    ret = (sdf,)
    if with_penetration:
        ret += (pen,)
    if with_index:
        ret += (idx,)
    return ret[0] if len(ret) == 1 else ret
