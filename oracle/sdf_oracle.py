"""numpy front-end for oracle/sdf_oracle.c + a numpy-fp32 restatement of gf_optimize_obj.optimize's loop.

TEST INFRASTRUCTURE ONLY -- may be imported by tests/, __graft_entry__.smoke() and bench-side
cpu_baseline legs; never by anything under hotrack_amd/ or network/.
Pinned against the imported reference (tests/golden/make_golden_sdf.py -> tests/golden/sdf_*.npz).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import pn2_oracle as _base

_vp = ctypes.c_void_p
_ci, _cf = ctypes.c_int, ctypes.c_float
_ready = False


def _lib():
    global _ready
    lib = _base.lib()
    if not _ready:
        lib.pn2o_sdf_trilinear.argtypes = [_ci, _vp, _vp, _ci, _ci, _cf, _cf, _cf, _cf, _vp]
        lib.pn2o_sdf_particle_energy.argtypes = [_ci, _ci, _vp, _vp, _vp, _vp, _ci, _ci, _cf, _cf, _cf, _cf, _vp]
        lib.pn2o_sdf_nearest.argtypes = [_ci, _ci, _vp, _vp, _vp, _vp, _ci, _ci, _cf, _vp, _vp, _vp]
        lib.pn2o_div_floor.argtypes = [_ci, _vp, _cf, _vp]
        for n in ("pn2o_sdf_trilinear", "pn2o_sdf_particle_energy", "pn2o_sdf_nearest", "pn2o_div_floor"):
            getattr(lib, n).restype = _ci
        _ready = True
    return lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _vol(vol):
    vol = np.ascontiguousarray(vol)
    if vol.dtype not in (np.float16, np.float32):
        raise TypeError("sdf volume must be float16 or float32")
    res = round(vol.size ** (1.0 / 3.0))
    assert res ** 3 == vol.size, "volume must be res^3"
    return vol, int(vol.dtype == np.float16), res


def _p(a):
    return a.ctypes.data_as(_vp)


def distance(V, vol, stride, bbox_min=-0.2, clamp=(-0.05, 0.05)):
    """gf_optimize_obj.Distance (optimization_obj.py:184-228).  V (M,3) -> (M,) fp32."""
    V = _f32(V).reshape(-1, 3)
    vol, f16, res = _vol(vol)
    out = np.empty(V.shape[0], np.float32)
    rc = _lib().pn2o_sdf_trilinear(V.shape[0], _p(V), _p(vol), f16, res, bbox_min, stride, clamp[0], clamp[1], _p(out))
    assert rc == 0, rc
    return out


def particle_energy(pcld, rot, trans, vol, stride, bbox_min=-0.2, clamp=(-0.05, 0.05)):
    """gf_optimize_obj.evaluate (optimization_obj.py:230-237): pcld (N,3), rot (P,3,3), trans (P,3) -> sdf_energy (P,)."""
    pcld = _f32(pcld).reshape(-1, 3)
    rot = _f32(rot).reshape(-1, 3, 3)
    trans = _f32(trans).reshape(-1, 3)
    vol, f16, res = _vol(vol)
    out = np.empty(rot.shape[0], np.float32)
    rc = _lib().pn2o_sdf_particle_energy(rot.shape[0], pcld.shape[0], _p(pcld), _p(rot), _p(trans), _p(vol), f16, res,
                                         bbox_min, stride, clamp[0], clamp[1], _p(out))
    assert rc == 0, rc
    return out


def nearest(hand, obj_r, obj_t, vol, voxel_scale):
    """gf_optimize_hand_pose.query_sdf + get_penetration_loss (optimization_hand.py:252-268).
    hand (B,N,3) -> (flat voxel index (B,N) int32, sdf (B,N) vol dtype, penetration (B,) vol dtype)."""
    hand = _f32(hand)
    B, N, _ = hand.shape
    obj_r = _f32(obj_r).reshape(3, 3)
    obj_t = _f32(obj_t).reshape(3)
    vol, f16, res = _vol(vol)
    idx = np.empty((B, N), np.int32)
    sdf = np.empty((B, N), vol.dtype)
    pen = np.empty((B,), vol.dtype)
    rc = _lib().pn2o_sdf_nearest(B, N, _p(hand), _p(obj_r), _p(obj_t), _p(vol), f16, res, voxel_scale, _p(idx), _p(sdf), _p(pen))
    if rc != 0:
        raise ValueError("query_sdf: voxel index out of range (the reference's asserts, optimization_hand.py:258-260)")
    return idx, sdf, pen


def div_floor(a, b):
    a = _f32(a).reshape(-1)
    out = np.empty_like(a)
    _lib().pn2o_div_floor(a.size, _p(a), float(b), _p(out))
    return out


# ---- gf_optimize_obj.optimize (optimization_obj.py:244-301), update_shape_flag = False ----------------------------
_F = np.float32


def quat_to_matrix(q):
    """unit_quaternion_to_matrix (pose_utils/rotations.py:105-113), fp32 op for op.  q (...,4) w,x,y,z."""
    q = q.astype(_F)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    two = _F(2)
    one = _F(1)
    m = np.stack((one - two * y * y - two * z * z, two * x * y - two * z * w, two * x * z + two * y * w,
                  two * x * y + two * z * w, one - two * x * x - two * z * z, two * y * z - two * x * w,
                  two * x * z - two * y * w, two * y * z + two * x * w, one - two * x * x - two * y * y), axis=-1)
    return m.reshape(q.shape[:-1] + (3, 3)).astype(_F)


def _mat3(a, b):
    """(…,3,3) @ (…,3,3) with the chain o_ij = fma(a_i2, b_2j, fma(a_i1, b_1j, a_i0*b_0j)) (evaluated in fp64-free fp32)."""
    a = a.astype(_F)
    b = b.astype(_F)
    out = np.empty(np.broadcast_shapes(a.shape, b.shape), _F)
    for i in range(3):
        for j in range(3):
            t = a[..., i, 0] * b[..., 0, j]
            t = _fma(a[..., i, 1], b[..., 1, j], t)
            out[..., i, j] = _fma(a[..., i, 2], b[..., 2, j], t)
    return out


def _fma(a, b, c):
    # exact fused multiply-add for fp32 operands: the fp64 product of two fp32 values is exact and the fp64
    # sum carries >= 2*24+2 bits, so rounding it to fp32 equals a single fp32 rounding (no double rounding).
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(_F)


def ortho6d_rows(R):
    """compute_rotation_matrix_from_ortho6d(R.reshape(-1,9)[:, :6]).transpose(-1,-2) (rotations.py:328-369) for one matrix."""
    R = R.astype(_F).reshape(3, 3)

    def normalize(v):
        mag = np.sqrt((v * v).sum(dtype=_F)).astype(_F)
        if not mag > _F(1e-8):
            return np.array([1, 0, 0], _F)
        return (v / max(mag, _F(1e-8))).astype(_F)

    def cross(u, v):
        return np.array([u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]], _F)

    x = normalize(R[0])
    z = normalize(cross(x, R[1]))
    y = cross(z, x)
    return np.stack([x, y, z]).astype(_F)


def obj_optimize(pcld, rotation, translation, pre_sampled, vol, stride, iterations=10, c1=0.02, c2=2.0, beta=0.9,
                 trace=None):
    """The particle loop of gf_optimize_obj.optimize.  pcld (N,3); rotation (3,3); translation (3,);
    pre_sampled (P,6) with row 0 == 0.  Returns (rotation (3,3), translation (3,))."""
    pcld = _f32(pcld).reshape(-1, 3)
    rotation = _f32(rotation).reshape(3, 3).copy()
    translation = _f32(translation).reshape(3).copy()
    pre = _f32(pre_sampled)
    search = np.full(6, c1, _F)
    prev_search = float(c1)  # a Python float until the first successful iteration (:250-251)
    prev_success = True
    for it in range(iterations):
        part = (pre * search[None, :]).astype(_F)  # :259
        qw = np.sqrt(_F(1) - part[:, 0] ** 2 - part[:, 1] ** 2 - part[:, 2] ** 2).astype(_F)  # :260
        sample = np.concatenate([qw[:, None], part], axis=1).astype(_F)  # (P,7)
        new_r = _mat3(rotation[None], quat_to_matrix(sample[:, :4]))  # :263
        new_t = (translation[None, :] + sample[:, 4:]).astype(_F)  # :264
        sdf_energy = particle_energy(pcld, new_r, new_t, vol, stride)  # :267
        energy = (sdf_energy * _F(500)).astype(_F)
        origin = energy[0]
        better = energy < origin  # :271
        weight = ((origin - energy) * better).astype(_F)
        wsum = _F(weight.sum(dtype=np.float64)) + _F(1e-5)
        success = bool(better.any())
        if success:
            mean_sdf = _F((sdf_energy.astype(np.float64) * weight).sum()) / wsum
            mt = ((sample.astype(np.float64) * weight[:, None]).sum(axis=0)).astype(_F) / wsum  # (7,)
            mt = mt.astype(_F)
            mt[:4] = mt[:4] / (np.sqrt((mt[:4] * mt[:4]).sum(dtype=_F)).astype(_F) + _F(1e-8))
            rotation = ortho6d_rows(_mat3(rotation, quat_to_matrix(mt[:4])))  # :285-287
            translation = (translation + mt[4:]).astype(_F)
        else:
            mean_sdf = sdf_energy[0]
            mt = np.zeros(7, _F)
        s = (np.abs(mt[1:]) + _F(1e-3)).astype(_F)  # update_seach_size :239-242
        new_search = (mean_sdf * _F(c2) * s / np.sqrt((s * s).sum(dtype=_F)).astype(_F) + _F(1e-3)).astype(_F)
        search = new_search
        if prev_success and success:  # :294-299
            carry = _F((1 - beta) * prev_search) if isinstance(prev_search, float) else _F(1 - beta) * prev_search
            search = (_F(beta) * search + carry).astype(_F)
            prev_search = search
        elif success:
            prev_search = search
        prev_success = success
        if trace is not None:
            trace.append(dict(sdf_energy=sdf_energy, success=success, search=search.copy(), rotation=rotation.copy(),
                              translation=translation.copy()))
    return rotation, translation
