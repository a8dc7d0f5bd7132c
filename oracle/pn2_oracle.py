"""ctypes loader + numpy front-end for the CPU oracle (oracle/pn2_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by anything under hotrack_amd/ or network/.
Parity is "unpinned" by the reference's own tests (it has none); see pn2_oracle.c header.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpn2_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (seconds).  Returns the .so path."""
    srcs = [os.path.join(_HERE, f) for f in ("pn2_oracle.c", "sdf_oracle.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libpn2_oracle.so"])
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        for name in (
            "pn2o_furthest_point_sampling", "pn2o_furthest_point_sampling_keyed", "pn2o_ball_query",
            "pn2o_knn", "pn2o_three_nn", "pn2o_group_points", "pn2o_group_points_grad",
            "pn2o_gather_points", "pn2o_gather_points_grad", "pn2o_three_interpolate",
            "pn2o_three_interpolate_grad", "pn2o_opt_n_threads",
        ):
            getattr(_lib, name).restype = ctypes.c_int
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def _chk(rc, what):
    if rc != 0:
        raise ValueError(f"oracle {what} failed with code {rc}")


def opt_n_threads(n: int) -> int:
    return lib().pn2o_opt_n_threads(int(n))


def furthest_point_sample(xyz, npoint: int, keyed: bool = False):
    """xyz (B,N,3) f32 -> idx (B,npoint) int32.  keyed=True uses the closed-form tie key."""
    xyz, px = _f(xyz)
    B, N, _ = xyz.shape
    out = np.empty((B, npoint), dtype=np.int32)
    if keyed:
        rc = lib().pn2o_furthest_point_sampling_keyed(B, N, npoint, px, out.ctypes.data_as(_i32p))
    else:
        rc = lib().pn2o_furthest_point_sampling(B, N, npoint, px, None, out.ctypes.data_as(_i32p))
    _chk(rc, "fps")
    return out


def ball_query(radius: float, nsample: int, xyz, new_xyz):
    """xyz (B,N,3), new_xyz (B,S,3) -> idx (B,S,nsample) int32."""
    xyz, px = _f(xyz)
    new_xyz, pn = _f(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    out = np.empty((B, S, nsample), dtype=np.int32)
    _chk(lib().pn2o_ball_query(B, N, S, ctypes.c_float(radius), nsample, pn, px,
                               out.ctypes.data_as(_i32p)), "ball_query")
    return out


def knn(k: int, unknown, known):
    """-> (dist2 (B,n,k) f32 SQUARED, idx (B,n,k) int32)."""
    unknown, pu = _f(unknown)
    known, pk = _f(known)
    B, n, _ = unknown.shape
    m = known.shape[1]
    d = np.empty((B, n, k), dtype=np.float32)
    i = np.empty((B, n, k), dtype=np.int32)
    _chk(lib().pn2o_knn(B, n, m, k, pu, pk, d.ctypes.data_as(_f32p), i.ctypes.data_as(_i32p)), "knn")
    return d, i


def three_nn(unknown, known):
    """-> (dist2 (B,n,3) f32 SQUARED, idx (B,n,3) int32)."""
    unknown, pu = _f(unknown)
    known, pk = _f(known)
    B, n, _ = unknown.shape
    m = known.shape[1]
    d = np.empty((B, n, 3), dtype=np.float32)
    i = np.empty((B, n, 3), dtype=np.int32)
    _chk(lib().pn2o_three_nn(B, n, m, pu, pk, d.ctypes.data_as(_f32p), i.ctypes.data_as(_i32p)), "three_nn")
    return d, i


def group_points(points, idx):
    """points (B,C,N), idx (B,P,S) -> (B,C,P,S)."""
    points, pp = _f(points)
    idx, pi = _i(idx)
    B, C, N = points.shape
    _, P, S = idx.shape
    out = np.empty((B, C, P, S), dtype=np.float32)
    _chk(lib().pn2o_group_points(B, C, N, P, S, pp, pi, out.ctypes.data_as(_f32p)), "group")
    return out


def group_points_grad(grad_out, idx, N: int):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, C, P, S = grad_out.shape
    out = np.zeros((B, C, N), dtype=np.float32)
    _chk(lib().pn2o_group_points_grad(B, C, N, P, S, pg, pi, out.ctypes.data_as(_f32p)), "group_grad")
    return out


def gather_points(points, idx):
    """points (B,C,N), idx (B,M) -> (B,C,M)."""
    points, pp = _f(points)
    idx, pi = _i(idx)
    B, C, N = points.shape
    M = idx.shape[1]
    out = np.empty((B, C, M), dtype=np.float32)
    _chk(lib().pn2o_gather_points(B, C, N, M, pp, pi, out.ctypes.data_as(_f32p)), "gather")
    return out


def gather_points_grad(grad_out, idx, N: int):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, C, M = grad_out.shape
    out = np.zeros((B, C, N), dtype=np.float32)
    _chk(lib().pn2o_gather_points_grad(B, C, N, M, pg, pi, out.ctypes.data_as(_f32p)), "gather_grad")
    return out


def three_interpolate(points, idx, weight):
    """points (B,C,M), idx (B,n,3), weight (B,n,3) -> (B,C,n)."""
    points, pp = _f(points)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, C, M = points.shape
    n = idx.shape[1]
    out = np.empty((B, C, n), dtype=np.float32)
    _chk(lib().pn2o_three_interpolate(B, C, M, n, pp, pi, pw, out.ctypes.data_as(_f32p)), "interp")
    return out


def three_interpolate_grad(grad_out, idx, weight, M: int):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, C, n = grad_out.shape
    out = np.zeros((B, C, M), dtype=np.float32)
    _chk(lib().pn2o_three_interpolate_grad(B, C, n, M, pg, pi, pw, out.ctypes.data_as(_f32p)), "interp_grad")
    return out
