"""pointnet2_hip -- drop-in for the reference's native module `pointnet2_cuda`.

The reference builds a pybind11 module with ten functions
(network/models/pointnet_lib/src/pointnet2_api.cpp:11-24) that take the problem sizes as
ints and caller-allocated torch tensors.  This module exports the same ten names with the
same positional arguments and binds them over the C ABI of libpn2_hip.so
(include/pn2_hip.h) -- raw device pointers plus the current torch HIP stream, the stream the
reference obtained with at::cuda::getCurrentCUDAStream() (e.g. sampling.cpp:45).

Differences from the reference, all on the safe side:
  * every tensor is validated (device, dtype, contiguity, element count against the ints);
    the reference checked only ball_query's inputs (ball_query.cpp:10-17);
  * failures raise Python exceptions; the reference printed and exit(-1)ed
    (e.g. sampling_gpu.cu:39-43);
  * there is no CPU path: a CPU tensor raises, a missing library fails the import.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PN2_LIB_PATH: a tuning / profiling VARIANT of the library built by hotrack_amd._build.build_variant (probes only; the product
# and the tests load the in-tree build)
LIB_PATH = os.environ.get("PN2_LIB_PATH") or os.path.join(_HERE, "libpn2_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the HIP extension first "
        "(python -c 'import __graft_entry__ as g; g.build()' or python -m hotrack_amd._build). "
        "hotrack_amd has no CPU fallback."
    )

_lib = ctypes.CDLL(LIB_PATH)

_vp, _ci, _cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_SIGS = {
    "pn2_furthest_point_sampling": [_ci, _ci, _ci, _vp, _vp, _vp, _vp],
    "pn2_ball_query": [_ci, _ci, _ci, _cf, _ci, _vp, _vp, _vp, _vp],
    "pn2x_ball_query_grid": [_ci, _ci, _ci, _cf, _ci, _vp, _vp, _vp, _vp, ctypes.c_long, _vp],
    "pn2_group_points": [_ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp],
    "pn2_group_points_grad": [_ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp],
    "pn2_gather_points": [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp],
    "pn2_gather_points_grad": [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp],
    "pn2_knn": [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp],
    "pn2_three_nn": [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp],
    "pn2_three_interpolate": [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp],
    "pn2_three_interpolate_grad": [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp],
}
for _name, _args in _SIGS.items():
    _fn = getattr(_lib, _name)
    _fn.argtypes = _args
    _fn.restype = _ci
_lib.pn2_strerror.restype = ctypes.c_char_p
_lib.pn2_strerror.argtypes = [_ci]
_lib.pn2_abi_version.restype = _ci
_lib.pn2_last_hip_error.restype = _ci

_lib.pn2x_ball_query_grid_scratch_words.argtypes = [_ci, _ci]
_lib.pn2x_ball_query_grid_scratch_words.restype = ctypes.c_long
_lib.pn2x_scatter_cm_scratch_ints.argtypes = [_ci, _ci, _ci, _ci]
_lib.pn2x_scatter_cm_scratch_ints.restype = ctypes.c_long
_lib.pn2x_scatter_cm.argtypes = [_ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, ctypes.c_long, _vp]
_lib.pn2x_scatter_cm.restype = _ci
_PN2_ERANGE = -3
PN2_ESCRATCH = -4  # include/pn2_hip.h: this size needs the optional scratch buffer (FPS temp)


def _scatter_cm(t, b, c, n_dst, m_src, grad_out, idx_ptr, w_ptr, out_ptr, dev, stream):
    """Atomics-free backward with torch-allocated scratch (pn2_ext.h: pn2x_scatter_cm) -- capture-safe.  False when the
    shape is not covered: the caller then uses the reference-signature entry."""
    need = _lib.pn2x_scatter_cm_scratch_ints(t, b, n_dst, m_src)
    if need < 0:
        return False
    scratch = torch.empty(max(int(need), 1), dtype=torch.int32, device=dev)
    rc = _lib.pn2x_scatter_cm(t, b, c, n_dst, m_src, grad_out, idx_ptr, w_ptr, out_ptr, scratch.data_ptr(), need, stream)
    if rc == _PN2_ERANGE:
        return False
    _check(rc, "scatter_cm")
    return True


ABI_VERSION = _lib.pn2_abi_version()
KNN_MAX_K = 200  # interpolate_gpu.cu:30-31


# Optional profiling hook (bench.py): when set to a list, every C-ABI call appends
# (name, start_event, end_event) with the two HIP events recorded IMMEDIATELY around the enqueue on the
# launching stream, so the bracket holds the kernel and ~nothing else.
PROFILE = None


def _call(fn, name, stream_handle, *args):
    if PROFILE is None:
        return fn(*args)
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    rc = fn(*args)
    e.record()
    PROFILE.append((name, args, s, e))
    return rc


class Pn2Error(RuntimeError):
    """A C-ABI call returned a negative PN2_E* code."""


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = _lib.pn2_strerror(rc).decode()
        extra = f" (hipError {_lib.pn2_last_hip_error()})" if rc == -5 else ""
        raise Pn2Error(f"{what}: {msg}{extra} [code {rc}]")


def _ptr(t: torch.Tensor, name: str, dtype: torch.dtype, numel: int) -> int:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU (HIP) tensor -- hotrack_amd has no CPU path (got device {t.device})")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if t.numel() != numel:
        raise ValueError(f"{name} has {t.numel()} elements, the size arguments imply {numel}")
    return t.data_ptr()


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


_f32, _i32 = torch.float32, torch.int32


def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    """points (b,n,3) f32, temp (b,n) f32 scratch or None, idx (b,m) int32 out.  sampling.cpp:38-49."""
    p = _ptr(points, "points", _f32, b * n * 3)
    o = _ptr(idx, "idx", _i32, b * m)
    t = None if temp is None else _ptr(temp, "temp", _f32, b * n)
    with torch.cuda.device(points.device):
        _check(_call(_lib.pn2_furthest_point_sampling, "fps_kernel", None, b, n, m, p, t, o, _stream(points)), "furthest_point_sampling")
    return 1


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    """new_xyz (b,m,3), xyz (b,n,3), idx (b,m,nsample) int32 out.  ball_query.cpp:14-24."""
    pn = _ptr(new_xyz, "new_xyz", _f32, b * m * 3)
    px = _ptr(xyz, "xyz", _f32, b * n * 3)
    o = _ptr(idx, "idx", _i32, b * m * nsample)
    with torch.cuda.device(xyz.device):
        if n >= 4096 and m * n >= (1 << 22) and radius > 0:
            # large problem: the cell-grid search with torch-owned scratch (safe under HIP-graph capture); identical output
            words = int(_lib.pn2x_ball_query_grid_scratch_words(b, n))
            scratch = torch.empty(words, dtype=torch.int32, device=xyz.device)
            rc = _lib.pn2x_ball_query_grid(b, n, m, float(radius), nsample, pn, px, o, scratch.data_ptr(), words, _stream(xyz))
            if rc != -3:  # PN2_ERANGE: shape not covered -> the scan below
                _check(rc, "ball_query_grid")
                return 1
        _check(_lib.pn2_ball_query(b, n, m, float(radius), nsample, pn, px, o, _stream(xyz)), "ball_query")
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    """points (b,c,n), idx (b,npoints,nsample) int32, out (b,c,npoints,nsample).  group_points.cpp:25-37."""
    p = _ptr(points, "points", _f32, b * c * n)
    i = _ptr(idx, "idx", _i32, b * npoints * nsample)
    o = _ptr(out, "out", _f32, b * c * npoints * nsample)
    with torch.cuda.device(points.device):
        _check(_lib.pn2_group_points(b, c, n, npoints, nsample, p, i, o, _stream(points)), "group_points")
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    """grad_out (b,c,npoints,nsample), grad_points (b,c,n) accumulated into.  group_points.cpp:11-23."""
    g = _ptr(grad_out, "grad_out", _f32, b * c * npoints * nsample)
    i = _ptr(idx, "idx", _i32, b * npoints * nsample)
    o = _ptr(grad_points, "grad_points", _f32, b * c * n)
    with torch.cuda.device(grad_out.device):
        if not _scatter_cm(1, b, c, n, npoints * nsample, g, i, None, o, grad_out.device, _stream(grad_out)):
            _check(_lib.pn2_group_points_grad(b, c, n, npoints, nsample, g, i, o, _stream(grad_out)), "group_points_grad")
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    """points (b,c,n), idx (b,npoints) int32, out (b,c,npoints).  sampling.cpp:11-22."""
    p = _ptr(points, "points", _f32, b * c * n)
    i = _ptr(idx, "idx", _i32, b * npoints)
    o = _ptr(out, "out", _f32, b * c * npoints)
    with torch.cuda.device(points.device):
        _check(_lib.pn2_gather_points(b, c, n, npoints, p, i, o, _stream(points)), "gather_points")
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    """grad_out (b,c,npoints), grad_points (b,c,n) accumulated into.  sampling.cpp:24-35."""
    g = _ptr(grad_out, "grad_out", _f32, b * c * npoints)
    i = _ptr(idx, "idx", _i32, b * npoints)
    o = _ptr(grad_points, "grad_points", _f32, b * c * n)
    with torch.cuda.device(grad_out.device):
        if not _scatter_cm(1, b, c, n, npoints, g, i, None, o, grad_out.device, _stream(grad_out)):
            _check(_lib.pn2_gather_points_grad(b, c, n, npoints, g, i, o, _stream(grad_out)), "gather_points_grad")
    return 1


def knn_wrapper(b, n, m, k, unknown, known, dist2, idx):
    """unknown (b,n,3), known (b,m,3) -> dist2 (b,n,k) squared, idx (b,n,k).  interpolate.cpp:26-36."""
    if not 1 <= k <= KNN_MAX_K:
        raise ValueError(f"knn: k must be in [1, {KNN_MAX_K}] (reference keeps best[200] per thread), got {k}")
    u = _ptr(unknown, "unknown", _f32, b * n * 3)
    kn = _ptr(known, "known", _f32, b * m * 3)
    d = _ptr(dist2, "dist2", _f32, b * n * k)
    i = _ptr(idx, "idx", _i32, b * n * k)
    with torch.cuda.device(unknown.device):
        _check(_lib.pn2_knn(b, n, m, k, u, kn, d, i, _stream(unknown)), "knn")


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    """unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) squared, idx (b,n,3).  interpolate.cpp:14-24."""
    u = _ptr(unknown, "unknown", _f32, b * n * 3)
    kn = _ptr(known, "known", _f32, b * m * 3)
    d = _ptr(dist2, "dist2", _f32, b * n * 3)
    i = _ptr(idx, "idx", _i32, b * n * 3)
    with torch.cuda.device(unknown.device):
        _check(_lib.pn2_three_nn(b, n, m, u, kn, d, i, _stream(unknown)), "three_nn")


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    """points (b,c,m), idx/weight (b,n,3) -> out (b,c,n).  interpolate.cpp:39-53."""
    p = _ptr(points, "points", _f32, b * c * m)
    i = _ptr(idx, "idx", _i32, b * n * 3)
    w = _ptr(weight, "weight", _f32, b * n * 3)
    o = _ptr(out, "out", _f32, b * c * n)
    with torch.cuda.device(points.device):
        _check(_lib.pn2_three_interpolate(b, c, m, n, p, i, w, o, _stream(points)), "three_interpolate")


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    """grad_out (b,c,n) -> grad_points (b,c,m) accumulated into.  interpolate.cpp:55-69."""
    g = _ptr(grad_out, "grad_out", _f32, b * c * n)
    i = _ptr(idx, "idx", _i32, b * n * 3)
    w = _ptr(weight, "weight", _f32, b * n * 3)
    o = _ptr(grad_points, "grad_points", _f32, b * c * m)
    with torch.cuda.device(grad_out.device):
        if not _scatter_cm(3, b, c, m, n, g, i, w, o, grad_out.device, _stream(grad_out)):
            _check(_lib.pn2_three_interpolate_grad(b, c, n, m, g, i, w, o, _stream(grad_out)), "three_interpolate_grad")


EXPORTED = (
    "ball_query_wrapper", "group_points_wrapper", "group_points_grad_wrapper", "gather_points_wrapper",
    "gather_points_grad_wrapper", "furthest_point_sampling_wrapper", "knn_wrapper", "three_nn_wrapper",
    "three_interpolate_wrapper", "three_interpolate_grad_wrapper",
)
