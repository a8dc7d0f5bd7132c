"""Seeded synthetic inputs for the SDF-lookup row (analytic signed-distance volumes, surface points, poses).
Used by tests/golden/make_golden_sdf.py, the SDF tests and scripts/bench_sdf.py -- numpy only."""
from __future__ import annotations

import numpy as np


def _sdf(p, shape):
    """Analytic signed distance (object frame, metres).  p (...,3) float64."""
    if shape == "sphere":
        return np.linalg.norm(p, axis=-1) - 0.08
    if shape == "box":
        q = np.abs(p) - np.array([0.07, 0.05, 0.09])
        return np.linalg.norm(np.maximum(q, 0), axis=-1) + np.minimum(q.max(axis=-1), 0)
    if shape == "capsule":
        z = np.clip(p[..., 2], -0.07, 0.07)
        c = np.stack([np.zeros_like(z), np.zeros_like(z), z], axis=-1)
        return np.linalg.norm(p - c, axis=-1) - 0.04
    raise ValueError(shape)


def make_volume(res, stride, shape, dtype=np.float16, centre_index=None):
    """res^3 volume, element (ix,iy,iz) = sdf at ((i - centre_index) * stride), clamped to +-0.1 (the reference clamps
    its decoded volumes the same way), row-major ix, iy, iz like the reference's `volume_ind`."""
    c = res // 2 if centre_index is None else centre_index
    ax = (np.arange(res) - c) * float(stride)
    g = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), axis=-1)
    return np.clip(_sdf(g, shape), -0.1, 0.1).astype(dtype).reshape(-1)


def object_points(seed, n, shape, noise=0.0015):
    """n points on the zero level set (+ sensor noise), object frame, float32."""
    rng = np.random.default_rng(seed)
    p = rng.uniform(-0.1, 0.1, (n, 3))
    h = 1e-5
    for _ in range(12):  # project along the numerical gradient
        d = _sdf(p, shape)
        g = np.stack([(_sdf(p + h * e, shape) - _sdf(p - h * e, shape)) / (2 * h) for e in np.eye(3)], axis=-1)
        p = p - d[:, None] * g / np.maximum((g * g).sum(-1, keepdims=True), 1e-12)
    return (p + rng.normal(0, noise, p.shape)).astype(np.float32)


def _axis_angle(axis, angle):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


def random_pose(seed, angle=None, trans=None):
    """(R (3,3), t (3,)) float32.  Default: arbitrary rotation, t near (0,0,0.5) m; angle/trans given -> a small delta."""
    rng = np.random.default_rng(seed)
    R = _axis_angle(rng.standard_normal(3), rng.uniform(0, np.pi) if angle is None else angle)
    if trans is None:
        t = np.array([0.0, 0.0, 0.5]) + rng.uniform(-0.05, 0.05, 3)
    else:
        v = rng.standard_normal(3)
        t = v / np.linalg.norm(v) * trans
    return R.astype(np.float32), t.astype(np.float32)


def particles(seed, p, R0, t0, angle=0.05, trans=0.01):
    """p candidate poses around (R0, t0); particle 0 is (R0, t0) itself."""
    rng = np.random.default_rng(seed)
    rot = np.empty((p, 3, 3), np.float32)
    tr = np.empty((p, 3), np.float32)
    for i in range(p):
        if i == 0:
            rot[i], tr[i] = R0, t0
        else:
            rot[i] = (R0.astype(np.float64) @ _axis_angle(rng.standard_normal(3), rng.normal(0, angle))).astype(np.float32)
            tr[i] = t0 + rng.normal(0, trans, 3)
    return rot, tr


def hand_particles(seed, b, n, R0, t0, extent):
    """(b,n,3) camera-frame points whose object-frame coordinates are uniform in +-extent (some beyond the volume)."""
    rng = np.random.default_rng(seed)
    p_obj = rng.uniform(-extent, extent, (b, n, 3))
    return (p_obj @ R0.astype(np.float64).T + t0).astype(np.float32)


def hand_pose_particles(seed, b, n, R0, t0, jitter_t=0.01, jitter_r=0.05):
    """(b,n,3) camera-frame vertices of b candidate hands: one hand-sized blob of n vertices (16 x 8 x 4 cm ellipsoid
    touching the object) under b small rigid perturbations -- the access pattern of the hand optimiser's particles
    (optimization_hand.py:155-160 scales N(0,1) samples by a ~1 cm / few-degree search size)."""
    rng = np.random.default_rng(seed)
    u = rng.standard_normal((n, 3))
    u = u / np.linalg.norm(u, axis=1, keepdims=True) * rng.uniform(0, 1, (n, 1)) ** (1 / 3)
    verts = u * np.array([0.08, 0.04, 0.02]) + np.array([0.0, 0.06, 0.03])      # object frame, metres
    out = np.empty((b, n, 3), np.float32)
    for i in range(b):
        dR = _axis_angle(rng.standard_normal(3), rng.normal(0, jitter_r)) if i else np.eye(3)
        dt = rng.normal(0, jitter_t, 3) if i else np.zeros(3)
        p_obj = verts @ dR.T + dt
        out[i] = (p_obj @ R0.astype(np.float64).T + t0).astype(np.float32)
    return out
