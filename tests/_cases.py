"""Shared seeded input generators for oracle / GPU parity tests."""
import numpy as np


def cloud(seed, B, N, kind="uniform"):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((B, N, 3), dtype=np.float32)
    if kind == "hand":  # N(0, 0.05^2) clipped to 0.15 ball, canonicalised by /0.2
        p = rng.normal(0, 0.05, (B, N, 3))
        r = np.linalg.norm(p, axis=-1, keepdims=True)
        p = np.where(r > 0.15, p * 0.15 / np.maximum(r, 1e-9), p)
        return (p / 0.2).astype(np.float32)
    if kind == "lattice":  # many exact fp32 ties
        return rng.integers(0, 5, (B, N, 3)).astype(np.float32) * 0.25
    if kind == "dup":  # few distinct points -> all-zero distance ties
        base = rng.random((B, 7, 3), dtype=np.float32)
        pick = rng.integers(0, 7, (B, N))
        return np.take_along_axis(base, pick[..., None].repeat(3, -1), 1)
    raise ValueError(kind)


def take_points(xyz, idx):
    return np.take_along_axis(xyz, idx[..., None].astype(np.int64).repeat(3, -1), 1)
