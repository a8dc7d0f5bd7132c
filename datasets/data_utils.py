"""Loader-side point down-sampling (counterpart of the reference's datasets/data_utils.py:226-249).

The reference runs FPS on the GPU from inside every DataLoader worker, one cloud at a time (B=1,
N = min(len, 5*npoint) after a random pre-subsample, M = npoint in {512, 1024}).  Same semantics here on the
MI355X kernel (whose register-resident design covers N up to 16384), plus a batched form: clouds that end up
with the same length after the pre-subsample -- the common case, exactly 5*npoint points -- are sampled in ONE
launch, one workgroup per cloud.  GPU only: without a device the reference
silently degrades to random sampling; here that is an explicit choice of the caller (`allow_random=True`).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch


def _presubsample(xyz: np.ndarray, npoint: int, rng) -> np.ndarray:
    """Indices kept before FPS: a random 5*npoint subset when the cloud is larger (data_utils.py:235-236)."""
    n = len(xyz)
    return rng.permutation(n)[:5 * npoint] if n > 5 * npoint else np.arange(n)


def farthest_point_sample(xyz: np.ndarray, npoint: int, device, rng=None, allow_random: bool = False) -> np.ndarray:
    """xyz (N,3) -> indices (npoint,) into xyz."""
    return farthest_point_sample_batch([xyz], npoint, device, rng, allow_random)[0]


def farthest_point_sample_batch(clouds: Sequence[np.ndarray], npoint: int, device, rng=None,
                                allow_random: bool = False) -> List[np.ndarray]:
    """FPS for several clouds with as few launches as possible (one per distinct post-subsample length)."""
    rng = rng or np.random
    if not torch.cuda.is_available():
        if not allow_random:
            raise RuntimeError("farthest_point_sample needs the GPU operator (pass allow_random=True to get the "
                               "reference's CPU behaviour: random sampling)")
        return [rng.permutation(len(c))[:npoint] for c in clouds]
    from hotrack_amd import pointnet2_utils as ops
    keep = [_presubsample(np.asarray(c), npoint, rng) for c in clouds]
    out: List[np.ndarray] = [None] * len(clouds)
    by_len = {}
    for i, k in enumerate(keep):
        by_len.setdefault(len(k), []).append(i)
    for n, members in by_len.items():
        batch = np.stack([np.asarray(clouds[i], dtype=np.float32)[keep[i]] for i in members])
        idx = ops.furthest_point_sample(torch.from_numpy(batch).to(device), npoint).cpu().numpy()
        for row, i in zip(idx, members):
            out[i] = keep[i][row]
    return out
