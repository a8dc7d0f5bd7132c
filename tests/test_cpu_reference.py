"""The cpu_baseline port (oracle/cpu_reference.py) against the IMPORTED reference fallback.
Runs only where /root/reference exists (the build container); skipped on the GPU box."""
import os
import sys

import pytest
import torch

from _cases import cloud

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref_pu():
    sys.path[:0] = [REF, os.path.join(REF, "network"), os.path.join(REF, "network", "models")]
    mod = __import__("pointnet_utils")
    assert mod.__file__.startswith(REF) and mod.CUDA is False
    yield mod
    for p in sys.path[:3]:
        if p.startswith(REF):
            sys.path.remove(p)
    sys.modules.pop("pointnet_utils", None)


def test_port_equals_reference_fallback(ref_pu):
    from oracle import cpu_reference as P
    torch.manual_seed(0)
    xyz = torch.from_numpy(cloud(1, 2, 512, "hand"))
    orig = torch.randint
    torch.randint = lambda lo, hi, size, **kw: torch.zeros(size, dtype=torch.long)
    try:
        ref_idx = ref_pu.farthest_point_sample(xyz, 64)
    finally:
        torch.randint = orig
    idx = P.furthest_point_sample(xyz, 64)
    assert torch.equal(idx.long(), ref_idx)
    new = torch.gather(xyz, 1, ref_idx.unsqueeze(-1).expand(-1, -1, 3))
    assert torch.equal(P.ball_query(0.2, 16, xyz, new).long(), ref_pu.query_ball_point(0.2, 16, xyz, new))
    d, i = P.three_nn(xyz, new)
    rd, ri = ref_pu.three_nn(xyz, new)
    assert torch.equal(i.long(), ri) and torch.equal(d, rd)
    d, i = P.knn(8, new[:, :21].contiguous(), xyz)
    rd, ri = ref_pu.knn_point(8, new[:, :21].contiguous(), xyz)
    assert torch.equal(i.long(), ri) and torch.allclose(d, rd, equal_nan=True)
    f = torch.randn(2, 7, 512)
    gi = torch.randint(0, 512, (2, 9, 4))
    assert torch.equal(P.grouping_operation(f, gi), ref_pu.group_operation(f, gi))
    assert torch.equal(P.gather_operation(f, gi[:, :, 0]), ref_pu.gather_operation(f, gi[:, :, 0]))
    w = torch.rand(2, 64, 3)
    ii = torch.randint(0, 512, (2, 64, 3))
    assert torch.allclose(P.three_interpolate(f, ii, w), ref_pu.three_interpolate(f, ii, w))
