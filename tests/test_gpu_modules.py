"""GPU: behaviour of the grouping modules of the operator API -- QueryAndGroup, GroupAll, KNNAndGroup
(reference network/models/pointnet_lib/pointnet2_utils.py:275-387) -- against the CPU oracle's operators composed
the way the reference composes them: channel order, use_xyz, features=None, centre subtraction, caller-supplied idx,
and gradients (to features only; scatter-add of the grouped gradient)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
B, N, M, C = 3, 500, 40, 7


@pytest.fixture(scope="module")
def data():
    rng = np.random.default_rng(11)
    xyz = rng.random((B, N, 3), dtype=np.float32)
    new_xyz = xyz[:, rng.permutation(N)[:M]].copy()
    new_xyz[:, -1] = 5.0  # a centroid with no point in any ball: all-zero row (ball_query_gpu.cu: idx pre-zeroed)
    feat = rng.normal(size=(B, C, N)).astype(np.float32)
    return xyz, new_xyz, feat


def _d(a):
    return torch.from_numpy(a).cuda()


def _group(oracle, a, idx):
    return oracle.group_points(np.ascontiguousarray(a), idx)


@pytest.mark.parametrize("use_xyz", [True, False])
def test_query_and_group(data, oracle, use_xyz):
    from hotrack_amd import pointnet2_utils as ops
    xyz, new_xyz, feat = data
    radius, K = 0.15, 16
    idx = oracle.ball_query(radius, K, xyz, new_xyz)
    gxyz = _group(oracle, xyz.transpose(0, 2, 1), idx) - new_xyz.transpose(0, 2, 1)[..., None]
    gfeat = _group(oracle, feat, idx)
    mod = ops.QueryAndGroup(radius, K, use_xyz=use_xyz)
    out = mod(_d(xyz), _d(new_xyz), _d(feat)).cpu().numpy()
    want = np.concatenate([gfeat, gxyz], 1) if use_xyz else gfeat  # features FIRST (:303)
    assert out.shape == want.shape == (B, C + 3 if use_xyz else C, M, K)
    np.testing.assert_array_equal(out, want)
    assert (idx[:, -1] == 0).all()  # the far centroid groups point 0 sixteen times
    if use_xyz:
        np.testing.assert_array_equal(mod(_d(xyz), _d(new_xyz), None).cpu().numpy(), gxyz)
    else:
        with pytest.raises(AssertionError):
            mod(_d(xyz), _d(new_xyz), None)


def test_query_and_group_backward(data, oracle):
    from hotrack_amd import pointnet2_utils as ops
    xyz, new_xyz, feat = data
    radius, K = 0.15, 16
    f = _d(feat).requires_grad_(True)
    x = _d(xyz).requires_grad_(True)
    out = ops.QueryAndGroup(radius, K)(x, _d(new_xyz), f)
    go = torch.randn_like(out)
    out.backward(go)
    idx = oracle.ball_query(radius, K, xyz, new_xyz)
    want = oracle.group_points_grad(go[:, :C].contiguous().cpu().numpy(), idx, N)
    np.testing.assert_allclose(f.grad.cpu().numpy(), want, atol=1e-5)
    # xyz receives the gradient of its grouped copy too (grouping_operation is differentiable in its features argument)
    wantx = oracle.group_points_grad(go[:, C:].contiguous().cpu().numpy(), idx, N)
    np.testing.assert_allclose(x.grad.cpu().numpy(), wantx.transpose(0, 2, 1), atol=1e-5)


@pytest.mark.parametrize("use_xyz", [True, False])
def test_group_all(data, use_xyz):
    from hotrack_amd import pointnet2_utils as ops
    xyz, new_xyz, feat = data
    mod = ops.GroupAll(use_xyz=use_xyz)
    out = mod(_d(xyz), None, _d(feat)).cpu().numpy()
    want = np.concatenate([xyz.transpose(0, 2, 1), feat], 1)[:, :, None] if use_xyz else feat[:, :, None]  # xyz FIRST (:329)
    assert out.shape == (B, 3 + C if use_xyz else C, 1, N)
    np.testing.assert_array_equal(out, want)
    np.testing.assert_array_equal(mod(_d(xyz), None, None).cpu().numpy(), xyz.transpose(0, 2, 1)[:, :, None])


@pytest.mark.parametrize("use_xyz", [True, False])
def test_knn_and_group(data, oracle, use_xyz):
    from hotrack_amd import pointnet2_utils as ops
    xyz, new_xyz, feat = data
    K = 12
    _, idx = oracle.knn(K, new_xyz, xyz)
    gxyz = _group(oracle, xyz.transpose(0, 2, 1), idx) - new_xyz.transpose(0, 2, 1)[..., None]
    gfeat = _group(oracle, feat, idx)
    mod = ops.KNNAndGroup(0.0, K, use_xyz=use_xyz)
    out = mod(_d(xyz), _d(new_xyz), None, _d(feat)).cpu().numpy()
    want = np.concatenate([gxyz, gfeat], 1) if use_xyz else gfeat  # xyz FIRST (:379)
    np.testing.assert_array_equal(out, want)
    # caller-supplied neighbour lists are used as given (:361-363)
    rev = np.ascontiguousarray(idx[:, :, ::-1])
    out2 = mod(_d(xyz), _d(new_xyz), _d(rev), _d(feat)).cpu().numpy()
    np.testing.assert_array_equal(out2, want[..., ::-1])
    # new_xyz=None: every point is its own centre (:358-359); nearest neighbour of a point is itself -> zero offset
    if use_xyz:
        self_out = mod(_d(xyz)).cpu().numpy()
        assert self_out.shape == (B, 3, N, K)
        assert np.abs(self_out[..., 0]).max() == 0.0
