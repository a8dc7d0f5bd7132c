"""GPU box: the committed per-operator golden vectors (tests/golden/ops_*.npz -- the vectors the imported reference agreed
with when they were generated, GOLDEN_REPORT.json) fed straight to the HIP operators through the C-ABI, and the oracle that
was rebuilt ON THIS BOX (a different gcc than the build container's) re-pinned against the same fixtures.

Chain closed here:  reference -> fixture (committed)  ==  HIP kernels on this box   (direct, no oracle in between)
                    reference -> fixture (committed)  ==  oracle .so rebuilt on this box  (so every other -m gpu test that
                                                                                           compares HIP with the oracle is anchored)
Index outputs bit-exact, copies bit-exact, float outputs within 1e-5 (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ATOL = 1e-5


def _load(name):
    return np.load(os.path.join(G, name))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def ops():
    from hotrack_amd import pointnet2_utils
    return pointnet2_utils


def test_oracle_rebuilt_on_this_box_matches_the_fixtures(oracle):
    """tests/test_oracle_golden.py's fixture tests (CPU suite) executed against the oracle library of THIS box."""
    import test_oracle_golden as t
    t.test_golden_report_says_oracle_agreed_with_reference_fallback()
    t.test_fps_golden(oracle)
    t.test_ball_query_golden(oracle)
    t.test_nn_golden(oracle)
    t.test_group_interp_golden(oracle)


def test_hip_fps_against_golden(ops):
    g = _load("ops_fps.npz")
    n = len([k for k in g.files if k.endswith("_xyz")])
    assert n >= 10
    for i in range(n):
        xyz, idx = g[f"fps{i}_xyz"], g[f"fps{i}_idx"]
        got = ops.furthest_point_sample(dev(xyz), idx.shape[1]).cpu().numpy()
        np.testing.assert_array_equal(got, idx, err_msg=f"fps case {i} {xyz.shape}")


def test_hip_ball_query_against_golden(ops):
    g = _load("ops_ball_query.npz")
    n = len([k for k in g.files if k.endswith("_xyz")])
    assert n >= 4
    for i in range(n):
        r, K = g[f"bq{i}_rk"]
        got = ops.ball_query(float(r), int(K), dev(g[f"bq{i}_xyz"]), dev(g[f"bq{i}_new"])).cpu().numpy()
        np.testing.assert_array_equal(got, g[f"bq{i}_idx"], err_msg=f"ball query case {i}")


def test_hip_three_nn_and_knn_against_golden(ops):
    g = _load("ops_nn.npz")
    for i in range(4):
        d, idx = ops.three_nn(dev(g[f"nn{i}_u"]), dev(g[f"nn{i}_k"]))
        np.testing.assert_array_equal(idx.cpu().numpy(), g[f"nn{i}_idx"], err_msg=f"three_nn case {i}")
        ref = np.sqrt(g[f"nn{i}_d2"])  # the operator API returns distances (pointnet2_utils.py:129), the kernel d^2
        got = d.cpu().numpy()
        assert np.array_equal(np.isinf(got), np.isinf(ref))
        np.testing.assert_allclose(np.where(np.isinf(got), 0, got), np.where(np.isinf(ref), 0, ref), rtol=0, atol=ATOL)
    for i in range(6):
        k = g[f"knn{i}_idx"].shape[-1]
        d, idx = ops.knn(k, dev(g[f"knn{i}_u"]), dev(g[f"knn{i}_k"]))
        np.testing.assert_array_equal(idx.cpu().numpy(), g[f"knn{i}_idx"], err_msg=f"knn case {i}")
        ref = np.sqrt(g[f"knn{i}_d2"])
        got = d.cpu().numpy()
        assert np.array_equal(np.isinf(got), np.isinf(ref))
        np.testing.assert_allclose(np.where(np.isinf(got), 0, got), np.where(np.isinf(ref), 0, ref), rtol=0, atol=ATOL)


def test_hip_group_and_interpolate_against_golden(ops):
    g = _load("ops_group_interp.npz")
    for i in range(4):
        f, idx = g[f"grp{i}_f"], g[f"grp{i}_idx"]
        ft = dev(f).requires_grad_(True)
        out = ops.grouping_operation(ft, dev(idx))
        np.testing.assert_array_equal(out.detach().cpu().numpy(), g[f"grp{i}_out"], err_msg=f"group case {i}")  # pure copy
        out.backward(dev(g[f"grp{i}_go"]))
        np.testing.assert_allclose(ft.grad.cpu().numpy(), g[f"grp{i}_gin"], rtol=1e-5, atol=ATOL)
    for i in range(2):
        f, idx, w = g[f"itp{i}_f"], g[f"itp{i}_idx"], g[f"itp{i}_w"]
        ft = dev(f).requires_grad_(True)
        out = ops.three_interpolate(ft, dev(idx), dev(w))
        np.testing.assert_allclose(out.detach().cpu().numpy(), g[f"itp{i}_out"], rtol=0, atol=ATOL)
        out.backward(dev(g[f"itp{i}_go"]))
        np.testing.assert_allclose(ft.grad.cpu().numpy(), g[f"itp{i}_gin"], rtol=1e-5, atol=ATOL)
