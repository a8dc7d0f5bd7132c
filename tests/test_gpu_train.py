"""GPU: BASELINE.json configs[2] -- one HandTrackNet TRAINING step (train-mode BatchNorm, losses, backward) through the
HIP forward AND backward operators, against the golden vectors of the imported reference's own train step
(tests/golden/handtracknet_reference.npz, make_golden.py:179-246): total loss, pred_kp, the grad-is-None mask
(30 tensors / 3,746,944 parameters never receive a gradient) and the per-parameter gradient norms of the reference's
fp64 re-run.  Also asserts that the C-ABI backward entry points really ran (no torch-indexing substitute)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "network"))
pytestmark = pytest.mark.gpu


def test_train_step_hip_operators_match_reference_golden(monkeypatch):
    import test_network as tn
    from hotrack_amd import pointnet2_hip
    calls = {}
    for name in pointnet2_hip.EXPORTED:
        fn = getattr(pointnet2_hip, name)

        def counted(*a, _fn=fn, _name=name, **k):
            calls[_name] = calls.get(_name, 0) + 1
            return _fn(*a, **k)
        monkeypatch.setattr(pointnet2_hip, name, counted)
    model, ret, total = tn._train_step("cuda", True)
    gold = tn.GOLD
    assert abs(float(total) - float(gold["train_total_loss"])) < 2e-4 * abs(float(gold["train_total_loss"]))
    np.testing.assert_allclose(ret["pred_kp"].detach().cpu().numpy(), gold["train_pred_kp"], atol=5e-4)
    none_mask = np.array([p.grad is None for _, p in model.named_parameters()])
    np.testing.assert_array_equal(none_mask, gold["param_grad_is_none"])
    assert int(none_mask.sum()) == 30
    assert sum(p.numel() for p in model.parameters() if p.grad is None) == 3746944
    gn = np.array([0.0 if p.grad is None else float(p.grad.norm()) for _, p in model.named_parameters()])
    truth = gold["param_grad_norm_f64"]
    live = truth > 1e-6 * truth.max()
    np.testing.assert_allclose(gn[live], truth[live], rtol=1e-2, atol=5e-4)
    # the HIP operators (forward and backward) carried the step
    for name in ("furthest_point_sampling_wrapper", "ball_query_wrapper", "knn_wrapper", "three_nn_wrapper",
                 "three_interpolate_wrapper", "three_interpolate_grad_wrapper", "group_points_wrapper",
                 "group_points_grad_wrapper"):
        assert calls.get(name, 0) > 0, (name, calls)


def test_train_step_is_run_to_run_stable():
    """Two identical steps: same loss to fp32 round-off (the LDS-slab backward kernels use no global atomics)."""
    import test_network as tn
    a = float(tn._train_step("cuda", True)[2])
    b = float(tn._train_step("cuda", True)[2])
    assert abs(a - b) <= 1e-6 * abs(a)
