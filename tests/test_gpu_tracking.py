"""GPU: the tracking loop (network/test.py path): HIP-graph replay per frame == eager per frame, and the
test.py entry point runs end to end on synthetic sequences."""
import argparse
import copy
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "network"))
pytestmark = pytest.mark.gpu


def _cfg():
    from _netinit import make_cfg
    cfg = make_cfg(torch.device("cuda", 0))
    cfg.update(num_points=1024, hand_jitter_cfg={"rand_scale": 0.01}, track="hand", use_optimization=False)
    return cfg


def test_graph_tracking_equals_eager_tracking():
    from _netinit import deterministic_init
    from datasets.synthetic import SyntheticSequences
    from hotrack_amd import fused, pointnet2_utils
    from models import pointnet_utils
    from models.hand_network import HandTrackNet
    from models.track_network import HandTrackModel
    pointnet_utils.set_operator_backend(pointnet2_utils)
    cfg = _cfg()
    model = HandTrackModel(cfg, handnet=HandTrackNet)
    deterministic_init(model)
    model = model.cuda().eval()
    seq = SyntheticSequences(cfg, 1, 6)[0]
    flags = {"track_flag": True, "test_flag": True, "save_flag": False, "IKNet_flag": False}
    try:
        pointnet_utils.set_fused_backend(fused)
        with torch.no_grad():
            model.use_graph = True
            a = model(copy.deepcopy(seq), dict(flags))
            assert len(model._graphs) == 1
            b2 = model(copy.deepcopy(seq), dict(flags))  # second sequence reuses the captured graph
            model.use_graph = False
            b = model(copy.deepcopy(seq), dict(flags))
    finally:
        pointnet_utils.set_fused_backend(None)
    for ra, rb, rc in zip(a, b, b2):
        assert torch.equal(ra["pred_kp"], rc["pred_kp"])
        assert torch.allclose(ra["pred_kp"], rb["pred_kp"], atol=1e-5)
    loss, _ = model.compute_loss(seq, a, dict(flags))
    assert all(v == v for v in loss.values())  # finite


def test_test_py_entry_point_on_gpu(tmp_path, monkeypatch):
    monkeypatch.setenv("HOTRACK_DATA_ROOT", str(tmp_path))
    import test as test_entry
    from parse_args import add_args
    p = add_args(argparse.ArgumentParser())
    p.add_argument("--mode_name", default="test")
    a = p.parse_args(["--config", "handtracknet_test_SimGrasp.yml"])
    a.num_points, a.synthetic_frames = 1024, 5
    test_entry.main(a)


def test_loader_side_fps_batched_equals_per_cloud(oracle):
    import numpy as np
    from datasets.data_utils import farthest_point_sample_batch
    rng = np.random.default_rng(0)
    clouds = [rng.random((n, 3)).astype(np.float32) for n in (6000, 9000, 2000, 5120, 700)]
    npoint = 512

    class Fixed:  # deterministic "random" permutation so both calls see the same pre-subsample
        def __init__(self):
            self.r = np.random.default_rng(1)

        def permutation(self, n):
            return self.r.permutation(n)
    batched = farthest_point_sample_batch(clouds, npoint, "cuda", rng=Fixed())
    f = Fixed()
    single = [farthest_point_sample_batch([c], npoint, "cuda", rng=f)[0] for c in clouds]
    for b, s, c in zip(batched, single, clouds):
        assert len(b) == npoint and len(set(b.tolist())) == npoint and b.max() < len(c)
        np.testing.assert_array_equal(b, s)
    # the sampled subset is exactly the oracle's FPS of the pre-subsampled cloud
    f = Fixed()
    keep = f.permutation(6000)[:5 * npoint]
    ref = oracle.furthest_point_sample(clouds[0][keep][None], npoint)[0]
    np.testing.assert_array_equal(batched[0], keep[ref])
