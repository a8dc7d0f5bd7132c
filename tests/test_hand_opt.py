"""The hand-pose particle optimiser (network/models/optimization_hand.py) against the IMPORTED reference's
gf_optimize_hand_pose driving the same hand model on a four-frame synthetic sequence (tests/golden/hand_opt_sequence.npz,
made by tests/golden/make_golden_hand.py in the build container): all energy terms on frame 0's first candidate set, then per
frame the optimised keypoints, pose code, rotation and translation -- with frame t fed by frame t-1's result like the tracker
does.  CPU (torch composition of the SDF lookup) in the default suite, the fused HIP lookup under -m gpu."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "network"))
G = os.path.join(ROOT, "tests", "golden")


def _setup(device):
    from models.hand_model import SyntheticLBSHand
    from models.optimization_hand import gf_optimize_hand_pose
    g = np.load(os.path.join(G, "hand_opt_sequence.npz"))
    res, stride = int(g["meta"][0]), float(g["meta"][1])
    cfg = {"device": device, "opt": {"energy_weight": {"penetrate_sum_loss": 1, "sil_loss": 0.1, "attraction_loss": 0.05,
                                                        "vis_regu_loss": 10, "invis_regu_loss": 0, "temporal_smooth": 1}}}
    opt = gf_optimize_hand_pose(cfg, hand_model=SyntheticLBSHand(), particle_size=g["pre_sampled_particle"].shape[0])
    if device == "cpu":  # the product has no CPU lookup: the reference's torch composition is injected (test infrastructure)
        sys.path.insert(0, ROOT)
        from oracle import sdf_torch
        opt.sdf_lookup = sdf_torch.lookup
    opt.pre_sampled_particle = torch.from_numpy(g["pre_sampled_particle"]).to(device)
    opt.load_volume(torch.from_numpy(g["volume"]).reshape(res, res, res), stride)
    proj = dict(zip(("fx", "fy", "cx", "cy", "w", "h"), g["proj"].tolist()))
    obj_pose = {"rotation": torch.from_numpy(g["R_obj"])[None].to(device), "translation": torch.from_numpy(g["t_obj"]).reshape(1, 3, 1).to(device)}
    mask = torch.from_numpy(g["background_mask"]).to(device)
    return g, opt, proj, obj_pose, mask


def _frame_inputs(g, f, device):
    t = lambda k: torch.from_numpy(g[f"f{f}_{k}"]).to(device)
    last = g[f"f{f}_last_kp"]
    return (t("init_mano"), {"rotation": t("init_rot"), "translation": t("init_trans")}, t("init_kp"),
            None if last.size == 0 else torch.from_numpy(last).to(device), t("vis_mask"))


def _run(device, tol_e, tol_kp):
    g, opt, proj, obj_pose, mask = _setup(device)
    with torch.no_grad():
        mano, pose, kp0, last, vis = _frame_inputs(g, 0, device)
        opt.set_init_para(mano, pose, kp0, last, vis, obj_pose, None, proj, mask)
        sp = opt.pre_sampled_particle * opt.initial_scale
        sample = torch.cat([torch.sqrt(1 - sp[:, 0] ** 2 - sp[:, 1] ** 2 - sp[:, 2] ** 2).unsqueeze(1), sp], 1)
        hand, kp = opt.get_kp_from_delta(sample)
        energy = opt.evaluate(hand, kp).float().cpu().numpy()
        np.testing.assert_allclose(energy, g["e0_energy"], rtol=0, atol=tol_e)
        assert (g["e0_penetration"] > 0).all()  # the scenario exercises the penetration and attraction terms
        prev = None
        for f in range(4):
            mano, pose, kp0, _, vis = _frame_inputs(g, f, device)
            final_kp, theta, R, t = opt.optimize(mano, pose, kp0, prev, vis, obj_pose, None, proj, mask)
            np.testing.assert_allclose(final_kp.cpu().numpy(), g[f"f{f}_final_kp"], rtol=0, atol=tol_kp, err_msg=f"frame {f} keypoints")
            np.testing.assert_allclose(theta.cpu().numpy(), g[f"f{f}_theta"], rtol=0, atol=20 * tol_kp, err_msg=f"frame {f} pose code")
            np.testing.assert_allclose(R.cpu().numpy(), g[f"f{f}_R"], rtol=0, atol=5 * tol_kp, err_msg=f"frame {f} rotation")
            np.testing.assert_allclose(t.cpu().numpy(), g[f"f{f}_t"], rtol=0, atol=tol_kp, err_msg=f"frame {f} translation")
            prev = final_kp  # OUR result feeds the next frame (errors would compound)
        # and the optimiser does its job on this scenario: closer to the ground truth than the initial estimate
        err = np.linalg.norm(final_kp.cpu().numpy() - g["f3_gt_kp"], axis=-1).mean()
        assert err < np.linalg.norm(g["f3_init_kp"] - g["f3_gt_kp"], axis=-1).mean()


def test_golden_report_hand():
    rep = json.load(open(os.path.join(G, "GOLDEN_REPORT_HAND.json")))
    assert rep["frames"] == 4 and rep["penetrating_candidates_frame0"] > 0
    assert all(a < b for a, b in zip(rep["mean_kp_error_optimised_m"], rep["mean_kp_error_init_m"]))


def test_product_lookup_has_no_cpu_path():
    """Without the injected test lookup the optimiser's SDF query is the HIP kernel and refuses CPU tensors loudly."""
    g, opt, proj, obj_pose, mask = _setup("cpu")
    opt.sdf_lookup = None
    with torch.no_grad():
        mano, pose, kp0, last, vis = _frame_inputs(g, 0, "cpu")
        opt.set_init_para(mano, pose, kp0, last, vis, obj_pose, None, proj, mask)
        with pytest.raises(RuntimeError, match="CPU|cuda|GPU|device"):
            opt.query_sdf_and_penetration(torch.zeros(2, 778, 3))


def test_hand_optimiser_matches_reference_cpu():
    _run("cpu", tol_e=2e-6, tol_kp=2e-6)


@pytest.mark.gpu
def test_hand_optimiser_matches_reference_gpu():
    """Same sequence on the GPU: candidates through the fused SDF lookup + penetration kernel (csrc/sdf.hip)."""
    _run("cuda", tol_e=1e-5, tol_kp=2e-5)


def test_synthetic_hand_model_is_a_consistent_lbs_hand():
    from models.hand_model import SyntheticLBSHand, rodrigues
    m = SyntheticLBSHand()
    assert m.rest_verts.shape == (778, 3) and m.th_comps.shape == (45, 45) and sorted(m.contact_zones) == [1, 2, 3, 4, 5]
    v0, j0 = m(torch.zeros(3, 48), th_trans=torch.zeros(3, 3))
    assert torch.allclose(v0[0], m.rest_verts, atol=1e-6) and torch.allclose(j0[0], m.rest_joints, atol=1e-6)
    # a global rotation + translation moves everything rigidly
    aa = torch.tensor([[0.3, -0.2, 0.5]])
    tr = torch.tensor([[0.1, 0.2, 0.3]])
    v1, j1 = m(torch.cat([aa, torch.zeros(1, 45)], 1), th_trans=tr)
    R = rodrigues(aa)[0]
    assert torch.allclose(v1[0], m.rest_verts @ R.t() + tr, atol=1e-6) and torch.allclose(j1[0], m.rest_joints @ R.t() + tr, atol=1e-6)
    # bending one joint moves only what hangs below it
    pose = torch.zeros(1, 48)
    pose[0, 3 + 3 * 4 + 0] = 0.8   # articulated joint 4 = index finger, second joint (keypoint 6)
    _, j2 = m(pose, th_trans=torch.zeros(1, 3))
    moved = (j2[0] - m.rest_joints).norm(dim=1) > 1e-6
    assert moved.nonzero().flatten().tolist() == [7, 8]
