"""CPU: the train.py / test.py entry points, Trainer schedule / checkpoint logic, and the tracking loop,
driven on synthetic data with the CPU oracle injected as operator backend (host logic only)."""
import argparse
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "network"))


@pytest.fixture()
def env(tmp_path, monkeypatch):
    monkeypatch.setenv("HOTRACK_DATA_ROOT", str(tmp_path))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)  # force the CPU branch of the config
    from models import pointnet_utils
    from oracle import torch_ops
    pointnet_utils.set_operator_backend(torch_ops)
    yield tmp_path
    pointnet_utils.set_operator_backend(None)


def _args(config, **kw):
    from parse_args import add_args
    p = add_args(argparse.ArgumentParser())
    p.add_argument("--mode_name", default="test")
    a = p.parse_args(["--config", config])
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_config_composition_and_cli_override(env):
    from configs.config import get_config
    a = _args("handtracknet_train_SimGrasp.yml", num_points=1024)
    setattr(a, "network/backbone_out_dim", 96)
    cfg = get_config(a)
    assert cfg["num_points"] == 1024 and cfg["network"]["backbone_out_dim"] == 96
    assert cfg["pointnet"]["camera"]["sa1"]["npoint"] == 256 and cfg["pointnet"]["camera"]["sa2"]["radius_list"] == [0.2]
    assert cfg["batch_size"] == 32 and cfg["learning_rate"] == 1e-4 and cfg["weight_decay"] == 1e-4
    assert os.path.exists(os.path.join(cfg["experiment_dir"], "config.yml"))


def test_train_then_resume_then_track(env):
    import train
    import test as test_entry
    a = _args("handtracknet_train_SimGrasp.yml", batch_size=2, total_epoch=1, synthetic_frames=4, max_iters=2, num_points=256)
    setattr(a, "freq/save", 1)
    train.main(a)
    ckpt = os.path.join(str(env), "exps", "train_debug", "ckpt", "model_0001.pt")
    assert os.path.exists(ckpt)
    state = torch.load(ckpt)
    assert state["epoch"] == 1 and state["iteration"] == 2 and "optimizer" in state
    # resume continues from epoch 1 and runs epoch 2
    a2 = _args("handtracknet_train_SimGrasp.yml", batch_size=2, total_epoch=2, synthetic_frames=4, max_iters=1, num_points=256)
    setattr(a2, "freq/save", 1)
    train.main(a2)
    assert os.path.exists(os.path.join(str(env), "exps", "train_debug", "ckpt", "model_0002.pt"))
    # tracking test over two short synthetic sequences, loading the trained weights with the handnet. prefix
    a3 = _args("handtracknet_test_SimGrasp.yml", experiment_dir="train_debug", synthetic_frames=3, num_points=256)
    test_entry.main(a3)


def test_schedule_matches_reference_rules(env):
    from configs.config import get_config
    from trainer import Trainer
    cfg = get_config(_args("handtracknet_train_SimGrasp.yml", num_points=256), save=False)
    tr = Trainer(cfg)
    # xavier(gain sqrt2) on Conv*/Linear* modules, zero bias; MultiheadAttention.in_proj keeps torch's default
    assert float(tr.model.bhand.conv1.bias.abs().max()) == 0.0
    lrs, moms = [], []
    for _ in range(45):
        tr.step_epoch()
        lrs.append(tr.lr)
        moms.append(tr.model.bhand.bn1.momentum)
    assert abs(lrs[0] - 1e-4) < 1e-12 and abs(lrs[20] - 5e-5) < 1e-12 and abs(lrs[40] - 2.5e-5) < 1e-12  # StepLR(20, 0.5)
    assert moms[0] == 0.1 and moms[19] == 0.05 and moms[39] == 0.025                                      # 0.1 * 0.5^(epoch//20)
    w = tr.loss_weights
    assert w == {"hand_pred_kp_loss": 10, "hand_pred_r_loss": 1, "hand_pred_t_loss": 1}
