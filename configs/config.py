"""YAML composition + command-line overrides (counterpart of the reference's configs/config.py:31-99):
all_config/<cfg> -> data_config/<data_config> -> pointnet_config/<pointnet_cfg.*>; any key can be
overridden from the CLI, nested keys with '/' (e.g. --network/backbone_out_dim 384).

Differences: the output root defaults to ./data but is created if missing (the reference asserts it
exists), no MANO directory is required (no MANO layer on this path), and `device` is the local rank's
GPU under torchrun."""
from __future__ import annotations

import os
from os.path import join as pjoin

import torch
import yaml

BASE = os.path.dirname(os.path.abspath(__file__))


def overwrite_config(cfg, key, key_path, value):
    cur = key_path[0]
    if len(key_path) == 1:
        old = cfg.get(cur)
        if old != value:
            print(f"{key} (originally {old}) overwritten by arg {value}")
            cfg[cur] = value
    else:
        overwrite_config(cfg.setdefault(cur, {}), key, key_path[1:], value)


def _load(*parts):
    with open(pjoin(BASE, *parts), "r") as f:
        return yaml.safe_load(f)


def get_config(args, save=True):
    cfg = _load("all_config", args.config)
    cli = dict(vars(args))
    cli.pop("config")
    for key, item in cli.items():
        if item is not None and key not in ("mode_name", "debug", "debug_save", "save", "num_workers", "synthetic_frames", "max_iters"):
            overwrite_config(cfg, key, key.split("/"), item)
    data_cfg = _load("data_config", cfg["data_config"])
    cfg["pointnet"] = {k: _load("pointnet_config", v) for k, v in cfg["pointnet_cfg"].items()}

    root = os.environ.get("HOTRACK_DATA_ROOT", "data")
    cfg["root_dir"] = root
    cfg["save_dir"] = pjoin(root, "exps", cfg.get("save_dir", cfg["experiment_dir"]), "results")
    cfg["experiment_dir"] = pjoin(root, "exps", cfg["experiment_dir"])
    os.makedirs(cfg["save_dir"], exist_ok=True)
    os.makedirs(cfg["experiment_dir"], exist_ok=True)
    if save and int(os.environ.get("RANK", "0")) == 0:
        with open(pjoin(cfg["experiment_dir"], "config.yml"), "w") as f:
            yaml.safe_dump({k: v for k, v in cfg.items()}, f, default_flow_style=False)

    cat = cfg["obj_category"][0] if isinstance(cfg["obj_category"], list) else cfg["obj_category"]
    cfg["num_parts"] = data_cfg[cat]["num_parts"]
    cfg["obj_sym"] = data_cfg[cat]["sym"]
    cfg["data_cfg"] = data_cfg
    cfg["data_cfg"]["basepath"] = pjoin(root, data_cfg["basepath"])
    if torch.cuda.is_available():
        local = int(os.environ.get("LOCAL_RANK", cfg.get("cuda_id", 0)))
        cfg["device"] = torch.device("cuda", local % torch.cuda.device_count())  # (% only matters for the shared-GPU self-test)
    else:
        cfg["device"] = "cpu"
    if cfg.get("hand_model") == "synthetic":  # the same instance poses the synthetic sequences and drives the optimiser
        from models.hand_model import SyntheticLBSHand
        cfg["hand_model"] = SyntheticLBSHand()
    print("Running on ", cfg["device"])
    return cfg
