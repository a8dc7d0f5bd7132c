"""Trainer: model + optimiser + schedule + checkpoints (counterpart of the reference's trainer.py:105-331).

Kept from the reference: xavier(gain sqrt 2) init of every module whose class name starts with Conv /
Linear (:20-39), Adam(lr, wd) + StepLR stepped per epoch while lr > lr_clip (:49-52,:173-175), BatchNorm
momentum schedule (:180-190), loss = sum_k w_k loss_k (:157-165), checkpoint dict {epoch, iteration,
model, optimizer} at <exp>/ckpt/model_%04d.pt (:253-268), resume from the newest one, `handnet.`-prefixed
loading for tracking (:206-215).
New: one process per GPU under torchrun -> DistributedDataParallel over RCCL (backend "nccl"), local
BatchNorm, find_unused_parameters=True so the never-used attention parameters keep grad=None exactly as
in the single-GPU reference (3.75 M of 7.92 M parameters)."""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from os.path import join as pjoin

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.init as init
from torch.optim import lr_scheduler

from models.hand_network import HandTrackNet
from models.track_network import HandTrackModel


def weights_init(init_type="gaussian"):
    def init_fun(m):
        name = m.__class__.__name__
        if (name.startswith("Conv") or name.startswith("Linear")) and hasattr(m, "weight"):
            if init_type == "gaussian":
                init.normal_(m.weight.data, 0.0, 0.02)
            elif init_type == "xavier":
                init.xavier_normal_(m.weight.data, gain=math.sqrt(2))
            elif init_type == "kaiming":
                init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
            elif init_type == "orthogonal":
                init.orthogonal_(m.weight.data, gain=math.sqrt(2))
            elif init_type != "default":
                raise ValueError(f"Unsupported initialization: {init_type}")
            if getattr(m, "bias", None) is not None:
                init.constant_(m.bias.data, 0.0)
    return init_fun


def get_last_model(dirname, key=""):
    if not os.path.exists(dirname):
        return None
    models = sorted(pjoin(dirname, f) for f in os.listdir(dirname) if key in f and f.endswith(".pt"))
    return models[-1] if models else None


class Trainer(nn.Module):
    def __init__(self, cfg, logger=None, dataset_len=None):
        super().__init__()
        self.cfg, self.logger = cfg, logger
        self.device = cfg["device"]
        self.ckpt_dir = pjoin(cfg["experiment_dir"], "ckpt")
        os.makedirs(self.ckpt_dir, exist_ok=True)
        self.loss_weights = cfg["network"].get("loss_weight", {})
        if cfg["network"]["type"] != "HandTrackNet":
            raise NotImplementedError("only HandTrackNet is on this path (IKNet needs MANO assets)")
        self.optimizer = self.scheduler = None
        if cfg["track"] == "hand":
            self.model = HandTrackModel(cfg, handnet=HandTrackNet)
        elif not cfg["track"]:
            self.model = HandTrackNet(cfg)
            params = [p for p in self.model.parameters() if p.requires_grad]
            if cfg["optimizer"] == "Adam":
                self.optimizer = torch.optim.Adam(params, lr=cfg["learning_rate"], betas=(0.9, 0.999), eps=1e-8,
                                                  weight_decay=cfg["weight_decay"])
            else:
                self.optimizer = torch.optim.SGD(params, lr=cfg["learning_rate"], momentum=0.9)
            self.scheduler = self._make_scheduler()
        else:
            raise NotImplementedError(cfg["track"])
        self.warm_up = cfg["warm_up"] / 100 * cfg["total_epoch"]
        self.apply(weights_init(cfg["weight_init"]))
        self.epoch = self.iteration = 0
        self.lr = cfg["learning_rate"]
        self.to(self.device)
        # ---- data parallelism: one process per GPU, gradient all-reduce over RCCL / xGMI --------------
        self.ddp = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and self.optimizer is not None:
            ids = [self.device.index] if isinstance(self.device, torch.device) and self.device.type == "cuda" else None
            self.ddp = nn.parallel.DistributedDataParallel(_StepModule(self.model), device_ids=ids,
                                                           find_unused_parameters=True, broadcast_buffers=False)

    def _make_scheduler(self, last_epoch=-1):
        cfg = self.cfg
        if cfg.get("lr_policy", "constant") == "step":
            return lr_scheduler.StepLR(self.optimizer, step_size=cfg["lr_step_size"], gamma=cfg["lr_gamma"], last_epoch=last_epoch)
        return None

    def log_string(self, s):
        if int(os.environ.get("RANK", "0")) == 0:
            print(s)
            if self.logger is not None:
                self.logger.info(s)

    def summarize_losses(self, loss_dict):
        total = 0
        for key, w in self.loss_weights.items():
            if key in loss_dict:
                total = total + loss_dict[key] * w
        loss_dict["total_loss"] = total
        return loss_dict

    def step_epoch(self):
        cfg = self.cfg
        self.epoch += 1
        if self.epoch < self.warm_up:
            self.lr = self.epoch * cfg["learning_rate"] / self.warm_up
        elif self.scheduler is not None:
            if self.scheduler.get_last_lr()[0] > cfg["lr_clip"]:
                self.scheduler.step()
            self.lr = self.scheduler.get_last_lr()[0]
        self.log_string("Epoch %d/%d, learning rate = %f" % (self.epoch, cfg["total_epoch"], self.lr))
        momentum = max(cfg["momentum_original"] * cfg["momentum_decay"] ** (self.epoch // cfg["momentum_step_size"]),
                       cfg["momentum_min"])
        self.log_string("BN momentum updated to %f" % momentum)
        for m in self.model.modules():
            if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                m.momentum = momentum

    def resume(self, dataset_len=None):
        ckpt = OrderedDict()
        if self.cfg["track"] == "hand":
            name = get_last_model(self.ckpt_dir)
            if name is None:
                self.log_string("No HandTrackNet checkpoint found: tracking with freshly initialised weights")
            else:
                sd = torch.load(name, map_location=self.device)["model"]
                ckpt.update({"handnet." + k: v for k, v in sd.items()})
        else:
            e = self.cfg.get("resume_epoch", -1)
            name = pjoin(self.ckpt_dir, f"model_{e:04d}.pt") if e and e > 0 else get_last_model(self.ckpt_dir)
            if name is None or not os.path.exists(name):
                self.log_string("Initialize from 0")
            else:
                state = torch.load(name, map_location=self.device)
                self.epoch, self.iteration = state["epoch"], state["iteration"]
                ckpt.update(state["model"])
                try:
                    self.optimizer.load_state_dict(state["optimizer"])
                except (ValueError, KeyError):
                    pass
                self.scheduler = self._make_scheduler(last_epoch=self.epoch)
            self.log_string("Resume from epoch %d" % self.epoch)
        self.model.load_state_dict(ckpt, strict=False)
        return self.epoch

    def save(self, name=None):
        if int(os.environ.get("RANK", "0")) != 0:
            return
        name = name or f"model_{self.epoch:04d}"
        path = pjoin(self.ckpt_dir, name + ".pt")
        torch.save({"epoch": self.epoch, "iteration": self.iteration, "model": self.model.state_dict(),
                    "optimizer": self.optimizer.state_dict()}, path)
        self.log_string(f"Saving model at epoch {self.epoch}, path {path}")

    @staticmethod
    def init_flag_dict():
        return {"track_flag": False, "save_flag": False, "test_flag": False, "IKNet_flag": False}

    def update(self, data):
        self.model.train()
        self.optimizer.zero_grad()
        flags = self.init_flag_dict()
        if self.ddp is not None:
            loss_dict = self.ddp(data, flags)  # forward + compute_loss inside the DDP-wrapped module
        else:
            ret = self.model(data, flags)
            loss_dict, _ = self.model.compute_loss(data, ret, flags)
        loss_dict = self.summarize_losses(loss_dict)
        loss_dict["total_loss"].backward()  # DDP: bucketed all-reduce overlapped with backward
        self.optimizer.step()
        self.iteration += 1
        loss_dict["learning_rate"] = self.lr
        return loss_dict

    def test(self, data, save_flag=False):
        flags = self.init_flag_dict()
        flags["test_flag"], flags["save_flag"] = True, save_flag
        self.model.eval()
        with torch.no_grad():
            ret = self.model(data, flags)
            loss_dict, ret = self.model.compute_loss(data, ret, flags)
        return loss_dict, ret


class _StepModule(nn.Module):
    """forward + loss as one module so DDP sees every parameter use of a training step."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, data, flags):
        ret = self.model(data, flags)
        return self.model.compute_loss(data, ret, flags)[0]
