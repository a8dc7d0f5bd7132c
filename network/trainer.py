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
from models.track_network import HandTrackModel, ObjTrackModel_Optimization


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
        elif cfg["track"] == "hand_IKNet":
            # reference: HandTrackModel(handnet, IKnet=IKNet) + MANO shape / pose optimisation (trainer.py:120-127).  IKNet and
            # the MANO layer need licensed assets.  With a hand model supplied
            # (cfg['hand_model']: any models/hand_model.HandModel -- a MANO layer, or `--hand_model synthetic`) the entry runs
            # HandTrackNet tracking + the hand-pose particle optimisation (models/optimization_hand.py; IKNet's role is taken
            # by HandTrackModel._pose_init); without one, the HandTrackNet tracking branch of the same loop
            # (track_network.py:214-217) -- and says so.
            hm = cfg.get("hand_model")
            if isinstance(hm, str):
                if hm != "synthetic":
                    raise ValueError("hand_model: 'synthetic' or a models.hand_model.HandModel instance")
                from models.hand_model import SyntheticLBSHand
                hm = SyntheticLBSHand()
                cfg["hand_model"] = hm  # the synthetic sequences pose the same model
            if hm is not None and cfg.get("use_optimization", False):
                self.log_string("track=hand_IKNet: HandTrackNet tracking + hand-pose particle optimisation (hand model: %s; "
                                "IKNet's initial pose from the previous frame + a rigid keypoint fit)" % type(hm).__name__)
                self.model = HandTrackModel(cfg, handnet=HandTrackNet, hand_model=hm)
            else:
                self.log_string("track=hand_IKNet: no hand model (IKNet / MANO assets are not available) -> HandTrackNet tracking branch only")
                self.model = HandTrackModel(dict(cfg, use_optimization=False), handnet=HandTrackNet)
        elif cfg["track"] == "obj_opt":
            self.model = ObjTrackModel_Optimization(cfg)
        elif not cfg["track"]:
            self.model = HandTrackNet(cfg)
            params = [p for p in self.model.parameters() if p.requires_grad]
            # Whole-step HIP graph (forward + loss + backward + Adam in one replay): the training step is ~800 small
            # launches and host-bound (about 14 ms of Python / autograd dispatch against 10.4 ms of kernels), so
            # replaying it removes the host from the loop.  Single process only (DDP's bucketed all-reduce stays eager).
            self.graph_step = bool(cfg.get("graph_step", os.environ.get("HOTRACK_GRAPH_STEP", "0") == "1"))
            self._graph = self._graph_sig = self._static = self._static_loss = None
            # geometry prefetch (graph_step only): the batch-only part of a step (hand frame, sampling, neighbour searches,
            # inverted lists: HandTrackNet.precompute_geometry) as its own HIP graph on its own stream, replayed for batch t+1
            # while batch t's dense step runs -- update(data, next_data=...)
            self.prefetch_geometry = bool(cfg.get("prefetch_geometry", os.environ.get("HOTRACK_PREFETCH_GEOMETRY", "1") == "1"))
            self._geo_graph = self._geo_in = self._geo_pack_g = self._geo_pack_d = self._static_geo = None
            self._geo_stream = self._geo_done = self._geo_ready_for = self._geo_srcs = self._geo_copied = None
            self._geo_slot = 0
            if cfg["optimizer"] == "Adam":
                # GPU: the fused multi-tensor Adam (one or two launches over all parameters instead of ~15 foreach kernels;
                # device-side step counters, so it is graph-capturable as is).  Same update rule: L2 weight decay, no amsgrad.
                on_gpu = isinstance(self.device, torch.device) and self.device.type == "cuda"
                if on_gpu and cfg.get("fused_adam", True) and cfg.get("adam_impl", "hip") == "hip":
                    # hotrack_amd.optim.FusedAdam: the whole update as one stream over all parameters (csrc/adam.hip), torch's
                    # state_dict layout, capture-safe; adam_impl: torch selects torch's own fused multi-tensor kernel
                    from hotrack_amd.optim import FusedAdam
                    self.optimizer = FusedAdam(params, lr=cfg["learning_rate"], betas=(0.9, 0.999), eps=1e-8, weight_decay=cfg["weight_decay"])
                else:
                    kw = dict(fused=True, capturable=self.graph_step) if on_gpu and cfg.get("fused_adam", True) else dict(capturable=self.graph_step)
                    self.optimizer = torch.optim.Adam(params, lr=cfg["learning_rate"], betas=(0.9, 0.999), eps=1e-8,
                                                      weight_decay=cfg["weight_decay"], **kw)
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
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # "ddp": torch DistributedDataParallel (bucketed all-reduce overlapped with backward, eager launches).
        # "flat": ONE all-reduce of a flat gradient buffer between backward and the optimiser -- what lets the step run as
        # captured HIP graphs under data parallelism (graph: forward+backward | eager: RCCL all-reduce of 16.7 MB | graph:
        # Adam); parameters that never receive a gradient keep grad=None on every rank, exactly as with "ddp".
        self.dp_mode = cfg.get("dp", "flat" if getattr(self, "graph_step", False) else "ddp") if self.world > 1 else None
        if (self.world == 1 and dist.is_available() and dist.is_initialized() and self.optimizer is not None
                and str(cfg.get("dp_force", os.environ.get("HOTRACK_DP_FORCE", ""))) == "flat"):
            self.dp_mode = "flat"  # a one-rank process group: the exchange path end to end on one GPU (bench_train.py --dp-selftest)
        # "flat", backward in SEGMENTS (bwd_segments: 2 = everything after the backbone | the backbone): the first segment's
        # gradients (14.5 of the 16.7 MB) are exchanged while the second segment's backward runs, only the last segment's
        # exchange is exposed.  Default 1 since round 6: measured with a one-rank RCCL group on one MI355X
        # (profiles/r06_misc_measurements.md) the single exchange costs +42 us per step (35 of them RCCL's own one-rank kernel),
        # two segments in stream order +95 us (a third graph, the end-of-pass weight-gradient launches once per segment) and
        # with the first exchange on the collective's stream +160 us -- more than the 14.5 MB exchange it would hide is expected
        # to take over xGMI.  dp_overlap: False keeps two segments but issues every exchange in stream order (A/B measurements).
        self.bwd_segments = int(cfg.get("bwd_segments", os.environ.get("HOTRACK_BWD_SEGMENTS", "1")))
        self.dp_overlap = bool(cfg.get("dp_overlap", os.environ.get("HOTRACK_DP_OVERLAP", "1") == "1"))
        self._segs, self._active_segs = {}, []
        self._opt_graph = self._graph_rest = self._comm_stream = self._seg_done = None
        if torch.cuda.is_available():  # gradient homes of an earlier trainer in this process (tests) are not this one's
            from hotrack_amd import train_stack as _ts0
            _ts0.clear_grad_homes()
        if self.world > 1 and self.optimizer is not None:
            if self.dp_mode == "ddp":
                if torch.cuda.is_available():  # DDP's bucket hooks read .grad inside the pass: no deferred weight-gradient sums
                    from hotrack_amd import train_stack as _ts
                    _ts.DEFER_REDUCE = False
                ids = [self.device.index] if isinstance(self.device, torch.device) and self.device.type == "cuda" else None
                self.ddp = nn.parallel.DistributedDataParallel(_StepModule(self.model), device_ids=ids,
                                                               find_unused_parameters=True, broadcast_buffers=False)
            else:  # every rank starts from rank 0's parameters and buffers (what DDP's constructor does)
                with torch.no_grad():
                    for t in list(self.model.parameters()) + [b for b in self.model.buffers() if b.is_floating_point()]:
                        dist.broadcast(t, src=0)

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
        vals = getattr(loss_dict, "fused_values", None)
        if vals is not None and getattr(loss_dict, "fused_total", None) is not None and \
                loss_dict.fused_total_weights is getattr(self, "_wvec", None):
            loss_dict["total_loss"] = loss_dict.fused_total  # formed inside the loss kernel with this trainer's weights
            return loss_dict
        if vals is not None:  # all terms live in one (9,) tensor (hotrack_amd.ext.HandLosses): the weighted total is one multiply + one sum
            from hotrack_amd.ext import HAND_LOSS_NAMES
            key = (vals.device, tuple(sorted(self.loss_weights.items())))
            if getattr(self, "_wvec_key", None) != key:
                self._wvec = torch.tensor([float(self.loss_weights.get(k, 0.0)) for k in HAND_LOSS_NAMES], dtype=vals.dtype, device=vals.device)
                self._wvec_key = key
                net = self.model if isinstance(self.model, HandTrackNet) else getattr(self.model, "handnet", None)
                if net is not None:
                    net.fused_loss_weights = self._wvec  # later steps: the total comes out of the loss kernel itself
            # (not torch.dot: rocBLAS returns its device-pointer result through a memcpy node, which costs ~40 us of idle
            # time inside a replayed graph)
            loss_dict["total_loss"] = (vals * self._wvec).sum()
            return loss_dict
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
        self._graph = None  # learning rate / BN momentum are baked into a captured step: re-capture per epoch

    def resume(self, dataset_len=None):
        ckpt = OrderedDict()
        if self.cfg["track"] == "obj_opt":
            return self.epoch  # nothing to load: the optimiser has no learned parameters (reference trainer.py:128-133)
        if self.cfg["track"] in ("hand", "hand_IKNet"):
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
                tail = getattr(self._bare_model(), "_ftail", None)
                if "tail_dropout_state" in state:  # the fused tail's dropout counter continues where it stopped
                    if not tail:
                        from models.fast_train import FastTail
                        net = self._bare_model()
                        if hasattr(net, "_ftail") and FastTail.supported(net):
                            tail = net._ftail = FastTail(net)
                    if tail:
                        tail.set_dropout_state(state["tail_dropout_state"])
                self.scheduler = self._make_scheduler(last_epoch=self.epoch)
            self.log_string("Resume from epoch %d" % self.epoch)
        self.model.load_state_dict(ckpt, strict=False)
        return self.epoch

    def _bare_model(self):
        """The HandTrackNet inside whatever self.model is (the tracking models wrap it as .handnet)."""
        return getattr(self.model, "handnet", self.model)

    def save(self, name=None):
        if int(os.environ.get("RANK", "0")) != 0:
            return
        name = name or f"model_{self.epoch:04d}"
        path = pjoin(self.ckpt_dir, name + ".pt")
        state = {"epoch": self.epoch, "iteration": self.iteration, "model": self.model.state_dict(),
                 "optimizer": self.optimizer.state_dict()}  # the reference's four keys (trainer.py:253-268)
        tail = getattr(self._bare_model(), "_ftail", None)
        if tail and tail.dropout_state() is not None:
            state["tail_dropout_state"] = tail.dropout_state()
        torch.save(state, path)
        self.log_string(f"Saving model at epoch {self.epoch}, path {path}")

    @staticmethod
    def init_flag_dict():
        return {"track_flag": False, "save_flag": False, "test_flag": False, "IKNet_flag": False}

    def _step(self, data, zero=True):
        if torch.cuda.is_available():
            from hotrack_amd import gemm_tuning
            with gemm_tuning.scope():  # recorded gfx950 GEMM solutions (incl. the split-R weight-gradient GEMMs) for this step
                return self._step_impl(data, zero)
        return self._step_impl(data, zero)

    def _step_impl(self, data, zero=True):
        if self.dp_mode != "flat":
            loss_dict = self._forward_backward(data, zero)
            self.optimizer.step()
            return loss_dict
        # "flat", eager: segment 0's exchange is in flight (async) while segment 1's backward is issued
        loss_dict, cut = self._fb_head(data, zero)
        self._settle_segment(0, cut is not None)
        works = [self._exchange(0, async_op=cut is not None and self.dp_overlap)]
        if cut is not None:
            self._fb_rest(cut)
            self._settle_segment(1, True)
            works.append(self._exchange(1))
        self._finish_exchange(works)
        self.optimizer.step()  # reads the reduced gradients in the flat buffers (every .grad is a view of one)
        return loss_dict

    def _forward_backward(self, data, zero=True, geo=None):
        """Forward, loss and the whole backward (both segments when the model was cut), no gradient exchange."""
        loss_dict, cut = self._fb_head(data, zero, geo)
        self._last_cut = cut is not None
        if cut is not None:
            self._fb_rest(cut)
        return loss_dict

    def _fb_head(self, data, zero=True, geo=None):
        """Forward + loss + the backward down to the backbone cut (the whole backward when the model is not cut: one segment,
        "ddp", single process).  Returns (loss dict, cut | None); cut = (backbone output, its detached leaf) for _fb_rest."""
        if zero:
            self.optimizer.zero_grad()
        flags = self.init_flag_dict()
        if geo is not None:  # this batch's precomputed geometry (static buffers of the captured step)
            data = dict(data, _geometry=geo)
        net = self._bare_model()
        can_cut = hasattr(net, "cut_backbone_grad")
        seg = can_cut and self.dp_mode == "flat" and self.bwd_segments > 1
        # the cut request is module state only for the duration of THIS forward (left set, every later training-mode
        # forward of the model -- a plain loss.backward() outside the Trainer, a deepcopy -- silently detached the backbone)
        cut = None
        try:
            if can_cut:
                net.cut_backbone_grad, net.backward_cut = seg, None
            if self.ddp is not None:
                loss_dict = self.ddp(data, flags)  # forward + compute_loss inside the DDP-wrapped module
            else:
                ret = self.model(data, flags)
                loss_dict, _ = self.model.compute_loss(data, ret, flags)
        finally:
            if can_cut:
                cut, net.backward_cut = (net.backward_cut if seg else None), None
                net.cut_backbone_grad = False
        loss_dict = self.summarize_losses(loss_dict)
        total = loss_dict["total_loss"]
        one = getattr(self, "_unit_grad", None)  # (the root gradient as a kept tensor: backward() alone fills a fresh one every step)
        if one is None or one.device != total.device or one.dtype != total.dtype or one.shape != total.shape:
            capturing = total.is_cuda and torch.cuda.is_current_stream_capturing()
            one = None if capturing else torch.ones_like(total)
            self._unit_grad = one
        total.backward(one)  # "ddp": bucketed all-reduce overlapped with backward
        return loss_dict, cut

    @staticmethod
    def _fb_rest(cut):
        """Second segment: the backbone's backward from the gradient that segment 1 left at the cut."""
        out, leaf = cut
        g, leaf.grad = leaf.grad, None
        if g is not None:
            out.backward(g)

    # ---- "flat" data parallelism: one all-reduce per backward segment, over flat gradient buffers ---------------------------
    def _segment_params(self, s, cut_exists):
        """Parameters whose gradients are complete once segment s has run (in parameter order: the same on every rank)."""
        params = [p for p in self.model.parameters() if p.grad is not None]
        if not cut_exists:
            return params
        up = {id(p) for p in self._bare_model().segment_upstream_parameters()}
        if s == 0 and any(id(p) in up for p in params):
            raise RuntimeError("dp=flat: a parameter upstream of the backward cut received a gradient in the first segment "
                               "(the model's cut does not separate its parameters); use bwd_segments=1")
        return [p for p in params if (id(p) in up) == (s == 1)]

    def _settle_segment(self, s, cut_exists):
        """Segment s's gradients INTO its persistent flat buffer; afterwards every .grad of the segment IS a view of the buffer,
        so the exchange reduces the gradients in place and the optimiser reads the reduced values where they lie: no packing
        concatenation before and no scatter after the exchange (rounds 4-5: two passes over 16.7 MB per step).  The large
        gradients are already there -- their producers write into the registered slice (hotrack_amd.train_stack.grad_buffer:
        the fused stacks' end-of-pass reduction, the grouped weight-gradient launch, the first-layer BatchNorm backward);
        whatever autograd produced elsewhere (LayerNorm / bias / small BatchNorm tensors; everything on the CPU, and on the first
        step, before the slices are registered) is moved by ONE multi-tensor copy.  Capture-safe: the buffers are allocated
        eagerly (Trainer._capture lays them out after its warm-up steps), only the copy is recorded."""
        params = self._segment_params(s, cut_exists)
        if s == 0:
            self._active_segs = []
        self._active_segs.append(s)
        seg = self._segs.get(s)
        key = tuple(id(p) for p in params)
        if seg is None or seg["key"] != key:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("dp=flat: the gradient layout of segment %d changed inside a graph capture" % s)
            seg = self._layout_segment(s, params)
        src, dst = [], []
        for p, v in zip(params, seg["views"]):
            g = p.grad
            if g.data_ptr() != v.data_ptr() or g.shape != v.shape or not g.is_contiguous():
                src.append(g if g.dtype == v.dtype else g.to(v.dtype))
                dst.append(v)
        seg["moved"] = sum(t.numel() for t in dst)
        if dst:
            torch._foreach_copy_(dst, src)
        for p, v in zip(params, seg["views"]):
            p.grad = v

    def _segments_from_grads(self):
        """[parameters of segment 0, (of segment 1)] from the gradients a WHOLE backward left (what _segment_params gives
        segment by segment during a step)."""
        net = self._bare_model()
        params = [p for p in self.model.parameters() if p.grad is not None]
        cut = (hasattr(net, "cut_backbone_grad") and self.dp_mode == "flat" and self.bwd_segments > 1
               and getattr(self, "_last_cut", False))
        if not cut:
            return [params]
        up = {id(p) for p in net.segment_upstream_parameters()}
        return [[p for p in params if id(p) not in up], [p for p in params if id(p) in up]]

    def _layout_segment(self, s, params):
        """A flat buffer for `params`' gradients (parameter order: the same on every rank), every slice 16-byte aligned, and
        the slices registered as the gradients' homes with the kernels that produce them."""
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        dev = params[0].device if params else torch.device(self.device)
        flat = torch.zeros(n, dtype=params[0].dtype if params else torch.float32, device=dev)  # (the pads stay zero)
        views = [flat[o:o + p.numel()].view_as(p) for o, p in zip(offs, params)]
        seg = self._segs[s] = {"flat": flat, "views": views, "params": params, "key": tuple(id(p) for p in params),
                               "count": len(params), "checked": False, "moved": 0}
        if dev.type == "cuda":
            from hotrack_amd import train_stack as _ts
            _ts.set_grad_homes((p, v) for sg in self._segs.values() for p, v in zip(sg["params"], sg["views"]))
        return seg

    def _exchange(self, s, async_op=False):
        """All-reduce (mean) of segment s's flat buffer; returns the work handle (async_op) or None.  Never inside a capture."""
        seg = self._segs[s]
        flat = seg["flat"]
        if not seg["checked"]:
            self._check_flat_layout(flat, seg["count"])
            seg["checked"] = True
        if flat.numel() == 0:
            return None  # nothing received a gradient in this segment on this rank (and, by the layout check, on no rank)
        if dist.get_backend() == "nccl":
            return dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=async_op)  # RCCL over xGMI: ring / tree all-reduce
        w = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)
        seg["divide"] = True
        return w

    def _finish_exchange(self, works):
        """Order the step's stream behind the exchanges (the current stream waits for the collective's stream: the direction
        that costs nothing on this runtime); backends without a mean reduce divide here."""
        for w in works:
            if w is not None:
                w.wait()
        for s in self._active_segs:
            seg = self._segs[s]
            if seg.pop("divide", False):
                seg["flat"].div_(self.world)

    def _check_flat_layout(self, flat, count):
        """The flat exchange assumes every rank holds the same set of non-None gradients (same code path, same shapes: a
        DistributedSampler with drop_last).  Checked whenever a segment's layout is (re)built -- on every rank at the same step
        if the assumption holds; a mismatch raises on all ranks instead of silently misaligning gradients."""
        numel = flat.numel()
        t = torch.tensor([numel, -numel, count, -count], dtype=torch.int64,
                         device=flat.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        mx_n, mn_n, mx_c, mn_c = int(t[0]), -int(t[1]), int(t[2]), -int(t[3])
        if mx_n != mn_n or mx_c != mn_c:
            raise RuntimeError(f"dp=flat: gradient layouts differ across ranks (numel {mn_n}..{mx_n}, tensors {mn_c}..{mx_c}): "
                               "a data-dependent branch produced different parameter uses; use dp=ddp for such models")

    def _agree(self, ok: bool) -> bool:
        """Collective AND over ranks (capture / fallback decisions must be taken by all ranks together: a rank that falls
        back to eager alone would issue a different number of collectives)."""
        if self.world == 1 or self.dp_mode != "flat":
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t[0]))

    @property
    def _flat(self):
        """(compatibility: benches read the exchanged bytes) every segment's flat buffer, in segment order."""
        return [self._segs[s]["flat"] for s in sorted(self._segs)] or None

    def _allreduce_flat(self):
        """All segments' exchanges back to back on the current stream (what a step costs in exchange alone: bench legs)."""
        self._finish_exchange([self._exchange(s) for s in self._active_segs])

    def update(self, data, next_data=None):
        """One training step on `data`.  next_data: the batch the NEXT call will be given (the same object), if the loop knows
        it -- with graph_step its geometry stage is then replayed on a second stream beside this step's dense work."""
        self.model.train()
        loss_dict = None
        if getattr(self, "graph_step", False) and self.ddp is None and torch.cuda.is_available():  # ("ddp" mode stays eager)
            sig = tuple((path, tuple(t.shape), t.dtype) for path, t in self._leaves(data))
            if self._graph is None or sig != self._graph_sig:
                # (re)capture: ranks see same-shaped batches, so all of them are here together; whether the capture worked
                # is agreed collectively, so either every rank replays graphs from now on or every rank stays eager
                err = None
                try:
                    self._capture(data, sig)
                except RuntimeError as exc:  # an op of this configuration cannot be captured
                    err = exc
                    torch.cuda.synchronize()
                if not self._agree(err is None):
                    self.log_string(f"graph_step disabled ({err or 'capture failed on another rank'})")
                    self.graph_step, self._graph, self._opt_graph, self._graph_rest = False, None, None, None
            if self.graph_step:
                loss_dict = dict(self._graphed_step(data, sig, next_data))
        if loss_dict is None:
            loss_dict = self._step(data)
        self.iteration += 1
        loss_dict["learning_rate"] = self.lr
        return loss_dict

    # ---- whole-step HIP graph ---------------------------------------------------------------------------------------
    @staticmethod
    def _leaves(d, prefix=()):
        for k in sorted(d):
            v = d[k]
            if isinstance(v, dict):
                yield from Trainer._leaves(v, prefix + (k,))
            elif torch.is_tensor(v):
                yield prefix + (k,), v

    def _copy_leaves(self, dst_tree, src_tree, more=()):
        """Batch leaves -> static buffers (+ the (dst, src) pairs of `more`).  Device-resident sources go in one multi-tensor
        launch per dtype (the hand-over of a 32 x 1024 batch was five eager copies, ~5 us of launch gap each, in front of every
        replayed step); host sources (a DataLoader batch) are copied leaf by leaf."""
        pairs = [(dst, src) for (_, dst), (_, src) in zip(self._leaves(dst_tree), self._leaves(src_tree))] + list(more)
        fast = [(d, s_) for d, s_ in pairs if s_.is_cuda and s_.device == d.device and s_.dtype == d.dtype and s_.shape == d.shape]
        if len(fast) > 1:
            # ONE launch of our own (pn2x_copy_multi: the table in the kernel arguments): torch's multi-tensor copy needs one dtype
            # (the int32 geometry pack beside the fp32 batch took the per-tensor path: five launches) and takes 17 us for these
            # few hundred KB even then
            contig = [(d, s_) for d, s_ in fast if d.is_contiguous() and s_.is_contiguous()]
            if len(contig) == len(fast):
                from hotrack_amd import ext
                ext.copy_multi([d for d, _ in fast], [s_ for _, s_ in fast])
            else:
                torch._foreach_copy_([d for d, _ in fast], [s_ for _, s_ in fast])
        else:
            fast = []
        done = {id(d) for d, _ in fast}
        for d, s_ in pairs:
            if id(d) not in done:
                d.copy_(s_, non_blocking=True)

    def _geometry_for(self, data, next_data, static=None):
        """Geometry of `data` into the dense step's static buffers; then, if the loop named its next batch, that batch's
        geometry graph on the side stream.

        Synchronisation (measured on this runtime, scripts/probes/two_graph_overlap.py): a wait of the BUSY stream on an event
        of the side stream is free, but an event recorded on the busy stream that another stream waits for stalls the busy
        stream by 50 - 200 us -- whether or not the event completed long ago.  So the dense stream never records an event for the
        side stream: the geometry graph's results are packed (one eager concatenation on the side stream) into one of TWO
        buffers, alternately, and before the side stream is given the buffer of two steps ago the HOST waits for the dense
        stream's copy out of it (an event nobody waits for on the device).  That bounds the host's run-ahead to about two
        steps; it needs 0.3 ms per step against 3.3 ms on the device.  One graph executable, so its replays are strictly
        ordered: a prefetch still in flight is always waited for before the graph or its buffers are touched."""
        cur = torch.cuda.current_stream()
        if self._geo_ready_for is not None:
            cur.wait_event(self._geo_done)
        inline = self._geo_ready_for is None or self._geo_ready_for is not data
        if inline:  # not prefetched: here and now, on this stream
            self._copy_leaves(self._geo_in, data)
            self._geo_graph.replay()
            self._pack_geometry(self._geo_slot)
        slot = self._geo_slot
        # (the pack's copy rides in the batch hand-over's multi-tensor launch when the caller passes its static buffers)
        self._copy_leaves(static if static is not None else {}, data if static is not None else {},
                          more=[(self._geo_pack_d, self._geo_pack_g[slot])])
        self._geo_copied[slot].record(cur)  # (no stream waits for it: the host does, two steps from now)
        self._geo_slot, self._geo_ready_for = 1 - slot, None
        if next_data is not None:
            self._geo_copied[1 - slot].synchronize()  # the copy that last read the other pack (previous step) has finished
            if inline:  # the graph has just been replayed on THIS stream: its next replay must not start beside that one
                self._geo_copied[slot].synchronize()
            with torch.cuda.stream(self._geo_stream):
                self._copy_leaves(self._geo_in, next_data)
                self._geo_graph.replay()
                self._pack_geometry(1 - slot)
                self._geo_done.record(self._geo_stream)
            self._geo_ready_for = next_data

    def _pack_geometry(self, slot):
        """The geometry graph's outputs (static buffers of its private pool) -> pack_g[slot], one concatenation on the current stream."""
        torch.cat(self._geo_srcs, out=self._geo_pack_g[slot])

    def _graphed_step(self, data, sig=None, next_data=None):
        if sig is None:
            sig = tuple((path, tuple(t.shape), t.dtype) for path, t in self._leaves(data))
        if self._graph is None or sig != self._graph_sig:
            self._capture(data, sig)
        if self._geo_graph is not None:
            if next_data is not None and tuple((p_, tuple(t.shape), t.dtype) for p_, t in self._leaves(next_data)) != sig:
                next_data = None  # (a differently shaped batch re-captures everything at its own step)
            self._geometry_for(data, next_data, static=self._static)  # (copies the batch into the static buffers too)
        else:
            self._copy_leaves(self._static, data)
        self._graph.replay()
        if self._opt_graph is not None:
            # data parallel: [forward + backward (gradients land in the flat buffer)] | all-reduce of the buffer, in place | [Adam].
            # bwd_segments = 2: [forward + backward segment 0] | [backward segment 1] | exchanges | [Adam]; with dp_overlap
            # segment 0's exchange runs on the collective's own stream beside graph 2.  The step's stream never records an
            # event that another stream waits for (that stalls it by 50-200 us on this runtime: profiles/r04_two_graph_overlap.txt):
            # the HOST waits for graph 1 (an event nobody waits for on the device) and then issues the collective from an idle
            # stream; the step's stream only ever waits FOR the collective (free).
            works = []
            if self._graph_rest is not None:
                cur = torch.cuda.current_stream()
                if self.dp_overlap:
                    self._seg_done.record(cur)
                self._graph_rest.replay()
                if self.dp_overlap:
                    self._seg_done.synchronize()
                    with torch.cuda.stream(self._comm_stream):
                        works.append(self._exchange(0, async_op=True))
                else:
                    works.append(self._exchange(0))
                works.append(self._exchange(1, async_op=self.dp_overlap))
            else:
                works.append(self._exchange(0))
            self._finish_exchange(works)
            self._opt_graph.replay()
        return self._static_loss

    def _capture(self, data, sig):
        def clone(d):  # static DEVICE buffers: a DataLoader batch arrives on the host, and the captured region must not contain
            # host-to-device copies (pageable-memory copies inside a capture either fail or bake host pointers in)
            return {k: (clone(v) if isinstance(v, dict) else (v.to(self.device, copy=True) if torch.is_tensor(v) else v))
                    for k, v in d.items()}

        self._graph, self._static = None, clone(data)
        pts = data.get("hand_points") if isinstance(data, dict) else None
        if (torch.is_tensor(pts) and pts.dim() == 3 and pts.shape[0] * pts.shape[1] not in tuple(b * 1024 for b in (16, 24, 32, 48, 64))
                and os.environ.get("HOTRACK_TUNE_GEMMS", "0") != "1" and not getattr(Trainer, "_tune_hint_given", False)):
            Trainer._tune_hint_given = True  # once per process
            self.log_string("the shipped GEMM solution table covers per-GPU batches of 16 / 24 / 32 / 48 / 64 x 1024 points; for this batch (%d x %d) "
                            "HOTRACK_TUNE_GEMMS=1 tunes the library GEMMs during this warm-up (~15 s; measured 2.68 -> 2.34 ms/step at "
                            "batch 16, 4.58 -> 3.91 at 48) and HOTRACK_GEMM_CACHE=<file> keeps the result" % (pts.shape[0], pts.shape[1]))
        if self._geo_ready_for is not None:  # a prefetch of the previous capture's graph may still be running: its graph, its
            self._geo_done.synchronize()     # pool and its pack buffers are about to be dropped (host wait: captures are rare)
        self._geo_graph = self._static_geo = self._geo_ready_for = None
        # Warm-up on a side stream (MIOpen / BLAS pick their algorithms, autograd builds its buffers, Adam creates its
        # state) -- then put every value back, IN PLACE, so the captured step starts from the state update() was called
        # with and the optimizer state tensors the graph will update already exist (creating them inside the capture
        # would replay their zero-fill every step).
        model_snap = {k: v.clone() for k, v in self.model.state_dict().items()}
        tail = getattr(self._bare_model(), "_ftail", None)
        seed_snap = tail.seed.clone() if tail and tail.seed is not None else None  # the fused tail's device dropout counter
        opt_snap = {p: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for p, st in self.optimizer.state.items()}
        flat = self.dp_mode == "flat"
        err = None
        # HOTRACK_KEEP_GRAPH=1: keep the hipGraph_t next to its executable (bench legs count its kernel nodes)
        keep = {"keep_graph": True} if os.environ.get("HOTRACK_KEEP_GRAPH", "0") == "1" else {}
        graph = torch.cuda.CUDAGraph(**keep)
        self._opt_graph = self._graph_rest = None
        self._segs, self._active_segs = {}, []  # (laid out again below, from the warm-up's gradients)
        rest = None
        from hotrack_amd import gemm_tuning
        try:
            # The warm-up steps are LOCAL (no gradient exchange: their effect is undone below anyway).  A rank that raises here
            # has therefore issued exactly as many collectives as its peers -- none -- when the ranks agree below (ADVICE r3:
            # with the all-reduce inside the warm-up a failing rank met its peers' all-reduce with the agreement's).
            if self.prefetch_geometry and hasattr(self._bare_model(), "precompute_geometry"):
                try:
                    self._capture_geometry(clone(data))
                except Exception as exc:  # (TypeError of _tree_flatten included) no prefetch: the step runs its geometry in line
                    torch.cuda.synchronize()
                    self._geo_graph = self._static_geo = self._geo_ready_for = None
                    self.log_string(f"geometry prefetch disabled for this batch shape ({exc})")
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            try:
                with torch.cuda.stream(side), gemm_tuning.scope(tune=True):  # (HOTRACK_TUNE_GEMMS=1: this batch shape's GEMMs are tuned here)
                    for _ in range(2):
                        self._forward_backward(self._static, geo=self._static_geo)
                        self.optimizer.step()
                    # (flat) which parameters received a gradient, per backward segment: the flat buffers are laid out from this
                    # BEFORE the capture, so that the captured backward already writes its large gradients into them
                    seg_lists = self._segments_from_grads() if flat else None
            finally:  # whatever happened, the step update() was called for starts from the state it was called with
                torch.cuda.current_stream().wait_stream(side)
                with torch.no_grad():
                    for k, v in self.model.state_dict().items():
                        v.copy_(model_snap[k])
                    for p, st in self.optimizer.state.items():
                        for k, v in st.items():
                            if torch.is_tensor(v):
                                v.copy_(opt_snap[p][k]) if p in opt_snap else v.zero_()
                    # the warm-up forwards advanced the fused tail's dropout counter: a captured / resumed run must draw the
                    # masks an uninterrupted eager run draws; a counter created BY the warm-up goes back to its start value
                    tail = getattr(self._bare_model(), "_ftail", None)
                    if tail:
                        tail.rewind_dropout_counter(seed_snap)
                self.optimizer.zero_grad(set_to_none=True)
            if flat:
                for s_, params_ in enumerate(seg_lists):
                    self._layout_segment(s_, params_)
            with gemm_tuning.scope():
                with torch.cuda.graph(graph):
                    self._static_loss, cut = self._fb_head(self._static, zero=False, geo=self._static_geo)
                    if flat:
                        self._settle_segment(0, cut is not None)
                    else:
                        self.optimizer.step()
                if cut is not None:  # (flat only) the backbone's backward as its own graph: segment 0 travels beside it
                    rest = torch.cuda.CUDAGraph(**keep)
                    with torch.cuda.graph(rest, pool=graph.pool()):
                        self._fb_rest(cut)
                        self._settle_segment(1, True)
                del cut
        except RuntimeError as exc:
            err = exc
            torch.cuda.synchronize()
        if flat:
            # first collective of the capture: every rank arrives here, having exchanged nothing so far
            if not self._agree(err is None):
                raise RuntimeError(f"graph capture failed ({err or 'on another rank'})")
            with gemm_tuning.scope():
                self._allreduce_flat()  # eager (collectives stay outside the graphs); also the collective layout checks
                opt_graph = torch.cuda.CUDAGraph(**keep)
                with torch.cuda.graph(opt_graph, pool=graph.pool()):
                    self.optimizer.step()  # (every .grad is a view of an exchanged flat buffer: nothing to scatter)
            self._opt_graph, self._graph_rest = opt_graph, rest
            if rest is not None and self._comm_stream is None:
                self._comm_stream, self._seg_done = torch.cuda.Stream(), torch.cuda.Event()
        elif err is not None:
            raise err
        self._graph, self._graph_sig = graph, sig

    def _capture_geometry(self, geo_in):
        """The geometry stage of `geo_in`-shaped batches as its own HIP graph: inputs geo_in (static), outputs packed into ONE flat
        buffer (pack_g) so that handing a batch's geometry to the dense step is a single device-to-device copy into pack_d, of
        which self._static_geo holds typed views.  Leaves self._geo_graph None when the model has no such stage here."""
        net, flags = self._bare_model(), self.init_flag_dict()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            geo = None
            for _ in range(2):  # warm-up (allocations, kernel selection) outside the capture
                geo = net.precompute_geometry(geo_in, flags)
        cur.wait_stream(side)
        if geo is None:
            return
        leaves, spec = _tree_flatten(geo)
        if not all(t.dtype in (torch.float32, torch.int32) for t in leaves):
            raise RuntimeError("geometry prefetch: geometry leaves must be 4-byte tensors")
        dev = leaves[0].device
        graph = torch.cuda.CUDAGraph(**({"keep_graph": True} if os.environ.get("HOTRACK_KEEP_GRAPH", "0") == "1" else {}))
        with torch.cuda.graph(graph, stream=side):
            geo = net.precompute_geometry(geo_in, flags)
        # the captured outputs live at fixed addresses of the graph's pool: flat int32 views of them, interleaved with zero pads
        # so that every leaf starts 16-byte aligned in the pack (the kernels of the dense step load rows as float4 / int4)
        outs = [t.contiguous().view(-1).view(torch.int32) for t in _tree_flatten(geo)[0]]
        if not all(o.data_ptr() == t.data_ptr() for o, t in zip(outs, _tree_flatten(geo)[0])):
            raise RuntimeError("geometry prefetch: geometry outputs must be contiguous")
        srcs, offs, total = [], [], 0
        for o in outs:
            offs.append(total)
            srcs.append(o)
            total += o.numel()
            if total % 4:
                srcs.append(torch.zeros(4 - total % 4, dtype=torch.int32, device=dev))
                total += srcs[-1].numel()
        pack_g = [torch.zeros(total, dtype=torch.int32, device=dev) for _ in range(2)]
        pack_d = torch.zeros(total, dtype=torch.int32, device=dev)
        self._static_geo = _tree_unflatten(spec, [pack_d[o:o + t.numel()].view(t.dtype).view(t.shape) for o, t in zip(offs, leaves)])
        self._geo_graph, self._geo_in, self._geo_pack_g, self._geo_pack_d, self._geo_srcs = graph, geo_in, pack_g, pack_d, srcs
        self._geo_stream, self._geo_done, self._geo_slot = side, torch.cuda.Event(), 0
        self._geo_copied = [torch.cuda.Event(), torch.cuda.Event()]
        for e in self._geo_copied:
            e.record(cur)
        # the static batch's own geometry, for the warm-up steps and the capture of the dense step
        cur.wait_stream(side)
        graph.replay()
        self._pack_geometry(0)
        pack_d.copy_(pack_g[0])

    def test(self, data, save_flag=False):
        flags = self.init_flag_dict()
        flags["test_flag"], flags["save_flag"] = True, save_flag
        self.model.eval()
        with torch.no_grad():
            ret = self.model(data, flags)
            loss_dict, ret = self.model.compute_loss(data, ret, flags)
        return loss_dict, ret


def _tree_flatten(obj):
    """(tensor leaves, structure) of nested dicts / lists / tuples / None (dict keys in sorted order)."""
    leaves = []

    def walk(o):
        if torch.is_tensor(o):
            leaves.append(o)
            return ("t",)
        if o is None:
            return ("n",)
        if isinstance(o, dict):
            return ("d", [(k, walk(o[k])) for k in sorted(o)])
        if isinstance(o, (list, tuple)):
            return ("l" if isinstance(o, list) else "u", [walk(v) for v in o])
        raise TypeError(f"unsupported node {type(o)}")
    return leaves, walk(obj)


def _tree_unflatten(spec, leaves):
    it = iter(leaves)

    def build(sp):
        if sp[0] == "t":
            return next(it)
        if sp[0] == "n":
            return None
        if sp[0] == "d":
            return {k: build(v) for k, v in sp[1]}
        seq = [build(v) for v in sp[1]]
        return seq if sp[0] == "l" else tuple(seq)
    return build(spec)


class _StepModule(nn.Module):
    """forward + loss as one module so DDP sees every parameter use of a training step."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, data, flags):
        ret = self.model(data, flags)
        return self.model.compute_loss(data, ret, flags)[0]
