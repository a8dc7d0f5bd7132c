"""torch.optim.Adam as ONE stream over all parameters (csrc/adam.hip, include/pn2_ext.h: pn2x_adam_multi).

Same update rule as the reference's optimiser (trainer.py:49-52: Adam(lr, betas=(0.9, 0.999), eps, weight_decay) -- L2
decay folded into the gradient, no amsgrad) and the same `state_dict` layout as torch.optim.Adam (`step`, `exp_avg`,
`exp_avg_sq` per parameter), so checkpoints written by either load into the other.  Capture-safe: the step counters are
device tensors, the tensor table travels in the kernel arguments.  Learning rate / betas / eps / weight decay are read from
the parameter group at every step and baked into a captured graph (the Trainer re-captures when the learning rate changes).
Parameters without a gradient are skipped, exactly like torch (the 3.75 M never-used attention parameters of HandTrackNet).
GPU fp32 parameters only."""
from __future__ import annotations

import ctypes

import torch

from . import pointnet2_hip as _native

_lib = _native._lib
_vp = ctypes.c_void_p
_lib.pn2x_adam_multi.argtypes = [ctypes.c_int, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                 ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_long)] + [ctypes.c_double] * 5 + [_vp]
_lib.pn2x_adam_multi.restype = ctypes.c_int
_lib.pn2x_adam_multi2.argtypes = _lib.pn2x_adam_multi.argtypes[:-1] + [ctypes.c_int, _vp]
_lib.pn2x_adam_multi2.restype = ctypes.c_int
_lib.pn2x_adam_advance.argtypes = [_vp, ctypes.c_int, _vp]
_lib.pn2x_adam_advance.restype = ctypes.c_int


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("FusedAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        # the per-parameter `step` tensors are 0-dim views of one buffer per device: one launch advances them all
        self._step_bufs = {}

    def _new_step(self, device):
        n_total = sum(len(g["params"]) for g in self.param_groups)
        buf, used = self._step_bufs.get(device, (None, 0))
        if buf is None or used >= buf.numel():
            buf, used = torch.zeros(max(n_total, 1), dtype=torch.float32, device=device), 0
        self._step_bufs[device] = (buf, used + 1)
        return buf[used]

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = self._new_step(p.device)  # torch's capturable layout: a device scalar (here a view of the shared buffer)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        else:
            stp = st["step"]
            buf = self._step_bufs.get(p.device, (None, 0))[0]
            if (not torch.is_tensor(stp) or not stp.is_cuda or buf is None
                    or stp.untyped_storage().data_ptr() != buf.untyped_storage().data_ptr()):
                # a loaded state dict (torch.optim.Adam's or ours; host counters if it was not capturable): re-home the counter
                new = self._new_step(p.device)
                new.copy_(stp if torch.is_tensor(stp) else torch.tensor(float(stp)))  # (no host read of a device counter)
                st["step"] = new
        return st

    def load_state_dict(self, state_dict):
        """torch's loader, then every step counter is re-homed ONCE, eagerly, into a fresh shared buffer (counters of a loaded
        state -- torch.optim.Adam's or ours -- are separate tensors; re-homing them lazily inside step() could overflow the
        buffer of an optimiser that had already stepped and would then re-home, with a host sync, inside a later step or a
        graph capture)."""
        super().load_state_dict(state_dict)
        self._step_bufs = {}
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st and "step" in st:
                    stp = st["step"]
                    new = self._new_step(p.device)
                    new.copy_(stp if torch.is_tensor(stp) else torch.tensor(float(stp)))
                    st["step"] = new

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        stepped = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            n = len(ps)
            if any(p.device != ps[0].device for p in ps):
                raise RuntimeError("FusedAdam: the parameters of a group must live on one device (one launch, one stream)")
            arr = lambda: (_vp * n)()
            P, G, M, V, S = arr(), arr(), arr(), arr(), arr()
            N = (ctypes.c_long * n)()
            for i, p in enumerate(ps):
                g = p.grad
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or g.dtype != torch.float32 or g.is_sparse:
                    raise RuntimeError("FusedAdam: dense contiguous fp32 GPU parameters only")
                if not g.is_contiguous():
                    g = g.contiguous()
                    p.grad = g
                st = self._init_state(p)
                P[i], G[i], M[i], V[i], S[i], N[i] = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(), p.numel()
            b1, b2 = group["betas"]
            lr = group["lr"]
            lr = float(lr.item()) if torch.is_tensor(lr) else float(lr)
            with torch.cuda.device(ps[0].device):
                _native._check(_lib.pn2x_adam_multi2(n, P, G, M, V, S, N, lr, float(b1), float(b2), float(group["eps"]),
                                                     float(group["weight_decay"]), 0, _native._stream(ps[0])), "adam_multi")
            stepped += [self.state[p]["step"] for p in ps]
        self._advance(stepped)
        return loss

    def _advance(self, steps):
        """step += 1 for every parameter updated in this call: one launch where their counters are exactly the used prefix of the
        shared buffer (the usual case: the same parameters receive gradients every step), else one launch per run of counters."""
        by_dev = {}
        for s in steps:
            by_dev.setdefault(s.device, []).append(s)
        for dev, ss in by_dev.items():
            buf, used = self._step_bufs.get(dev, (None, 0))
            with torch.cuda.device(dev):
                st = _native._stream(ss[0])
                base = None if buf is None else buf.data_ptr()
                ptrs = sorted(s.data_ptr() for s in ss)
                if buf is not None and len(ptrs) == used and ptrs == [base + 4 * i for i in range(used)]:
                    _native._check(_lib.pn2x_adam_advance(base, used, st), "adam_advance")
                else:  # counters from a loaded state dict, or a parameter that got no gradient this time
                    for s in ss:
                        s.add_(1.0)
