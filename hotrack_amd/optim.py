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


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("FusedAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)  # torch's capturable layout: a device scalar
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        elif not st["step"].is_cuda:  # a state dict written by a non-capturable torch.optim.Adam (host counters)
            st["step"] = st["step"].to(device=p.device, dtype=torch.float32)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            n = len(ps)
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
                _native._check(_lib.pn2x_adam_multi(n, P, G, M, V, S, N, lr, float(b1), float(b2), float(group["eps"]),
                                                    float(group["weight_decay"]), _native._stream(ps[0])), "adam_multi")
        return loss
