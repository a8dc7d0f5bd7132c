"""Eval-time fused set-abstraction layers (enabled with pointnet_utils.set_fused_backend(fused)).

`sa_scale` replaces, for one scale of an SA-MSG / GivenCenterPoints module in eval mode,
    group(points) | group(xyz) - centre | cat | [Conv2d 1x1 + BN + ReLU] x3 | max over K
(reference pointnet_utils.py:389-403, :566-581) by
    two small dense GEMMs (the per-point and per-centroid halves of the linear first layer)
    + ONE hand-written MFMA kernel (pn2x_sa_mlp_max, csrc/sa_fused.hip)
so the (B, C, S, K) grouped tensors are never written to HBM.  BatchNorm (running statistics)
is folded into the convolutions; folded weights are cached per module and refreshed when the
parameters or buffers change (version counters).  Shapes the kernel does not cover fall back
to the unfused OPERATOR path (still the HIP kernels -- never a CPU path).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ext


def fold_conv_bn(conv: torch.nn.Module, bn: torch.nn.Module):
    """(W', b') with  bn(conv(x)) == W' x + b'  for eval-mode BatchNorm; W' is (Cout, Cin)."""
    W = conv.weight.detach().reshape(conv.weight.shape[0], -1)
    b = conv.bias.detach() if conv.bias is not None else W.new_zeros(W.shape[0])
    s = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
    return (W * s[:, None]).contiguous(), ((b - bn.running_mean) * s + bn.bias.detach()).contiguous()


def _versions(convs, bns):
    v = []
    for c, n in zip(convs, bns):
        v += [c.weight._version, c.bias._version, n.weight._version, n.bias._version, n.running_mean._version,
              n.running_var._version, c.weight.data_ptr()]
    return tuple(v)


def _folded(convs, bns):
    key = _versions(convs, bns)
    cache = getattr(convs, "_pn2_folded", None)
    if cache is None or cache[0] != key:
        cache = (key, [fold_conv_bn(c, n) for c, n in zip(convs, bns)])
        convs._pn2_folded = cache
    return cache[1]


def supported(convs, K: int) -> bool:
    if len(convs) != 3:
        return False
    c1, c2, c3 = (c.weight.shape[0] for c in convs)
    return ext.sa_mlp_max_supported(int(K), c1, c2, c3)


def sa_scale(convs, bns, xyz: torch.Tensor, points: Optional[torch.Tensor], new_xyz: torch.Tensor,
             group_idx: torch.Tensor, center_feat: Optional[torch.Tensor] = None) -> torch.Tensor:
    """xyz (B,3,N), points (B,D,N)|None, new_xyz (B,3,S), group_idx (B,S,K) int32,
    center_feat (B,D2,S)|None -> (B,C3,S).  Input channel order [points | xyz - centre | centre feat]."""
    K = group_idx.shape[-1]
    if not supported(convs, K):
        return None
    (W1, b1), (W2, b2), (W3, b3) = _folded(convs, bns)
    D = 0 if points is None else points.shape[1]
    B = xyz.shape[0]
    parts = getattr(convs, "_pn2_w1_parts", None)
    if parts is None or parts[0] is not W1:
        parts = (W1, W1[:, :D].t().contiguous(), W1[:, D:D + 3].contiguous(), W1[:, D + 3:].t().contiguous())
        convs._pn2_w1_parts = parts
    _, W1f_t, Wx, Wc_t = parts
    # per-point feature half of layer 1 as ONE dense GEMM over the N points: (B,N,C1) = points^T W1[:, :D]^T
    a1f = torch.matmul(points.transpose(1, 2), W1f_t) if D else None
    # per-centroid half W1[:, centre] . centre_feat_s; the [xyz_j - c_s] term and the bias are added in-kernel
    cadd = torch.matmul(center_feat.transpose(1, 2), Wc_t) if center_feat is not None else None
    return ext.sa_mlp_max(group_idx.contiguous(), W2, b2, W3, b3, a1f=a1f, xyz=xyz.transpose(1, 2).contiguous(),
                          cxyz=new_xyz.transpose(1, 2).contiguous(), wx=Wx, b1=b1, cadd=cadd)


def mlp_stack(x: torch.Tensor, convs, bns) -> torch.Tensor:
    """Eval-mode [Conv 1x1 + BN + ReLU]* on (B,C,N) (or (B,C,N,1)) activations: BatchNorm folded into
    the weights, GEMM by the BLAS library, bias + ReLU in one in-place streaming kernel.
    (reference: the Conv1d/Conv2d + BatchNorm + ReLU stacks of pointnet_utils.py:460-462, :504-506)."""
    shape4 = x.shape if x.dim() == 4 else None
    if shape4 is not None:
        x = x.reshape(shape4[0], shape4[1], shape4[2] * shape4[3])
    for W, b in _folded(convs, bns):
        x = ext.bias_act_(torch.matmul(W, x), b, relu=True)
    return x.view(shape4[0], -1, shape4[2], shape4[3]) if shape4 is not None else x


def conv_bn_relu(x: torch.Tensor, conv, bn) -> torch.Tensor:
    """Single eval-mode Conv1d 1x1 + BatchNorm1d + ReLU (backbones.py:131 conv1/bn1)."""
    key = _versions([conv], [bn])
    cache = getattr(conv, "_pn2_folded", None)
    if cache is None or cache[0] != key:
        cache = (key, fold_conv_bn(conv, bn))
        conv._pn2_folded = cache
    W, b = cache[1]
    return ext.bias_act_(torch.matmul(W, x), b, relu=True)
