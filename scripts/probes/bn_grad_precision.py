"""Which fp32 implementation of [1x1 conv (3 -> 32) + train-mode BatchNorm + ReLU] backward is closer to fp64?
torch's Conv2d/BatchNorm2d (channel-major) vs hotrack_amd.train_ops (point-major), on sa1-shaped data."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import torch
from hotrack_amd.train_ops import Workspace, bn_relu

torch.manual_seed(0)
B, S, K, Cin, C = 4, 256, 32, 3, 32
rel = (torch.randn(B, S, K, Cin, device="cuda") * 0.08)
W = torch.randn(C, Cin, device="cuda") * 0.8
gamma = 1 + 0.1 * torch.randn(C, device="cuda")
beta = 0.05 * torch.randn(C, device="cuda")
go = torch.randn(B, S, K, C, device="cuda")

def f64():
    w = W.double().requires_grad_(True)
    y = rel.double().view(-1, Cin) @ w.t()
    m, v = y.mean(0), y.var(0, unbiased=False)
    h = torch.relu((y - m) / torch.sqrt(v + 1e-5) * gamma.double() + beta.double())
    h.backward(go.double().view(-1, C))
    return w.grad

def torch32():
    conv = torch.nn.Conv2d(Cin, C, 1).cuda(); bn = torch.nn.BatchNorm2d(C).cuda().train()
    with torch.no_grad():
        conv.weight.copy_(W.view(C, Cin, 1, 1)); conv.bias.zero_(); bn.weight.copy_(gamma); bn.bias.copy_(beta)
    x = rel.permute(0, 3, 1, 2).contiguous()
    h = torch.relu(bn(conv(x)))
    h.backward(go.permute(0, 3, 1, 2).contiguous())
    return conv.weight.grad.view(C, Cin)

def mine():
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta)
    w = W.clone().requires_grad_(True)
    y = torch.nn.functional.linear(rel.view(-1, Cin), w)
    h = bn_relu(y, bn, Workspace("cuda"))
    h.backward(go.view(-1, C))
    return w.grad

t = f64()
for name, g in (("torch fp32 conv+bn", torch32()), ("train_ops", mine())):
    err = (g.double() - t).abs().max().item()
    print(f"{name:22s} max |dW - dW_fp64| = {err:.3e}   (max |dW_fp64| = {t.abs().max().item():.3e})")
