"""Run the same fused stack forward + backward many times; which result tensors ever deviate from the first run by more than
summation-order noise, and by how much?  (localises the sporadic 0.3 - 7 % gradient errors of test_mlp_stack_matches_torch.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network")]
import torch  # noqa: E402

from hotrack_amd import train_stack as TS  # noqa: E402
from hotrack_amd.train_ops import Workspace  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for R, widths, K in [(4000, [64, 64, 128], 0), (1500, [128, 128, 512], 0), (2048, [128, 128, 384], 0), (21 * 16 * 5, [128, 128, 192], 16)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    y1 = torch.randn(R, widths[0], device="cuda", generator=g) * 1.5 + 0.3
    convs = [torch.nn.Conv1d(a, b, 1).cuda() for a, b in zip(widths[:-1], widths[1:])]
    bns = [torch.nn.BatchNorm1d(c).cuda().train() for c in widths]
    go = None
    first, bad = None, {}
    for it in range(reps):
        for m in convs + bns:
            for p in m.parameters():
                p.grad = None
        ws = Workspace("cuda")
        ya = y1.clone().requires_grad_(True)
        layers = [TS.Layer(None, bns[0], None)] + [TS.Layer(c.weight, bn, c.bias) for c, bn in zip(convs, bns[1:])]
        out = TS.mlp_stack(ya, layers, ws, max_over=K)
        if go is None:
            go = torch.randn(out.shape, device="cuda", generator=g)
        out.backward(go)
        res = {"out": out.detach(), "dy1": ya.grad}
        for i, c in enumerate(convs):
            res[f"dW{i + 2}"] = c.weight.grad
        for i, b in enumerate(bns):
            res[f"dgamma{i + 1}"], res[f"dbeta{i + 1}"] = b.weight.grad, b.bias.grad
        res = {k: v.clone() for k, v in res.items()}
        if first is None:
            first = res
            continue
        for k, v in res.items():
            e = float((v - first[k]).abs().max()) / (float(first[k].abs().max()) + 1e-30)
            if e > 1e-4:
                bad.setdefault(k, []).append(round(e, 5))
    print(R, widths, K, {k: (len(v), max(v)) for k, v in bad.items()} or "all runs agree", flush=True)
