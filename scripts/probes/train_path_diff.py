"""Per-parameter gradient difference between the module training path and the point-major one (same weights, same batch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import test_network as tn
from models.hand_network import HandTrackNet
res = {}
for fast in (False, True):
    HandTrackNet._force_fast_train = fast
    model, ret, total = tn._train_step("cuda", True)
    res[fast] = (model, float(total))
(ma, la), (mb, lb) = res[False], res[True]
print("loss", la, lb)
truth = dict(zip(tn.GOLD["param_names"], tn.GOLD["param_grad_norm_f64"]))
pa, pb = dict(ma.named_parameters()), dict(mb.named_parameters())
for k in pa:
    if pa[k].grad is None:
        continue
    a, b = pa[k].grad, pb[k].grad
    ref = float(a.abs().max())
    t = torch.from_numpy(tn.GOLD["g64/" + k]).cuda()
    ea = float((a.flatten()[:1024].double() - t).abs().max()) / max(float(t.abs().max()), 1e-30)
    eb = float((b.flatten()[:1024].double() - t).abs().max()) / max(float(t.abs().max()), 1e-30)
    if ref > 1e-4:
        print(f"{k:42s} max|g64| {float(t.abs().max()):9.3e}  rel err vs fp64: module {ea:8.2e}  point-major {eb:8.2e}   module-vs-pm {float((a-b).abs().max())/ref:8.2e}")
