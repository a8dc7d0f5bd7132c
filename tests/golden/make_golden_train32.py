"""Golden vectors of BASELINE configs[2] at its PER-GPU size: the imported reference HandTrackNet's training step over 32
clouds x 1024 points, re-run in fp64 (ground truth for the two fp32 training paths of this repo, which disagree with each other
at the 1e-3 level through the train-mode BatchNorm chains).  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference); same
harness as make_golden.py (operator functions patched to the CUDA semantics via the oracle, deterministic name-keyed weights,
seeded inputs, dropout off).  Writes tests/golden/handtracknet_train32_f64.npz (~0.3 MB)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden import O, torch_ops, deterministic_init, make_cfg, synthetic_frames  # noqa: E402

B, N, SEED = 32, 1024, 3000


def main():
    assert os.path.isdir(mg.REF), "golden vectors can only be regenerated where /root/reference exists"
    O.build()
    ref_pu, ref_hn = mg.import_reference()
    ref_pu.CUDA = True

    class _F64Ops:  # as in make_golden.py: index operators see fp32 coordinates, gathers are dtype-agnostic torch indexing
        furthest_point_sample = staticmethod(lambda xyz, n: torch_ops.furthest_point_sample(xyz.float(), n))
        ball_query = staticmethod(lambda r, k, xyz, new: torch_ops.ball_query(r, k, xyz.float(), new.float()))
        knn = staticmethod(lambda k, u, kn: tuple(t if i else t.double() for i, t in enumerate(torch_ops.knn(k, u.float(), kn.float()))))

        @staticmethod
        def three_nn(u, kn):
            _, idx = torch_ops.three_nn(u.float(), kn.float())
            d = (u.unsqueeze(2) - torch.gather(kn.unsqueeze(1).expand(-1, u.shape[1], -1, -1), 2,
                                               idx.long().unsqueeze(-1).expand(-1, -1, -1, 3))).norm(dim=-1)
            return d, idx

        @staticmethod
        def three_interpolate(points, idx, weight):
            Bq, Nq = idx.shape[:2]
            g = ref_pu.index_points(points.permute(0, 2, 1), idx.long())
            return (g * weight.view(Bq, Nq, 3, 1)).sum(dim=2).permute(0, 2, 1)

    ref_pu.futils = _F64Ops
    cfg = make_cfg("cpu")
    torch.manual_seed(0)
    model = ref_hn.HandTrackNet(cfg)
    deterministic_init(model)
    model = model.double().train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    flags = {"track_flag": False, "test_flag": False, "save_flag": False, "IKNet_flag": False}
    to64 = lambda d: {k: (v.double() if torch.is_tensor(v) else to64(v)) for k, v in d.items()}
    data = to64(synthetic_frames(SEED, B, N))
    _float = torch.Tensor.float
    torch.Tensor.float = lambda self: self  # hand_network.py casts its inputs with .float(); keep fp64 for this run
    torch.set_default_dtype(torch.float64)
    try:
        ret = model(data, dict(flags))
        loss, ret = model.compute_loss(data, ret, dict(flags))
    finally:
        torch.Tensor.float = _float
        torch.set_default_dtype(torch.float32)
    total = 10 * loss["hand_pred_kp_loss"] + loss["hand_pred_r_loss"] + loss["hand_pred_t_loss"]
    total.backward()
    out = {"meta": np.array([B, N, SEED]), "train_total_loss_f64": np.array(float(total)),
           "train_pred_kp_f64": ret["pred_kp"].detach().numpy(),
           "param_names": np.array([n for n, _ in model.named_parameters()]),
           "param_grad_is_none": np.array([p.grad is None for _, p in model.named_parameters()]),
           "param_grad_norm_f64": np.array([0.0 if p.grad is None else float(p.grad.norm()) for _, p in model.named_parameters()])}
    for n, p in model.named_parameters():
        if p.grad is not None:
            out["g64/" + n] = p.grad.flatten()[:256].numpy().astype(np.float64)
    for n, b in model.named_buffers():  # BatchNorm running statistics after the step
        if n.endswith("running_mean") or n.endswith("running_var"):
            out["buf/" + n] = b.numpy().astype(np.float64)
    np.savez_compressed(os.path.join(HERE, "handtracknet_train32_f64.npz"), **out)
    print("train32 golden: loss %.9f, %d tensors without gradient (%d parameters)" % (
        float(total), int(out["param_grad_is_none"].sum()),
        sum(p.numel() for _, p in model.named_parameters() if p.grad is None)))


if __name__ == "__main__":
    main()
