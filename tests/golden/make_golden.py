"""Generates the committed golden vectors.  RUNS ONLY IN THE BUILD CONTAINER (needs
/root/reference); the GPU box and the tests only read the .npz files it writes.

  ops_*.npz      per-operator known-answer vectors: seeded inputs + oracle outputs, each
                 cross-checked here against the IMPORTED reference Python fallback
                 (network/models/pointnet_utils.py, CUDA=False branch) wherever the fallback's
                 semantics coincide with the CUDA kernels' (SURVEY.md 8(c)).
  handtracknet_*.npz  network-level vectors: the imported reference HandTrackNet (unmodified
                 source) run on CPU with its operator functions patched, harness-side, to the
                 CUDA semantics (oracle), deterministic name-keyed weights, seeded inputs.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"

from oracle import pn2_oracle as O  # noqa: E402
from oracle import torch_ops  # noqa: E402
from _cases import cloud, take_points  # noqa: E402
from _netinit import deterministic_init, make_cfg, synthetic_frames  # noqa: E402


def import_reference():
    class _Stub(types.ModuleType):  # harness-side stand-ins for modules only the MANO layer needs
        def __getattr__(self, item):
            if item.startswith("__"):
                raise AttributeError(item)
            return type(item, (), {})

    for name in ("chumpy", "cv2"):
        sys.modules.setdefault(name, _Stub(name))
    sys.path[:0] = [REF, os.path.join(REF, "network"), os.path.join(REF, "network", "models")]
    torch.Tensor.cuda = lambda self, *a, **k: self  # transformer.py:110 hard-codes .cuda()
    import pointnet_utils as ref_pu
    import hand_network as ref_hn
    return ref_pu, ref_hn


def gen_ops(ref_pu):
    torch.manual_seed(0)
    report = {}
    # ---- FPS: reference fallback draws a random start; force it to 0 (the CUDA start) ----------
    cases = {}
    agree = total = 0
    for i, (B, N, M, kind) in enumerate([(2, 1024, 256, "uniform"), (2, 256, 128, "hand"), (2, 1000, 100, "uniform"),
                                         (1, 2560, 512, "uniform"), (2, 1024, 128, "lattice"), (2, 343, 343, "lattice"),
                                         (2, 512, 40, "dup"), (2, 21, 8, "uniform"), (1, 5120, 256, "uniform"), (2, 3, 3, "uniform")]):
        xyz = cloud(100 + i, B, N, kind)
        idx = O.furthest_point_sample(xyz, M)
        assert np.array_equal(idx, O.furthest_point_sample(xyz, M, keyed=True))
        cases[f"fps{i}_xyz"], cases[f"fps{i}_idx"] = xyz, idx
        if kind in ("uniform", "hand"):  # tie-free clouds: the fallback (start forced to 0) must agree
            orig = torch.randint
            torch.randint = lambda lo, hi, size, **kw: torch.zeros(size, dtype=torch.long)
            try:
                ref = ref_pu.farthest_point_sample(torch.from_numpy(xyz), M).numpy()
            finally:
                torch.randint = orig
            total += ref.size
            agree += int((ref == idx).sum())
    report["fps_vs_reference_fallback"] = (agree, total)
    np.savez_compressed(os.path.join(HERE, "ops_fps.npz"), **cases)

    # ---- ball query ------------------------------------------------------------------------
    cases = {}
    agree = total = 0
    for i, (B, N, S, r, K, kind) in enumerate([(2, 1024, 256, 0.1, 32, "hand"), (2, 256, 128, 0.2, 32, "hand"),
                                               (2, 1024, 64, 0.1, 32, "uniform"), (2, 500, 50, 0.3, 64, "uniform"),
                                               (2, 300, 20, 0.25, 8, "lattice"), (1, 2000, 100, 0.05, 1, "uniform")]):
        xyz = cloud(200 + i, B, N, kind)
        new = take_points(xyz, O.furthest_point_sample(xyz, S))
        if kind == "lattice":
            new = new + np.float32(0.25)
        idx = O.ball_query(r, K, xyz, new)
        cases[f"bq{i}_xyz"], cases[f"bq{i}_new"], cases[f"bq{i}_idx"] = xyz, new, idx
        cases[f"bq{i}_rk"] = np.array([r, K], dtype=np.float64)
        if kind != "lattice":  # away from the exact boundary the fallback (> r^2, matmul distances) agrees
            ref = ref_pu.query_ball_point(r, K, torch.from_numpy(xyz), torch.from_numpy(new)).numpy()
            total += ref.shape[0] * ref.shape[1]
            agree += int((ref == idx).all(-1).sum())
    report["ball_rows_vs_reference_fallback"] = (agree, total)
    np.savez_compressed(os.path.join(HERE, "ops_ball_query.npz"), **cases)

    # ---- three_nn / knn ------------------------------------------------------------------------
    cases = {}
    agree = total = 0
    for i, (B, n, m, kind) in enumerate([(2, 256, 128, "hand"), (2, 1024, 256, "uniform"), (2, 40, 2, "uniform"),
                                         (2, 100, 64, "lattice")]):
        u, k = cloud(300 + i, B, n, kind), cloud(350 + i, B, m, kind)
        d2, idx = O.three_nn(u, k)
        cases[f"nn{i}_u"], cases[f"nn{i}_k"], cases[f"nn{i}_d2"], cases[f"nn{i}_idx"] = u, k, d2, idx
        if kind != "lattice" and m >= 3:
            rd, ri = ref_pu.three_nn(torch.from_numpy(u), torch.from_numpy(k))  # fallback: squared distances
            total += ri.numel()
            agree += int((ri.numpy() == idx).sum())
            assert np.allclose(rd.numpy(), d2, atol=1e-5)
    report["three_nn_idx_vs_reference_fallback"] = (agree, total)
    agree = total = 0
    for i, (B, n, m, kk, kind) in enumerate([(2, 21, 1024, 16, "hand"), (2, 21, 1024, 64, "uniform"), (2, 21, 1024, 4, "uniform"),
                                             (1, 10, 300, 200, "uniform"), (2, 9, 5, 8, "uniform"), (2, 16, 200, 32, "lattice")]):
        u, k = cloud(400 + i, B, n, kind), cloud(450 + i, B, m, kind)
        d2, idx = O.knn(kk, u, k)
        cases[f"knn{i}_u"], cases[f"knn{i}_k"], cases[f"knn{i}_d2"], cases[f"knn{i}_idx"] = u, k, d2, idx
        if kind != "lattice" and m >= kk:
            rd, ri = ref_pu.knn_point(kk, torch.from_numpy(u), torch.from_numpy(k))  # topk: tie order unspecified
            total += ri.numel()
            agree += int((ri.numpy() == idx).sum())
            assert np.allclose(rd.numpy(), np.sqrt(d2), atol=1e-5)
    report["knn_idx_vs_reference_fallback"] = (agree, total)
    np.savez_compressed(os.path.join(HERE, "ops_nn.npz"), **cases)

    # ---- group / gather / interpolate (the fallback coincides exactly given idx / weights) --------
    cases = {}
    rng = np.random.default_rng(5)
    for i, (B, C, N, P, S) in enumerate([(2, 3, 1024, 64, 32), (2, 64, 256, 32, 32), (2, 40, 100, 21, 16), (1, 0, 50, 5, 4)]):
        f = rng.normal(size=(B, C, N)).astype(np.float32)
        idx = rng.integers(0, N, (B, P, S)).astype(np.int32)
        out = O.group_points(f, idx)
        if C:
            ref = ref_pu.group_operation(torch.from_numpy(f), torch.from_numpy(idx).long()).numpy()
            assert np.array_equal(ref, out)
        go = rng.normal(size=(B, C, P, S)).astype(np.float32)
        cases[f"grp{i}_f"], cases[f"grp{i}_idx"], cases[f"grp{i}_out"] = f, idx, out
        cases[f"grp{i}_go"], cases[f"grp{i}_gin"] = go, O.group_points_grad(go, idx, N)
    for i, (B, C, M, n) in enumerate([(2, 32, 128, 256), (2, 16, 64, 100)]):
        f = rng.normal(size=(B, C, M)).astype(np.float32)
        idx = rng.integers(0, M, (B, n, 3)).astype(np.int32)
        w = rng.random((B, n, 3)).astype(np.float32)
        w /= w.sum(-1, keepdims=True)
        out = O.three_interpolate(f, idx, w)
        ref = ref_pu.three_interpolate(torch.from_numpy(f), torch.from_numpy(idx).long(), torch.from_numpy(w)).numpy()
        assert np.allclose(ref, out, atol=1e-5)
        go = rng.normal(size=(B, C, n)).astype(np.float32)
        cases[f"itp{i}_f"], cases[f"itp{i}_idx"], cases[f"itp{i}_w"], cases[f"itp{i}_out"] = f, idx, w, out
        cases[f"itp{i}_go"], cases[f"itp{i}_gin"] = go, O.three_interpolate_grad(go, idx, w, M)
    np.savez_compressed(os.path.join(HERE, "ops_group_interp.npz"), **cases)
    return report


def gen_network(ref_pu, ref_hn):
    """Reference HandTrackNet, unmodified source, operators patched to the CUDA semantics."""
    ref_pu.CUDA = True
    ref_pu.futils = torch_ops  # what `from pointnet_lib import pointnet2_utils as futils` would be on a CUDA box
    cfg = make_cfg("cpu")
    torch.manual_seed(0)
    model = ref_hn.HandTrackNet(cfg)
    deterministic_init(model)
    flags = {"track_flag": False, "test_flag": True, "save_flag": False, "IKNet_flag": False}
    out = {}
    # eval forward
    model.eval()
    data = synthetic_frames(1000, 2, 1024)
    with torch.no_grad():
        ret = model(data, dict(flags))
        loss, ret = model.compute_loss(data, ret, dict(flags))
    for k in ("hand_points", "jittered_hand_kp", "gt_hand_kp"):
        out[f"in_{k}"] = data[k].numpy()
    out["in_palm_template"] = data["gt_hand_pose"]["palm_template"].numpy()
    out["eval_pred_kp"] = ret["pred_kp"].numpy()
    out["eval_pred_kp_handframe"] = ret["pred_kp_handframe"].numpy()
    out["eval_rotation"] = ret["canon_pose"]["rotation"].numpy()
    out["eval_translation"] = ret["canon_pose"]["translation"].numpy()
    for k, v in loss.items():
        out[f"eval_loss_{k}"] = np.array(float(v))
    # backbone feature checksum (per-channel mean) for a mid-network check
    with torch.no_grad():
        feat = model.bhand(ret["points_handframe"])
    out["eval_backbone_mean"] = feat.mean(dim=(0, 2)).numpy()
    out["eval_backbone_absmax"] = feat.abs().amax(dim=(0, 2)).numpy()

    # train-mode step (BN batch statistics; dropout disabled so the result is RNG-free)
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    data = synthetic_frames(2000, 4, 1024)
    ret = model(data, dict(flags, test_flag=False))
    loss, ret = model.compute_loss(data, ret, dict(flags, test_flag=False))
    total = 10 * loss["hand_pred_kp_loss"] + loss["hand_pred_r_loss"] + loss["hand_pred_t_loss"]
    total.backward()
    out["train_total_loss"] = np.array(float(total))
    out["train_pred_kp"] = ret["pred_kp"].detach().numpy()
    names = [n for n, _ in model.named_parameters()]
    out["param_names"] = np.array(names)
    out["param_grad_is_none"] = np.array([p.grad is None for _, p in model.named_parameters()])
    out["param_grad_norm"] = np.array([0.0 if p.grad is None else float(p.grad.norm()) for _, p in model.named_parameters()])
    out["param_shapes"] = np.array([str(tuple(p.shape)) for _, p in model.named_parameters()])
    out["state_dict_keys"] = np.array(list(model.state_dict().keys()))
    # fp64 re-run of the same train step = ground truth for the gradient norms (the fp32 step above
    # carries ~1e-3 relative accumulation noise through the train-mode BatchNorm backward).  Index
    # operators still see fp32 coordinates; gather / group / interpolate run as the reference's own
    # dtype-agnostic torch indexing.
    class _F64Ops:
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
            B, N = idx.shape[:2]
            g = ref_pu.index_points(points.permute(0, 2, 1), idx.long())
            return (g * weight.view(B, N, 3, 1)).sum(dim=2).permute(0, 2, 1)

    ref_pu.futils = _F64Ops
    model64 = ref_hn.HandTrackNet(cfg)
    deterministic_init(model64)
    model64 = model64.double().train()
    for m in model64.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    to64 = lambda d: {k: (v.double() if torch.is_tensor(v) else to64(v)) for k, v in d.items()}
    data64 = to64(data)
    _float = torch.Tensor.float
    torch.Tensor.float = lambda self: self  # hand_network.py casts its inputs with .float(); keep fp64 for this run
    torch.set_default_dtype(torch.float64)  # torch.eye / torch.ones inside the reference
    try:
        ret64 = model64(data64, dict(flags, test_flag=False))
        loss64, ret64 = model64.compute_loss(data64, ret64, dict(flags, test_flag=False))
    finally:
        torch.Tensor.float = _float
        torch.set_default_dtype(torch.float32)
    total64 = 10 * loss64["hand_pred_kp_loss"] + loss64["hand_pred_r_loss"] + loss64["hand_pred_t_loss"]
    total64.backward()
    out["train_total_loss_f64"] = np.array(float(total64))
    out["param_grad_norm_f64"] = np.array([0.0 if p.grad is None else float(p.grad.norm()) for _, p in model64.named_parameters()])
    # element-wise fp64 gradients (first 1024 entries of each live parameter): the truth two fp32 implementations are judged
    # against when they disagree with each other (train-mode BatchNorm chains amplify fp32 round-off to the per-cent level)
    for n, p in model64.named_parameters():
        if p.grad is not None:
            out["g64/" + n] = p.grad.flatten()[:1024].numpy().astype(np.float64)
    ref_pu.futils = torch_ops
    np.savez_compressed(os.path.join(HERE, "handtracknet_reference.npz"), **out)
    n_none = int(out["param_grad_is_none"].sum())
    numel_none = sum(p.numel() for _, p in model.named_parameters() if p.grad is None)
    return {"params": len(names), "grad_none_tensors": n_none, "grad_none_numel": numel_none,
            "total_numel": sum(p.numel() for p in model.parameters()), "train_total_loss": float(total)}


if __name__ == "__main__":
    assert os.path.isdir(REF), "golden vectors can only be regenerated where /root/reference exists"
    O.build()
    ref_pu, ref_hn = import_reference()
    rep = gen_ops(ref_pu)
    rep.update(gen_network(ref_pu, ref_hn))
    import json
    with open(os.path.join(HERE, "GOLDEN_REPORT.json"), "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep, indent=1))
