"""GPU: the MI355X-side extensions (include/pn2_ext.h) against plain torch fp32 references."""
import os
import sys

import numpy as np
import pytest
import torch

from _netinit import deterministic_init, make_cfg, synthetic_frames

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "network"))
pytestmark = pytest.mark.gpu


def _kabsch_ref(x, y):
    x, y = x.double(), y.double()
    cx, cy = x.mean(1, keepdim=True), y.mean(1, keepdim=True)
    w = (x - cx).transpose(1, 2) @ (y - cy)
    u, _, vh = torch.linalg.svd(w)
    v = vh.transpose(1, 2)
    d = torch.det(v @ u.transpose(1, 2))
    fix = torch.eye(3, dtype=torch.float64).repeat(x.shape[0], 1, 1)
    fix[:, 2, 2] = d
    R = v @ fix @ u.transpose(1, 2)
    t = cy - cx @ R.transpose(1, 2)
    return R.float(), t.transpose(1, 2).float()


@pytest.mark.parametrize("B,num,shared", [(64, 6, False), (5, 6, True), (1, 14, False), (300, 4, False)])
def test_kabsch_matches_svd(B, num, shared):
    from hotrack_amd import ext
    g = torch.Generator().manual_seed(B + num)
    x = torch.randn(1 if shared else B, num, 3, generator=g) * 0.05
    q = torch.randn(B, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    w_, a, b_, c = q.unbind(1)
    Rgt = torch.stack([1 - 2 * (b_ * b_ + c * c), 2 * (a * b_ - c * w_), 2 * (a * c + b_ * w_),
                       2 * (a * b_ + c * w_), 1 - 2 * (a * a + c * c), 2 * (b_ * c - a * w_),
                       2 * (a * c - b_ * w_), 2 * (b_ * c + a * w_), 1 - 2 * (a * a + b_ * b_)], 1).view(B, 3, 3)
    y = x.expand(B, -1, -1) @ Rgt.transpose(1, 2) + torch.randn(B, 1, 3, generator=g) + 0.002 * torch.randn(B, num, 3, generator=g)
    R, t = ext.kabsch(x.cuda(), y.cuda())
    Rr, tr = _kabsch_ref(x.expand(B, -1, -1), y)
    assert torch.allclose(R.cpu(), Rr, atol=2e-6) and torch.allclose(t.cpu(), tr, atol=2e-6)
    assert torch.allclose(torch.det(R.cpu()), torch.ones(B), atol=1e-5)


def test_kabsch_degenerate_and_ill_separated_fits():
    """The fit's fast path (largest root of the characteristic polynomial + an adjugate column) hands ill-separated problems to the
    Jacobi sweep: planar / collinear / coincident point sets, mirrored targets (the SVD's det fix-up case), near-symmetric sets, tiny
    and huge scales.  Wherever the optimum is unique R must equal the fp64 SVD's; always R is a proper rotation and the residual the
    optimum's."""
    from hotrack_amd import ext
    g = torch.Generator().manual_seed(77)
    sets = []
    x = torch.randn(6, 3, generator=g) * 0.05
    planar = x.clone(); planar[:, 2] = 0
    line = torch.linspace(-1, 1, 6).view(6, 1) * torch.tensor([[0.03, 0.01, -0.02]])
    sets += [("generic", x, True), ("planar", planar, True), ("collinear", line, False), ("coincident", torch.zeros(6, 3) + 0.1, False),
             ("tiny", x * 1e-4, True), ("huge", x * 1e3, True),
             ("tetra", torch.tensor([[1., 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1], [1., 1, 1], [1, -1, -1]]) * 0.04, True)]
    q = torch.randn(4, generator=g); q = q / q.norm()
    w_, a, b_, c = q
    Rgt = torch.tensor([[1 - 2 * (b_ * b_ + c * c), 2 * (a * b_ - c * w_), 2 * (a * c + b_ * w_)],
                        [2 * (a * b_ + c * w_), 1 - 2 * (a * a + c * c), 2 * (b_ * c - a * w_)],
                        [2 * (a * c - b_ * w_), 2 * (b_ * c + a * w_), 1 - 2 * (a * a + b_ * b_)]])
    for name, xs, unique in sets:
        for mirror in (False, True):
            y = xs @ Rgt.t() + torch.tensor([0.3, -0.2, 0.5]) + (0.0 if name in ("collinear", "coincident") else 1e-3) * xs.abs().max() * torch.randn(6, 3, generator=g)
            if mirror:
                y = y * torch.tensor([1.0, 1.0, -1.0])  # the best ROTATION onto a mirrored copy: the smallest singular direction flips
            R, t = ext.kabsch(xs[None].cuda(), y[None].cuda())
            R, t = R.cpu()[0].double(), t.cpu()[0].double().view(3)
            assert torch.isfinite(R).all() and torch.isfinite(t).all(), (name, mirror)
            assert float((R @ R.t() - torch.eye(3, dtype=torch.float64)).abs().max()) < 1e-5 and abs(float(torch.det(R)) - 1) < 1e-5, (name, mirror)
            Rr, tr = _kabsch_ref(xs[None], y[None])
            Rr, tr = Rr[0].double(), tr[0].double().view(3)
            res = lambda RR, tt: float(((xs.double() @ RR.t() + tt - y.double()) ** 2).sum())
            scale = float((y.double() - y.double().mean(0)).pow(2).sum()) + 1e-30
            assert res(R, t) <= res(Rr, tr) + 1e-6 * scale + 1e-12, (name, mirror, res(R, t), res(Rr, tr))
            if unique and not (mirror and name in ("planar", "tetra")):  # (a mirrored planar / symmetric set has a family of optima)
                assert float((R - Rr).abs().max()) < 5e-6, (name, mirror, float((R - Rr).abs().max()))


def test_hand_frame_matches_kabsch_plus_canonicalize():
    from hotrack_amd import ext
    from models.hand_utils import canonicalize
    d = synthetic_frames(3, 16, 1024)
    pts, kp, palm = d["hand_points"].cuda(), d["jittered_hand_kp"].cuda(), d["gt_hand_pose"]["palm_template"].cuda()
    idx = torch.tensor([0, 1, 5, 9, 13, 17], dtype=torch.int32, device="cuda")
    R, t, xyz2, xyz1 = ext.hand_frame(palm, kp, idx, pts, 0.2)
    Rr, tr = _kabsch_ref(palm.cpu(), kp[:, idx.long()].cpu())
    assert torch.allclose(R.cpu(), Rr, atol=2e-6) and torch.allclose(t.cpu(), tr, atol=2e-6)
    pose = {"rotation": R, "translation": t, "scale": 0.2 * torch.ones(1, device="cuda")}
    ref = canonicalize(torch.cat([pts, kp], 1).transpose(1, 2), pose).transpose(1, 2)
    assert torch.allclose(xyz2, ref[:, :1024], atol=2e-6) and torch.allclose(xyz1, ref[:, 1024:], atol=2e-6)
    R2, t2 = ext.kabsch(palm, kp[:, idx.long()].contiguous())
    assert torch.equal(R2, R) and torch.equal(t2, t)


def _mk_sa(in_ch, widths, seed):
    from models.pointnet_utils import PointNetSetAbstractionMsg_GivenCenterPoints
    m = PointNetSetAbstractionMsg_GivenCenterPoints([0.2], [16], [widths], in_channel=in_ch, knn=True)
    deterministic_init(m)
    g = torch.Generator().manual_seed(seed)
    for p in m.parameters():
        p.data.add_(0.01 * torch.randn(p.shape, generator=g))
    return m.cuda().eval()


@pytest.mark.parametrize("D,D2,widths,N,S,K", [
    (0, 0, [32, 32, 64], 1024, 256, 32), (64, 0, [64, 64, 128], 256, 128, 32), (384, 0, [128, 128, 192], 1024, 21, 16),
    (384, 0, [128, 128, 192], 1024, 21, 64), (384, 384, [128, 128, 192], 1024, 21, 16), (384, 384, [128, 128, 192], 1024, 21, 64),
    (64, 0, [64, 64, 128], 300, 37, 16), (0, 0, [32, 32, 64], 100, 5, 64)])
def test_fused_sa_scale_matches_unfused(D, D2, widths, N, S, K):
    from hotrack_amd import fused, pointnet2_utils
    from models import pointnet_utils
    pointnet_utils.set_operator_backend(pointnet2_utils)
    B = 3
    g = torch.Generator().manual_seed(D + N + K)
    m = _mk_sa(D + 3 + D2, widths, seed=K)
    xyz = torch.rand(B, 3, N, generator=g).cuda()
    pts = torch.randn(B, D, N, generator=g).cuda() if D else None
    new_xyz = torch.rand(B, 3, S, generator=g).cuda()
    cf = torch.randn(B, D2, S, generator=g).cuda() if D2 else None
    idx = torch.randint(0, N, (B, S, K), generator=g, dtype=torch.int32).cuda()
    with torch.no_grad():
        pointnet_utils.set_fused_backend(None)
        ref = m._scale(0, xyz, pts, new_xyz, idx, cf)
        pointnet_utils.set_fused_backend(fused)
        assert fused.supported(m.conv_blocks[0], K)
        got = m._scale(0, xyz, pts, new_xyz, idx, cf)
        pointnet_utils.set_fused_backend(None)
    assert got.shape == ref.shape
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2e-5 * max(scale, 1.0) + 1e-5, (float((got - ref).abs().max()), scale)


def test_fused_unsupported_shape_falls_back_to_operator_path():
    from hotrack_amd import fused, pointnet2_utils
    from models import pointnet_utils
    pointnet_utils.set_operator_backend(pointnet2_utils)
    m = _mk_sa(3, [16, 16, 24], seed=1)
    assert not fused.supported(m.conv_blocks[0], 16)
    xyz = torch.rand(2, 3, 50).cuda()
    idx = torch.randint(0, 50, (2, 4, 16), dtype=torch.int32).cuda()
    new_xyz = torch.rand(2, 3, 4).cuda()
    with torch.no_grad():
        pointnet_utils.set_fused_backend(fused)
        a = m._scale(0, xyz, None, new_xyz, idx)
        pointnet_utils.set_fused_backend(None)
        b = m._scale(0, xyz, None, new_xyz, idx)
    assert torch.allclose(a, b, atol=1e-5, rtol=1e-5)  # same operator path; only the BN folding differs


def test_new_extension_kernels_match_torch():
    from hotrack_amd import ext
    g = torch.Generator().manual_seed(3)
    B, n, m, C = 3, 300, 77, 40
    u, k = torch.rand(B, n, 3, generator=g).cuda(), torch.rand(B, m, 3, generator=g).cuda()
    w, idx = ext.three_nn_weights(u, k)
    from hotrack_amd import pointnet2_utils as ops
    d, i2 = ops.three_nn(u, k)
    assert torch.equal(idx, i2)
    r = 1.0 / (d + 1e-8)
    assert torch.allclose(w, r / r.sum(2, keepdim=True), atol=1e-6)
    feats = torch.randn(B, m, C, generator=g).cuda()
    wide = torch.zeros(B, n, C + 8).cuda()
    ext.three_interpolate_pm(feats, idx, w, wide[:, :, 4:4 + C])
    ref = ops.three_interpolate(feats.transpose(1, 2).contiguous(), idx, w).transpose(1, 2)
    assert torch.allclose(wide[:, :, 4:4 + C], ref, atol=1e-6) and float(wide[:, :, :4].abs().max()) == 0.0
    sel = torch.randint(0, m, (B, 19), generator=g, dtype=torch.int32).cuda()
    assert torch.equal(ext.gather_rows(feats, sel), torch.gather(feats, 1, sel.long().unsqueeze(-1).expand(-1, -1, C)))
    y = torch.randn(B, n, C, generator=g).cuda()
    bias = torch.randn(B, C, generator=g).cuda()
    exp = torch.relu(y + bias[:, None, :])
    assert torch.allclose(ext.bias_act_pm_(y.clone(), bias, rows_per_bias=n), exp, atol=1e-6)


@pytest.mark.parametrize("B", [1, 2, 5])
def test_fast_point_major_forward_matches_module_path(B):
    """models/fast_eval.py (point-major inference path) == HandTrackNet.forward's channel-major path."""
    from hotrack_amd import fused, pointnet2_utils
    from models import pointnet_utils
    from models.hand_network import HandTrackNet
    pointnet_utils.set_operator_backend(pointnet2_utils)
    torch.manual_seed(0)
    model = HandTrackNet(make_cfg("cuda"))
    deterministic_init(model)
    model = model.cuda().eval()
    d = synthetic_frames(11 + B, B, 1024)
    d = {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in d.items()}
    flags = {"track_flag": False, "test_flag": True, "save_flag": False, "IKNet_flag": True}
    with torch.no_grad():
        pointnet_utils.set_fused_backend(None)
        ref = model(d, dict(flags))
        try:
            pointnet_utils.set_fused_backend(fused)
            fast = model(d, dict(flags))
            assert model._fast is not None
            model.use_fast_eval = False
            mid = model(d, dict(flags))  # fused kernels, channel-major module path
        finally:
            pointnet_utils.set_fused_backend(None)
    for other in (fast, mid):
        for k in ("pred_kp", "pred_kp_handframe", "init_kp_handframe", "points_handframe"):
            assert other[k].shape == ref[k].shape
            assert torch.allclose(other[k], ref[k], atol=2e-4), (k, float((other[k] - ref[k]).abs().max()))
        assert torch.equal(other["pred_kp_vis_mask"], ref["pred_kp_vis_mask"])
    loss_a, _ = model.compute_loss(dict(d, gt_hand_kp=d["gt_hand_kp"]), fast, dict(flags))
    loss_b, _ = model.compute_loss(dict(d, gt_hand_kp=d["gt_hand_kp"]), ref, dict(flags))
    for k in loss_b:
        assert abs(float(loss_a[k]) - float(loss_b[k])) < 1e-3 * max(1.0, abs(float(loss_b[k]))), k


def test_network_with_fused_backend_matches_reference_golden():
    from hotrack_amd import fused, pointnet2_utils
    from models import pointnet_utils
    from models.hand_network import HandTrackNet
    gold = np.load(os.path.join(ROOT, "tests", "golden", "handtracknet_reference.npz"))
    pointnet_utils.set_operator_backend(pointnet2_utils)
    torch.manual_seed(0)
    model = HandTrackNet(make_cfg("cuda"))
    deterministic_init(model)
    model = model.cuda().eval()
    data = {"hand_points": torch.from_numpy(gold["in_hand_points"]).cuda(),
            "jittered_hand_kp": torch.from_numpy(gold["in_jittered_hand_kp"]).cuda(),
            "gt_hand_pose": {"palm_template": torch.from_numpy(gold["in_palm_template"]).cuda()}}
    flags = {"track_flag": False, "test_flag": True, "save_flag": False, "IKNet_flag": False}
    try:
        pointnet_utils.set_fused_backend(fused)
        with torch.no_grad():
            ret = model(data, dict(flags))
    finally:
        pointnet_utils.set_fused_backend(None)
    np.testing.assert_allclose(ret["pred_kp"].cpu().numpy(), gold["eval_pred_kp"], atol=2e-4)
    np.testing.assert_allclose(ret["pred_kp_handframe"].cpu().numpy(), gold["eval_pred_kp_handframe"], atol=2e-4)


def test_tuned_gemm_table_loads():
    """hotrack_amd/tunableop_gfx950.csv was recorded for this image / GPU: TunableOp must accept it (validators match)."""
    import os
    if os.environ.get("PN2_TUNED_GEMMS", "1") == "0":
        pytest.skip("tuned table disabled by PN2_TUNED_GEMMS=0")
    import torch.cuda.tunable as tunable
    from hotrack_amd import gemm_tuning
    assert gemm_tuning.enable() is True
    assert not tunable.is_enabled()                       # loaded, but only applied inside scope()
    with gemm_tuning.scope():
        assert tunable.is_enabled() and not tunable.tuning_is_enabled()
        assert len(tunable.get_results()) > 50
    assert not tunable.is_enabled()


def test_tail_kernels_match_torch():
    """add_layernorm / pose_head / max_rows / knn_indices (the fused element-wise runs of the 21-token tail) vs torch."""
    import torch.nn as nn
    import torch.nn.functional as F
    from hotrack_amd import ext, pointnet2_utils as ops
    g = torch.Generator().manual_seed(5)
    for rows, C in ((21, 384), (1344, 384), (5, 100), (64, 1024), (3, 64)):
        x = torch.randn(rows, C, generator=g).cuda() * 3 + 1
        y = torch.randn(rows, C, generator=g).cuda()
        b = torch.randn(C, generator=g).cuda()
        ln1, ln2 = nn.LayerNorm(C).cuda(), nn.LayerNorm(C, eps=1e-6).cuda()
        for ln in (ln1, ln2):
            ln.weight.data = torch.randn(C, generator=g).cuda()
            ln.bias.data = torch.randn(C, generator=g).cuda()
        with torch.no_grad():
            torch.testing.assert_close(ext.add_layernorm(x, ln1), ln1(x), rtol=2e-5, atol=2e-5)
            torch.testing.assert_close(ext.add_layernorm(x, ln1, ln2=ln2), ln2(ln1(x)), rtol=2e-5, atol=2e-5)
            torch.testing.assert_close(ext.add_layernorm(x, ln1, y=y, bias=b, ln2=ln2), ln2(ln1(x + y + b)), rtol=2e-5, atol=2e-5)
            torch.testing.assert_close(ext.add_layernorm(x, ln1, y=y), ln1(x + y), rtol=2e-5, atol=2e-5)
    for B, J, C in ((1, 21, 256), (64, 21, 256), (3, 5, 70)):
        h = torch.randn(B * J, C, generator=g).cuda()
        w = torch.randn(3, C, generator=g).cuda() * 0.1
        bias = torch.randn(3, generator=g).cuda()
        xyz1 = torch.randn(B, J, 3, generator=g).cuda()
        R = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))[0].cuda().contiguous()
        t = torch.randn(B, 3, 1, generator=g).cuda()
        kh, kc = ext.pose_head(h, w, bias, xyz1, R, t, 0.2)
        ref_h = F.linear(h, w, bias).view(B, J, 3) + xyz1
        torch.testing.assert_close(kh, ref_h, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(kc, torch.matmul(ref_h, R.transpose(1, 2)) * 0.2 + t.transpose(1, 2), rtol=1e-5, atol=1e-5)
    for B, R_, C in ((1, 128, 512), (64, 128, 512), (3, 1, 7), (2, 37, 130), (2, 200, 64)):
        x = torch.randn(B, R_, C, generator=g).cuda()
        assert torch.equal(ext.max_rows(x), x.max(dim=1)[0])
    xyz = torch.rand(4, 1024, 3, generator=g).cuda()
    q = torch.rand(4, 21, 3, generator=g).cuda()
    assert torch.equal(ext.knn_indices(64, q, xyz), ops.knn(64, q, xyz)[1])


@pytest.mark.parametrize("B,N,dup", [(64, 1024, False), (33, 1024, False), (3, 512, False), (2, 2048, False), (2, 1024, True), (2, 4096, False), (1, 6000, False)])
def test_fast_path_other_shapes_match_module_path(B, N, dup):
    """Fast path vs module path at BASELINE configs[1]'s exact size (64 x 1024, the bench.py headline shape) and beyond the
    default shape: the gathered-row layer-1 branch (B*N >= 32768), other point counts, and clouds with duplicated points (tied FPS arg-maxima -> the second sampling level really runs)."""
    from hotrack_amd import fused, pointnet2_utils
    from models import pointnet_utils
    from models.hand_network import HandTrackNet
    pointnet_utils.set_operator_backend(pointnet2_utils)
    torch.manual_seed(0)
    model = HandTrackNet(make_cfg("cuda"))
    deterministic_init(model)
    model = model.cuda().eval()
    d = synthetic_frames(300 + B + N, B, N)
    if dup:
        d["hand_points"][:, N // 2:] = d["hand_points"][:, : N - N // 2]
    d = {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in d.items()}
    flags = {"track_flag": False, "test_flag": True, "save_flag": False, "IKNet_flag": False}
    with torch.no_grad():
        pointnet_utils.set_fused_backend(None)
        ref = model(d, dict(flags))
        try:
            pointnet_utils.set_fused_backend(fused)
            fast = model(d, dict(flags))
            assert model._fast is not None
        finally:
            pointnet_utils.set_fused_backend(None)
    for k in ("pred_kp", "pred_kp_handframe", "points_handframe"):
        assert torch.allclose(fast[k], ref[k], atol=2e-4), (k, float((fast[k] - ref[k]).abs().max()))


@pytest.mark.parametrize("with_cadd", [False, True])
@pytest.mark.parametrize("B,N,order", [(3, 300, (16, 64)), (64, 1024, (64, 16)), (1, 1024, (16, 64))])
def test_sa_mlp_max_pair_equals_two_launches(with_cadd, B, N, order):
    """Both scales of a keypoint-query module in one persistent launch (pn2x_sa_mlp_max_pair) == two pn2x_sa_mlp_max
    launches, bit for bit (same tiles, same arithmetic; only the assignment of tiles to workgroups differs)."""
    from hotrack_amd import ext
    J, C1, C2, C3 = 21, 128, 128, 192
    g = torch.Generator(device="cuda").manual_seed(B * 7 + N)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    xyz, kp = torch.rand(B, N, 3, device="cuda", generator=g), torch.rand(B, J, 3, device="cuda", generator=g)
    a1f_all = r(B, N, 2 * C1) * 0.5
    cadd_all = r(B, J, 2 * C1) * 0.5 if with_cadd else None
    probs = []
    for i, K in enumerate(order):
        idx = torch.randint(0, N, (B, J, K), device="cuda", generator=g, dtype=torch.int32)
        probs.append(dict(idx=idx, w2=r(C2, C1) * 0.1, b2=r(C2) * 0.1, w3=r(C3, C2) * 0.1, b3=r(C3) * 0.1,
                          a1f=a1f_all[:, :, i * C1:(i + 1) * C1], xyz=xyz, cxyz=kp, wx=r(C1, 3), b1=r(C1) * 0.1,
                          cadd=None if cadd_all is None else cadd_all[:, :, i * C1:(i + 1) * C1]))
    single = torch.empty(B, J, 2 * C3, device="cuda")
    pair = torch.empty(B, J, 2 * C3, device="cuda")
    for i, p in enumerate(probs):
        ext.sa_mlp_max(p["idx"], p["w2"], p["b2"], p["w3"], p["b3"], a1f=p["a1f"], xyz=p["xyz"], cxyz=p["cxyz"], wx=p["wx"], b1=p["b1"],
                       cadd=p["cadd"], out=single[:, :, i * C3:(i + 1) * C3])
    ext.sa_mlp_max_pair(*(dict(p, out=pair[:, :, i * C3:(i + 1) * C3]) for i, p in enumerate(probs)))
    assert torch.equal(single, pair)
    # an uncovered combination (K = 32 / 64) silently takes the two-launch route and still gives the same answer
    p32 = dict(probs[0], idx=torch.randint(0, N, (B, J, 32), device="cuda", generator=g, dtype=torch.int32))
    a, b2_ = torch.empty(B, J, C3, device="cuda"), torch.empty(B, J, C3, device="cuda")
    ext.sa_mlp_max(p32["idx"], p32["w2"], p32["b2"], p32["w3"], p32["b3"], a1f=p32["a1f"], xyz=xyz, cxyz=kp, wx=p32["wx"], b1=p32["b1"],
                   cadd=p32["cadd"], out=a)
    big = next(p for p in probs if p["idx"].shape[2] == 64)
    ext.sa_mlp_max_pair(dict(p32, out=b2_), dict(big, out=torch.empty(B, J, C3, device="cuda")))
    assert torch.equal(a, b2_)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,ldx,ldo", [(1, 128, 128), (63, 128, 128), (64, 132, 128), (1000, 128, 136), (65536, 128, 128), (70001, 256, 128)])
def test_mlp2_rows_matches_torch(rows, ldx, ldo):
    """pn2x_mlp2_rows (two fused per-point layers, the fp1 tail of the inference path) vs torch fp32 / fp64."""
    from hotrack_amd import ext
    g = torch.Generator().manual_seed(rows)
    C = 128
    xb = torch.randn(rows, ldx, generator=g).cuda()
    x = xb[:, :C]
    w2, b2 = (torch.randn(C, C, generator=g) / C ** 0.5).cuda(), torch.randn(C, generator=g).cuda() * 0.1
    w3, b3 = (torch.randn(C, C, generator=g) / C ** 0.5).cuda(), torch.randn(C, generator=g).cuda() * 0.1
    ob = torch.full((rows, ldo), 7.0).cuda()
    out = ext.mlp2_rows(x, w2, b2, w3, b3, out=ob[:, :C] if ldo != C else ob)
    ref = torch.relu(torch.relu(x.double() @ w2.double().t() + b2.double()) @ w3.double().t() + b3.double()).float()
    assert torch.allclose(out[:, :C], ref, atol=2e-5, rtol=1e-5), float((out[:, :C] - ref).abs().max())
    if ldo != C:
        assert float((ob[:, C:] - 7.0).abs().max()) == 0.0  # columns beyond c3 untouched
    assert ext.mlp2_rows_supported(128, 128, 128) and not ext.mlp2_rows_supported(64, 64, 64)
    if ldx >= C + 4:  # three more input columns behind the features (the [interpolated | xyz | pad] rows of fp1)
        w2e = torch.randn(C, 3, generator=g).cuda()
        out = ext.mlp2_rows(xb, w2, b2, w3, b3, w2e=w2e)
        ref = torch.relu(torch.relu(x.double() @ w2.double().t() + xb[:, C:C + 3].double() @ w2e.double().t() + b2.double())
                         @ w3.double().t() + b3.double()).float()
        assert torch.allclose(out, ref, atol=3e-5, rtol=1e-5), float((out - ref).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("B,num,shared", [(1, 6, True), (32, 6, True), (70, 14, False)])
def test_kabsch_fit_gradient_matches_svd_autograd(B, num, shared):
    """ext.KabschFit (pn2x_kabsch / pn2x_kabsch_backward: one launch per direction) against autograd through an fp64 SVD
    of the same fit, for a loss that uses both the rotation and the translation."""
    from hotrack_amd import ext
    g = torch.Generator().manual_seed(B + num)
    x = torch.randn(1 if shared else B, num, 3, generator=g)
    Rt = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))[0]
    Rt = Rt * torch.sign(torch.det(Rt))[:, None, None]
    y = (torch.bmm(x.expand(B, -1, -1), Rt.transpose(1, 2)) + 0.3 * torch.randn(B, 1, 3, generator=g) + 0.05 * torch.randn(B, num, 3, generator=g))
    cR, ct = torch.randn(B, 3, 3, generator=g), torch.randn(B, 3, 1, generator=g)

    def ref(y64):
        x64 = x.double().expand(B, -1, -1)
        cx, cy = x64.mean(1, keepdim=True), y64.mean(1, keepdim=True)
        w = torch.bmm((x64 - cx).transpose(1, 2), y64 - cy)
        u, _, vh = torch.linalg.svd(w)
        v = vh.transpose(1, 2)
        fix = torch.eye(3, dtype=torch.float64).repeat(B, 1, 1)
        fix[:, 2, 2] = torch.det(torch.bmm(v, u.transpose(1, 2)))
        R = torch.bmm(torch.bmm(v, fix), u.transpose(1, 2))
        t = (cy - torch.bmm(cx, R.transpose(1, 2))).transpose(1, 2)
        return R, t

    y64 = y.double().requires_grad_(True)
    R64, t64 = ref(y64)
    ((R64 * cR.double()).sum() + (t64 * ct.double()).sum()).backward()
    yg = y.cuda().requires_grad_(True)
    R, t = ext.KabschFit.apply(x.cuda(), yg)
    ((R * cR.cuda()).sum() + (t * ct.cuda()).sum()).backward()
    assert torch.allclose(R.cpu().double(), R64.detach(), atol=1e-5) and torch.allclose(t.cpu().double(), t64.detach(), atol=1e-5)
    scale = float(y64.grad.abs().max())
    assert float((yg.grad.cpu().double() - y64.grad).abs().max()) < 2e-5 * max(1.0, scale)
    # rotation-only / translation-only losses (the other gradient arrives as None or zeros)
    yg2 = y.cuda().requires_grad_(True)
    (ext.KabschFit.apply(x.cuda(), yg2)[1] * ct.cuda()).sum().backward()
    y64b = y.double().requires_grad_(True)
    (ref(y64b)[1] * ct.double()).sum().backward()
    assert float((yg2.grad.cpu().double() - y64b.grad).abs().max()) < 2e-5 * max(1.0, float(y64b.grad.abs().max()))


@pytest.mark.gpu
def test_fast_path_with_fused_fp1_pair_matches_default(monkeypatch):
    """fp1's two layers through pn2x_mlp2_rows (the default, fast_eval.py) against the library GEMM pair (HOTRACK_MLP2=0)."""
    from hotrack_amd import fused, pointnet2_utils
    from models import pointnet_utils
    from models.hand_network import HandTrackNet
    pointnet_utils.set_operator_backend(pointnet2_utils)
    torch.manual_seed(0)
    model = HandTrackNet(make_cfg("cuda"))
    deterministic_init(model)
    model = model.cuda().eval()
    d = synthetic_frames(21, 3, 1024)
    d = {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in d.items()}
    flags = {"track_flag": False, "test_flag": True, "save_flag": False, "IKNet_flag": False}
    try:
        pointnet_utils.set_fused_backend(fused)
        with torch.no_grad():
            monkeypatch.setenv("HOTRACK_MLP2", "0")
            a = model(d, dict(flags))
            assert model._fast.P["fp1_fused"] is None
            monkeypatch.delenv("HOTRACK_MLP2")
            model._fast.prepare(force=True)
            assert model._fast.P["fp1_fused"] is not None
            b = model(d, dict(flags))
    finally:
        pointnet_utils.set_fused_backend(None)
    assert torch.allclose(a["pred_kp"], b["pred_kp"], atol=2e-5), float((a["pred_kp"] - b["pred_kp"]).abs().max())


@pytest.mark.gpu
def test_sa_compute_unit_cap_changes_nothing_but_the_grid():
    """pn2x_sa_set_compute_units: fewer workgroups walk the same tiles -- bit-identical output."""
    from hotrack_amd import ext
    g = torch.Generator(device="cuda").manual_seed(5)
    B, N, S, K, C1, C2, C3 = 9, 300, 21, 64, 128, 128, 192
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    args = dict(a1f=r(B, N, C1), xyz=torch.rand(B, N, 3, device="cuda", generator=g), cxyz=torch.rand(B, S, 3, device="cuda", generator=g),
                wx=r(C1, 3), b1=r(C1) * 0.1)
    idx = torch.randint(0, N, (B, S, K), device="cuda", generator=g, dtype=torch.int32)
    w2, b2, w3, b3 = r(C2, C1) * 0.1, r(C2) * 0.1, r(C3, C2) * 0.1, r(C3) * 0.1
    try:
        outs = []
        for cus in (0, 7, 100):
            ext.sa_set_compute_units(cus)
            outs.append(ext.sa_mlp_max(idx, w2, b2, w3, b3, point_major=True, **args))
    finally:
        ext.sa_set_compute_units(0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    with pytest.raises(ext._native.Pn2Error):
        ext.sa_set_compute_units(-1)


@pytest.mark.gpu
def test_serving_loop_several_graphs_in_flight_equals_eager():
    """bench.py's serving loop in miniature: three captured graphs on three streams, five batches rotated through their static
    inputs with no synchronisation between steps -- every replay must reproduce the eager forward of the batch it was given
    (no state shared between streams: folded weights, index caches, scratch, GEMM workspaces)."""
    from hotrack_amd import fused, pointnet2_utils
    from models import pointnet_utils
    from models.hand_network import HandTrackNet
    pointnet_utils.set_operator_backend(pointnet2_utils)
    torch.manual_seed(0)
    model = HandTrackNet(make_cfg("cuda"))
    deterministic_init(model)
    model = model.cuda().eval()
    flags = {"track_flag": False, "test_flag": True, "save_flag": False, "IKNet_flag": False}
    B, NB, NS = 6, 5, 3

    def to_dev(d):
        return {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in d.items()}

    batches = [to_dev(synthetic_frames(100 + i, B, 1024)) for i in range(NB)]
    keys = ("jittered_hand_kp", "hand_points")
    try:
        pointnet_utils.set_fused_backend(fused)
        with torch.no_grad():
            eager = [model(b, dict(flags))["pred_kp"].clone() for b in batches]
            streams = [torch.cuda.Stream() for _ in range(NS)]
            slots = [to_dev(synthetic_frames(100, B, 1024)) for _ in range(NS)]
            graphs, outs = [], []
            torch.cuda.synchronize()
            for i in range(NS):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    outs.append(model(slots[i], dict(flags))["pred_kp"])
                graphs.append(g)
            got = [None] * (4 * NB)
            for s in range(4 * NB):  # no sync between steps; a slot is re-used only after its stream finished the previous replay
                i = s % NS
                with torch.cuda.stream(streams[i]):
                    for k in keys:
                        slots[i][k].copy_(batches[s % NB][k], non_blocking=True)
                    slots[i]["gt_hand_pose"]["palm_template"].copy_(batches[s % NB]["gt_hand_pose"]["palm_template"], non_blocking=True)
                    graphs[i].replay()
                    got[s] = outs[i].clone()
            torch.cuda.synchronize()
    finally:
        pointnet_utils.set_fused_backend(None)
    for s, g_ in enumerate(got):
        assert torch.allclose(g_, eager[s % NB], atol=1e-5), (s, float((g_ - eager[s % NB]).abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,ldx_pad,off", [(21, 192, 96, 0, 0), (128, 131, 128, 1, 0), (2, 3, 5, 0, 0), (512, 384, 512, 4, 0), (33, 257, 65, 0, 1),
                                               (100, 128, 256, 0, 0), (21, 385, 64, 0, 0), (40, 700, 33, 3, 0), (1, 64, 64, 0, 0)])
def test_linear_small_matches_fp64(M, K, N, ldx_pad, off):
    """pn2x_linear_small through the C ABI: both variants (all operands requested up front for K <= 384, chunk-by-chunk beyond), ragged
    reductions, row strides, unaligned operands, bias / ReLU on and off; twice (bit-equal)."""
    import ctypes
    from hotrack_amd import pointnet2_hip as native
    lib = native._lib
    g = torch.Generator().manual_seed(M * 31 + K)
    xb = torch.randn(M, K + ldx_pad + off, generator=g).cuda()
    x = xb[:, off:off + K]
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for bias in (b, None):
        for relu in (0, 1):
            outs = []
            for _ in range(2):
                y = torch.full((M, N + 3), 7.0, device="cuda")
                rc = lib.pn2x_linear_small(M, K, N, ctypes.c_void_p(x.data_ptr()), x.stride(0), ctypes.c_void_p(w.data_ptr()), K,
                                           None if bias is None else ctypes.c_void_p(bias.data_ptr()), relu, ctypes.c_void_p(y.data_ptr()), N + 3,
                                           ctypes.c_void_p(st))
                assert rc == 0
                outs.append(y)
            assert torch.equal(outs[0], outs[1])
            ref = x.double() @ w.double().t() + (0 if bias is None else bias.double())
            ref = torch.relu(ref) if relu else ref
            assert float((outs[0][:, :N].double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
            assert bool((outs[0][:, N:] == 7.0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("rows,C,N,two,resid", [(21, 192, 1024, True, False), (21, 192, 1024, True, True), (21, 192, 96, False, True),
                                                (1, 64, 32, False, False), (42, 384, 200, True, True), (33, 100, 40, False, True), (64, 192, 512, True, True),
                                                (168, 192, 1024, True, True), (255, 192, 96, False, True)])
def test_ln_linear_equals_layernorm_then_linear(rows, C, N, two, resid):
    """pn2x_ln_linear_small == pn2x_add_layernorm followed by pn2x_linear_small, bit for bit (the normalised rows always; the product
    wherever the dispatcher would have used pn2x_linear_small for it), and torch within round-off."""
    from hotrack_amd import ext
    g = torch.Generator().manual_seed(rows * 7 + C)
    x = torch.randn(rows, C, generator=g).cuda()
    y = torch.randn(rows, C, generator=g).cuda() if resid else None
    yb = torch.randn(C, generator=g).cuda() if resid else None
    ln1, ln2 = torch.nn.LayerNorm(C).cuda(), (torch.nn.LayerNorm(C, eps=1e-6).cuda() if two else None)
    with torch.no_grad():
        for ln in (ln1, ln2):
            if ln is not None:
                ln.weight.copy_(torch.randn(C, generator=g).cuda())
                ln.bias.copy_(torch.randn(C, generator=g).cuda())
        w, b = (torch.randn(N, C, generator=g) / C ** 0.5).cuda(), torch.randn(N, generator=g).cuda()
        xn_ref = ext.add_layernorm(x, ln1, y=y, bias=yb, ln2=ln2)
        for relu in (True, False):
            xn, out = ext.ln_linear(x, ln1, w, b, relu=relu, y=y, ybias=yb, ln2=ln2)
            assert torch.equal(xn, xn_ref)
            ref = torch.nn.functional.linear(xn_ref.double(), w.double(), b.double())
            ref = torch.relu(ref) if relu else ref
            assert float((out.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
            if 2 <= rows <= ext.LINEAR_SMALL_MAX_ROWS and C <= ext.LINEAR_SMALL_MAX_K and N <= ext.LINEAR_SMALL_MAX_N:
                assert torch.equal(out, ext.linear(xn_ref, w, b, relu=relu))
        xn, out = ext.ln_linear(x, ln1, w, None, y=y, ybias=yb, ln2=ln2)  # no bias
        assert torch.equal(xn, xn_ref)
    assert ext.ln_linear_supported(21, 192) and ext.ln_linear_supported(168, 192) and not ext.ln_linear_supported(21, 512) and not ext.ln_linear_supported(100000, 192)


@pytest.mark.gpu
def test_fast_path_nan_contract():
    """NaN behaviour of the fused inference path, pinned against torch (the module path = the reference's composition).
    The fused kernels drop NaNs in their ReLU / max-pool maxima (sa_fused.hip is built -fno-honor-nans), so the path
    (a) flags a frame with a non-finite input point / keypoint on the device and returns NaN keypoints for THAT frame -- what
        the reference yields, where the NaN spreads through sampling, grouping and the global max-pool -- while the other
        frames of the batch are unaffected;
    (b) detects non-finite WEIGHTS once per weight change and runs the unfused module path, which propagates them."""
    from hotrack_amd import fused, pointnet2_utils
    from models import pointnet_utils
    from models.hand_network import HandTrackNet
    pointnet_utils.set_operator_backend(pointnet2_utils)
    torch.manual_seed(0)
    model = HandTrackNet(make_cfg("cuda"))
    deterministic_init(model)
    model = model.cuda().eval()
    flags = {"track_flag": False, "test_flag": True, "save_flag": False, "IKNet_flag": False}
    to_dev = lambda d: {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in d.items()}
    clean = to_dev(synthetic_frames(77, 4, 1024))
    bad = to_dev(synthetic_frames(77, 4, 1024))
    bad["hand_points"][1, 500, 1] = float("nan")     # a NaN depth sample in frame 1
    bad["jittered_hand_kp"][3, 5, 0] = float("inf")  # a non-finite initial palm keypoint in frame 3
    try:
        with torch.no_grad():
            pointnet_utils.set_fused_backend(None)
            ref = model(bad, dict(flags))["pred_kp"]
            pointnet_utils.set_fused_backend(fused)
            good = model(clean, dict(flags))["pred_kp"]
            got = model(bad, dict(flags))["pred_kp"]
            assert model._fast is not None and model._fast.finite_weights
            # (a) torch / module path: frames 1 and 3 are entirely non-finite, frames 0 and 2 finite
            for b in (1, 3):
                assert not torch.isfinite(ref[b]).any(), "module path: a non-finite input frame yields non-finite keypoints"
                assert torch.isnan(got[b]).all(), "fast path: flagged frame must be NaN, not a finite guess"
            for b in (0, 2):
                assert torch.isfinite(ref[b]).all() and torch.equal(got[b], good[b])  # neighbours in the batch untouched
            # (b) non-finite weights: detected at fold time -> module path -> NaN everywhere, as torch
            w = model.bhand.sa1.conv_blocks[0][1].weight
            keep = w.detach().clone()
            w[3, 5] = float("nan")  # in place under no_grad: bumps the parameter's version counter
            out = model(clean, dict(flags))["pred_kp"]
            assert model._fast.finite_weights is False and model._fast.P is None
            assert torch.isnan(out).all()
            w.copy_(keep)
            again = model(clean, dict(flags))["pred_kp"]
            assert model._fast.finite_weights and torch.equal(again, good)
    finally:
        pointnet_utils.set_fused_backend(None)
