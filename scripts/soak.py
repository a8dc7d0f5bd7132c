"""Soak test (not part of pytest): minutes of randomised launches looking for rare failures -- races in the SA kernel's
two-role pipeline, the optimiser's last-workgroup protocol, the two-level FPS shortcut and its co-launches (k-NN inside the sampling launch,
tie check inside the ball-query launch), the one-launch three-NN + interpolation, the grouped weight-gradient launch (stream-K shares over
random problem lists: csrc/train_wgrad.hip), the streamed-coordinate FPS of large clouds, the two-layer row MLP.  usage: python scripts/soak.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from _sdf_cases import make_volume, object_points, random_pose  # noqa: E402
from hotrack_amd import ext, sdf, pointnet2_utils as ops  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t_end = time.time() + budget
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
rng = np.random.default_rng(seed)
g = torch.Generator().manual_seed(seed + 1)
n_sa = n_opt = n_fps = n_wg = n_fs = n_m2 = 0
vol = torch.from_numpy(make_volume(81, 0.005, "capsule", np.float16)).cuda()
cvol = sdf.CornerVolume(vol)
while time.time() < t_end:
    # ---- fused SA kernel vs an unfused torch evaluation of the same thing, random shapes / operand sets ----
    C1, C2, C3 = [(32, 32, 64), (64, 64, 128), (128, 128, 192)][rng.integers(3)]
    K = int(rng.choice([16, 32, 64]))
    B, N, S = int(rng.integers(1, 70)), int(rng.integers(8, 1200)), int(rng.integers(1, 300))
    mode = int(rng.integers(3))
    idx = torch.randint(0, N, (B, S, K), generator=g, dtype=torch.int32).cuda()
    xyz, cxyz = torch.rand(B, N, 3, generator=g).cuda(), torch.rand(B, S, 3, generator=g).cuda()
    wx, b1 = torch.randn(C1, 3, generator=g).cuda(), torch.randn(C1, generator=g).cuda()
    a1f = torch.randn(B, N, C1, generator=g).cuda() if mode >= 1 else None
    cadd = torch.randn(B, S, C1, generator=g).cuda() if mode == 2 else None
    w2, b2 = (torch.randn(C2, C1, generator=g) * 0.1).cuda(), torch.randn(C2, generator=g).cuda()
    w3, b3 = (torch.randn(C3, C2, generator=g) * 0.1).cuda(), torch.randn(C3, generator=g).cuda()
    got = ext.sa_mlp_max(idx, w2, b2, w3, b3, a1f=a1f, xyz=xyz, cxyz=cxyz, wx=wx, b1=b1, cadd=cadd, point_major=True)
    li = idx.long()
    bi = torch.arange(B, device="cuda")[:, None, None]
    h1 = (xyz[bi, li] - cxyz[:, :, None]) @ wx.t() + b1
    if a1f is not None:
        h1 = h1 + a1f[bi, li]
    if cadd is not None:
        h1 = h1 + cadd[:, :, None]
    ref = torch.relu(torch.relu(torch.relu(h1) @ w2.t() + b2) @ w3.t() + b3).max(dim=2)[0]
    err = float((got - ref).abs().max())
    assert err <= 2e-4 * max(1.0, float(ref.abs().max())), ("sa_mlp_max", (B, N, S, K, C1, mode), err)
    again = ext.sa_mlp_max(idx, w2, b2, w3, b3, a1f=a1f, xyz=xyz, cxyz=cxyz, wx=wx, b1=b1, cadd=cadd, point_major=True)
    assert torch.equal(got, again), "sa_mlp_max not deterministic"
    n_sa += 1
    # ---- optimiser loop: bitwise repeatable (fixed reduction order, ticket protocol) ----
    if n_sa % 4 == 0:
        P, Np = int(rng.choice([64, 257, 2048])), int(rng.integers(16, 1500))
        pc = object_points(int(rng.integers(1 << 30)), Np, "capsule")
        R0, t0 = random_pose(int(rng.integers(1 << 30)))
        cam = torch.from_numpy((pc @ R0.T + t0).astype(np.float32)).cuda()
        pre = torch.randn(P, 6, generator=g).cuda()
        pre[0] = 0
        a = sdf.obj_optimize(cam, torch.from_numpy(R0).cuda(), torch.from_numpy(t0).cuda(), pre, cvol, 0.005)
        b = sdf.obj_optimize(cam, torch.from_numpy(R0).cuda(), torch.from_numpy(t0).cuda(), pre, cvol, 0.005)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), "obj_optimize not repeatable"
        assert torch.isfinite(a[0]).all() and torch.isfinite(a[1]).all()
        n_opt += 1
    # ---- two-level FPS on clouds with planted duplicates / lattice patches ----
    if n_sa % 2 == 0:
        Bf, Nf = int(rng.integers(1, 9)), int(rng.choice([256, 700, 1024, 2048]))
        x = torch.rand(Bf, Nf, 3, generator=g)
        if rng.random() < 0.5:
            x[:, : Nf // 3] = torch.round(x[:, : Nf // 3] * 6) / 6       # lattice patch -> exact ties
        if rng.random() < 0.3:
            x[:, Nf // 2:] = x[:, : Nf - Nf // 2]                          # duplicates
        x = x.cuda()
        m1 = int(rng.integers(2, min(Nf, 512)))
        m2 = int(rng.integers(1, m1 + 1))
        # with the round-4 co-launches: level 1's ball query + tie check in one launch, the k-NN lists inside the sampling launch
        nq, k = int(rng.integers(1, 30)), int(rng.integers(1, min(Nf, 200) + 1))
        k2 = int(rng.integers(0, k + 1))
        q = torch.rand(Bf, nq, 3, generator=g).cuda()
        r, ns = float(rng.choice([0.05, 0.1, 0.3])), int(rng.choice([8, 32, 64]))
        i1, l1, i2, idx1, (gi, gi2) = ext.fps_two_level(x, m1, m2, query=(r, ns), knn=(q, k, k2))
        r1 = ops.furthest_point_sample(x, m1)
        assert torch.equal(i1, r1) and torch.equal(i2, ops.furthest_point_sample(ext.gather_rows(x, r1), m2)), ("fps_two_level", Bf, Nf, m1, m2)
        assert torch.equal(idx1, ops.ball_query(r, ns, x, l1)) and torch.equal(gi, ops.knn(k, q, x)[1]), ("co-launch", Bf, Nf, m1, m2, nq, k)
        assert gi2 is None or torch.equal(gi2, gi[:, :, :k2].contiguous())
        # three-NN + interpolation as one launch vs two (bit-identical), on sizes either side of its switch-over
        Bq, nn, mm, Cc = int(rng.choice([1, 16, 40])), int(rng.choice([256, 1024])), int(rng.integers(16, 400)), int(rng.choice([4, 64, 128]))
        un, kn = torch.rand(Bq, nn, 3, generator=g).cuda(), torch.rand(Bq, mm, 3, generator=g).cuda()
        pts = torch.randn(Bq, mm, Cc, generator=g).cuda()
        oa, ob = torch.empty(Bq, nn, Cc, device="cuda"), torch.empty(Bq, nn, Cc, device="cuda")
        ext.three_nn_interpolate_pm(un, kn, pts, oa)
        w3_, i3_ = ext.three_nn_weights(un, kn)
        ext.three_interpolate_pm(pts, i3_, w3_, ob)
        assert torch.equal(oa, ob), ("three_nn_interpolate_pm", Bq, nn, mm, Cc)
        n_fps += 1
    # ---- grouped weight gradients: random problem lists (sizes, row strides, column-block outputs), twice (bit-equal), vs fp64 ----
    if n_sa % 3 == 0:
        from hotrack_amd import train_stack as ts
        probs = []
        for _ in range(int(rng.integers(1, 12))):
            r = int(rng.choice([1, 7, 33, 672, 1000, 4096, 9000, 20000]))
            nn_, kk_ = int(rng.integers(1, 300)), int(rng.integers(1, 300))
            pg, px_, pw = int(rng.integers(0, 9)), int(rng.integers(0, 9)), int(rng.integers(0, 9))
            gw = torch.randn(r, nn_ + pg, generator=g).cuda()
            xw = torch.randn(r, kk_ + px_, generator=g).cuda()
            full = torch.full((nn_, kk_ + pw), 7.0, device="cuda")
            c0 = int(rng.integers(0, pw + 1))
            probs.append((gw[:, pg // 2:pg // 2 + nn_], xw[:, px_ // 2:px_ // 2 + kk_], full, c0, kk_))
        st_ = torch.cuda.current_stream().cuda_stream
        outs = []
        for rep in range(2):
            for (gv, xv, full, c0, kk_) in probs:
                full.fill_(7.0)
            ts.wgrad_multi([ts.WgradItem(gv, xv, full, full.data_ptr() + 4 * c0, full.stride(0), gv.shape[1], kk_, st_)
                            for (gv, xv, full, c0, kk_) in probs])
            outs.append([full.clone() for (_, _, full, _, _) in probs])
        for (gv, xv, full, c0, kk_), a_, b_ in zip(probs, outs[0], outs[1]):
            assert torch.equal(a_, b_), "wgrad_multi not deterministic"
            ref64 = gv.double().t() @ xv.double()
            got = a_[:, c0:c0 + kk_].double()
            tol = 1e-5 * float(ref64.abs().max()) * max(1.0, (gv.shape[0] / 1000.0) ** 0.5) + 1e-6
            assert float((got - ref64).abs().max()) <= tol, ("wgrad_multi", tuple(gv.shape), tuple(xv.shape), c0)
            keep = torch.ones_like(a_, dtype=torch.bool)
            keep[:, c0:c0 + kk_] = False
            assert bool((a_[keep] == 7.0).all()), "wgrad_multi wrote outside its column block"
        n_wg += 1
    # ---- streamed-coordinate FPS (16385 .. 65536 points): twice (bit-equal), and every pick must be a maximiser of the running
    # distance as torch computes it (the exact tie ORDER is the pytest cases' business, against the oracle) ----
    if n_sa % 16 == 0:
        from hotrack_amd import pointnet2_hip as native
        Bs, Ns, Ms = int(rng.integers(1, 4)), int(rng.integers(16385, 65537)), int(rng.integers(2, 40))
        xs = torch.rand(Bs, Ns, 3, generator=g)
        if rng.random() < 0.5:
            xs[:, : Ns // 4] = torch.round(xs[:, : Ns // 4] * 8) / 8    # lattice patch: exact distance ties
        xs = xs.cuda()
        o1 = torch.empty(Bs, Ms, dtype=torch.int32, device="cuda")
        o2 = torch.empty_like(o1)
        native.furthest_point_sampling_wrapper(Bs, Ns, Ms, xs, None, o1)
        native.furthest_point_sampling_wrapper(Bs, Ns, Ms, xs, None, o2)
        assert torch.equal(o1, o2), "fps_stream not deterministic"
        dmin = torch.full((Bs, Ns), 1e10, device="cuda")
        for it in range(1, Ms):
            c = xs[torch.arange(Bs, device="cuda"), o1[:, it - 1].long()]
            d = xs - c[:, None]
            dd = torch.addcmul(torch.addcmul(d[..., 1] * d[..., 1], d[..., 0], d[..., 0]), d[..., 2], d[..., 2])  # fma order of sqdist: close enough for a maximum check
            dmin = torch.minimum(dmin, dd)
            picked = dmin[torch.arange(Bs, device="cuda"), o1[:, it].long()]
            assert bool((picked >= dmin.max(dim=1)[0] * (1 - 1e-6)).all()), ("fps_stream pick is not a maximiser", Bs, Ns, it)
        n_fs += 1
    # ---- mlp2_rows (transposed layer-3 product, 16-byte stores): random row counts / strides / column-block outputs vs fp64 ----
    if n_sa % 8 == 0:
        Rm = int(rng.choice([1, 5, 63, 64, 65, 1000, 4097, 70001]))
        ldx, ldo = 128 + 4 * int(rng.integers(0, 4)), 128 + 4 * int(rng.integers(0, 4))
        xb = torch.randn(Rm, ldx, generator=g).cuda()
        w2m, b2m = (torch.randn(128, 128, generator=g) / 11.3).cuda(), (torch.randn(128, generator=g) * 0.1).cuda()
        w3m, b3m = (torch.randn(128, 128, generator=g) / 11.3).cuda(), (torch.randn(128, generator=g) * 0.1).cuda()
        w2e = torch.randn(128, 3, generator=g).cuda() if ldx >= 132 and rng.random() < 0.5 else None
        ob = torch.full((Rm, ldo), 7.0, device="cuda")
        ext.mlp2_rows(xb if w2e is not None else xb[:, :128], w2m, b2m, w3m, b3m, out=ob[:, :128] if ldo != 128 else ob, w2e=w2e)
        h = xb[:, :128].double() @ w2m.double().t() + b2m.double()
        if w2e is not None:
            h = h + xb[:, 128:131].double() @ w2e.double().t()
        refm = torch.relu(torch.relu(h) @ w3m.double().t() + b3m.double())
        assert float((ob[:, :128].double() - refm).abs().max()) <= 5e-5, ("mlp2_rows", Rm, ldx, ldo)
        assert ldo == 128 or bool((ob[:, 128:] == 7.0).all()), "mlp2_rows wrote beyond its columns"
        n_m2 += 1
torch.cuda.synchronize()
print(f"soak ok: {n_fs} streamed-FPS clouds x2, {n_m2} mlp2_rows cases,", end=" ")
print(f"{n_wg} grouped weight-gradient lists x2,", end=" ")
print(f" {n_sa} SA launches x2, {n_opt} optimiser pairs, {n_fps} two-level FPS cases in {budget:.0f} s")
