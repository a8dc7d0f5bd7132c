"""GPU: randomized shape sweep of every operator against the oracle (seeded, bounded).  Complements the
hand-picked cases of test_gpu_ops.py with odd sizes: non-powers-of-two, tiny clouds, k > m, M > N."""
import numpy as np
import pytest
import torch

from _cases import cloud

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def ops():
    from hotrack_amd import pointnet2_utils
    return pointnet2_utils


def test_fuzz_fps(ops, oracle):
    rng = np.random.default_rng(100)
    for _ in range(40):
        B, N = int(rng.integers(1, 5)), int(rng.integers(1, 3000))
        M = int(rng.integers(1, min(N, 400) + 1)) if rng.random() < 0.9 else N + int(rng.integers(1, 5))  # M > N: repeats
        kind = ["uniform", "lattice", "dup", "hand"][int(rng.integers(0, 4))]
        xyz = cloud(int(rng.integers(0, 1 << 30)), B, N, kind)
        got = ops.furthest_point_sample(dev(xyz), M).cpu().numpy()
        np.testing.assert_array_equal(got, oracle.furthest_point_sample(xyz, M), err_msg=f"B={B} N={N} M={M} {kind}")


def test_fuzz_ball_query_knn_three_nn(ops, oracle):
    rng = np.random.default_rng(200)
    for _ in range(30):
        B, N, S = int(rng.integers(1, 4)), int(rng.integers(1, 2500)), int(rng.integers(1, 200))
        kind = ["uniform", "lattice"][int(rng.integers(0, 2))]
        xyz = cloud(int(rng.integers(0, 1 << 30)), B, N, kind)
        q = cloud(int(rng.integers(0, 1 << 30)), B, S, kind)
        r, ns = float(rng.choice([0.05, 0.1, 0.25, 0.5, 2.0])), int(rng.integers(1, 70))
        np.testing.assert_array_equal(ops.ball_query(r, ns, dev(xyz), dev(q)).cpu().numpy(), oracle.ball_query(r, ns, xyz, q))
        k = int(rng.integers(1, 201))
        d2, idx = oracle.knn(k, q, xyz)
        gd, gi = ops.knn(k, dev(q), dev(xyz))
        np.testing.assert_array_equal(gi.cpu().numpy(), idx, err_msg=f"knn N={N} S={S} k={k} {kind}")
        ref = np.sqrt(d2)
        g = gd.cpu().numpy()
        assert np.array_equal(np.isinf(g), np.isinf(ref)) and np.allclose(g[~np.isinf(g)], ref[~np.isinf(ref)], atol=1e-5)
        t2, tidx = oracle.three_nn(q, xyz)
        td, ti = ops.three_nn(dev(q), dev(xyz))
        np.testing.assert_array_equal(ti.cpu().numpy(), tidx)


def test_fuzz_group_gather_interp(ops, oracle):
    rng = np.random.default_rng(300)
    for _ in range(30):
        B, C, N = int(rng.integers(1, 4)), int(rng.integers(0, 70)), int(rng.integers(1, 1500))
        P, S = int(rng.integers(1, 60)), int(rng.integers(1, 40))
        f = rng.normal(size=(B, C, N)).astype(np.float32)
        idx = rng.integers(0, N, (B, P, S)).astype(np.int32)
        ft = dev(f).requires_grad_(C > 0)
        out = ops.grouping_operation(ft, dev(idx))
        np.testing.assert_array_equal(out.detach().cpu().numpy(), oracle.group_points(f, idx))
        if C:
            go = rng.normal(size=(B, C, P, S)).astype(np.float32)
            out.backward(dev(go))
            np.testing.assert_allclose(ft.grad.cpu().numpy(), oracle.group_points_grad(go, idx, N), rtol=1e-5, atol=1e-5)
            n = int(rng.integers(1, 300))
            i3 = rng.integers(0, N, (B, n, 3)).astype(np.int32)
            w = rng.random((B, n, 3)).astype(np.float32)
            f2 = dev(f).requires_grad_(True)
            o2 = ops.three_interpolate(f2, dev(i3), dev(w))
            np.testing.assert_allclose(o2.detach().cpu().numpy(), oracle.three_interpolate(f, i3, w), rtol=0, atol=1e-5)
            g2 = rng.normal(size=(B, C, n)).astype(np.float32)
            o2.backward(dev(g2))
            np.testing.assert_allclose(f2.grad.cpu().numpy(), oracle.three_interpolate_grad(g2, i3, w, N), rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("seed", range(6))
def test_sdf_fuzz(seed):
    """Randomised SDF-lookup cases (volume size / dtype / counts / out-of-volume queries) against the oracle."""
    from hotrack_amd import sdf
    from oracle import sdf_oracle as S
    rng = np.random.default_rng(7000 + seed)
    res = int(rng.choice([2, 3, 5, 17, 32, 41, 64]))
    dt = np.float16 if seed % 2 == 0 else np.float32
    stride = float(rng.choice([0.002, 0.01, 0.05, 0.2]))
    vol = rng.uniform(-0.1, 0.1, res ** 3).astype(dt)
    dv = torch.from_numpy(vol).cuda()
    m = int(rng.integers(1, 3000))
    ext = stride * res
    V = np.concatenate([rng.uniform(-0.2 - 0.3 * ext, -0.2 + 1.3 * ext, (m, 3)),
                        -0.2 + rng.integers(0, res, (64, 3)) * stride]).astype(np.float32)
    got = sdf.distance(torch.from_numpy(V).cuda(), dv, stride).cpu().numpy()
    want = S.distance(V, vol, stride)
    assert np.array_equal(got.view(np.int32), want.view(np.int32)), (res, dt, stride)
    # particle energy: ragged n, any P
    n, P = int(rng.integers(1, 700)), int(rng.integers(1, 40))
    pc = rng.uniform(-0.2, -0.2 + ext, (n, 3)).astype(np.float32)
    q = rng.standard_normal((P, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    rot = S.quat_to_matrix(q.astype(np.float32))
    tr = rng.normal(0, 0.1 * ext, (P, 3)).astype(np.float32)
    e = sdf.particle_energy(torch.from_numpy(pc).cuda(), torch.from_numpy(rot).cuda(), torch.from_numpy(tr).cuda(), dv, stride)
    np.testing.assert_allclose(e.cpu().numpy(), S.particle_energy(pc, rot, tr, vol, stride), rtol=3e-6, atol=1e-9)
    # nearest voxel (odd res only)
    if res % 2 == 1:
        B, N = int(rng.integers(1, 20)), int(rng.integers(1, 600))
        hand = rng.uniform(-0.8 * ext, 0.8 * ext, (B, N, 3)).astype(np.float32)
        R0, t0 = rot[0], tr[0]
        qs, pen, idx = sdf.query_sdf(torch.from_numpy(hand).cuda(), torch.from_numpy(R0).cuda(), torch.from_numpy(t0).cuda(), dv, stride,
                                     with_penetration=True, with_index=True)
        oi, osdf, open_ = S.nearest(hand, R0, t0, vol, stride)
        assert np.array_equal(idx.cpu().numpy(), oi)
        assert np.array_equal(qs.cpu().numpy().view(np.uint8), osdf.view(np.uint8))
        assert np.array_equal(pen.cpu().numpy().view(np.uint8), open_.view(np.uint8))
