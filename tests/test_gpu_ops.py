"""GPU parity: hotrack_amd HIP operators (through the C-ABI) vs the CPU oracle.

Index outputs must be bit-exact; float outputs within 1e-5 (BASELINE.json north_star).
"""
import os
import sys

import numpy as np
import pytest
import torch

from _cases import cloud, take_points

pytestmark = pytest.mark.gpu

ATOL = 1e-5


@pytest.fixture(scope="module")
def ops():
    from hotrack_amd import pointnet2_utils
    return pointnet2_utils


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


FPS_CASES = [
    # (B, N, M, kind)
    (2, 1, 1, "uniform"), (2, 3, 3, "uniform"), (3, 21, 8, "uniform"), (2, 64, 64, "uniform"),
    (2, 100, 40, "uniform"), (4, 256, 128, "uniform"), (4, 1000, 256, "uniform"), (8, 1024, 256, "uniform"),
    (4, 1024, 256, "hand"), (2, 2560, 512, "uniform"), (2, 5120, 1024, "uniform"), (1, 8192, 2048, "uniform"),
    (3, 1024, 300, "lattice"), (3, 1000, 300, "lattice"), (2, 343, 343, "lattice"), (2, 2560, 200, "lattice"),
    (2, 8192, 300, "lattice"), (2, 512, 64, "dup"), (2, 1024, 64, "dup"), (2, 33, 33, "lattice"),
    (1, 12000, 64, "uniform"), (1, 16384, 40, "lattice"), (2, 4096, 512, "uniform"), (2, 2048, 700, "lattice"),
]


@pytest.mark.parametrize("B,N,M,kind", FPS_CASES)
def test_fps_index_exact(ops, oracle, B, N, M, kind):
    xyz = cloud(B * 1000 + N, B, N, kind)
    ref = oracle.furthest_point_sample(xyz, M)
    got = ops.furthest_point_sample(dev(xyz), M)
    assert got.dtype == torch.int32 and tuple(got.shape) == (B, M)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


# large clouds (16 points per lane), degenerate extents, all-equal points
FPS_LARGE_CASES = [
    (2, 8192, 2048, "uniform"), (2, 8192, 700, "lattice"), (2, 4097, 300, "uniform"), (3, 5000, 1500, "hand"), (2, 6000, 600, "dup"),
    (1, 8192, 8192, "uniform"), (2, 5832, 900, "lattice"), (2, 7000, 256, "plane"), (2, 4500, 128, "line"), (1, 5000, 80, "same"),
    # 16 points per lane: range ends, a partly filled last register slot, few picks / many picks
    (2, 2049, 300, "uniform"), (1, 8191, 500, "lattice"), (2, 3000, 3000, "hand"), (3, 8192, 64, "dup"),
]


def _large_cloud(seed, B, N, kind):
    if kind in ("uniform", "lattice", "dup", "hand"):
        return cloud(seed, B, N, kind)
    rng = np.random.default_rng(seed)
    x = rng.random((B, N, 3), dtype=np.float32)
    if kind == "plane":       # zero extent along one axis
        x[:, :, 2] = 0.25
    elif kind == "line":
        x[:, :, 1:] = -1.5
    elif kind == "same":      # every point identical: all distances 0, every arg-max a tie over all points
        x[:] = 0.5
    return x


@pytest.mark.parametrize("B,N,M,kind", FPS_LARGE_CASES)
def test_fps_large_clouds_index_exact(ops, oracle, B, N, M, kind):
    xyz = _large_cloud(B * 77 + N, B, N, kind)
    ref = oracle.furthest_point_sample(xyz, M)
    got = ops.furthest_point_sample(dev(xyz), M)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("threads", ["64", "256", "1024"])
def test_fps_thread_configs_agree(ops, oracle, threads, monkeypatch):
    """Every (threads, points-per-lane) layout must give the reference's tie order."""
    monkeypatch.setenv("PN2_FPS_THREADS", threads)
    for N, M, kind in [(1024, 256, "uniform"), (1024, 300, "lattice"), (1000, 128, "lattice"), (700, 128, "uniform")]:
        xyz = cloud(7 + N, 3, N, kind)
        ref = oracle.furthest_point_sample(xyz, M)
        got = ops.furthest_point_sample(dev(xyz), M)
        np.testing.assert_array_equal(got.cpu().numpy(), ref)


# 16385 .. 65536 points: running distances in registers, coordinates streamed (fps_stream_kernel); beyond: the HBM-temp kernel
FPS_STREAM_CASES = [(2, 16385, 200, "uniform"), (1, 20000, 300, "lattice"), (2, 32768, 128, "uniform"), (1, 32769, 64, "dup"),
                    (1, 65536, 96, "uniform"), (2, 50000, 150, "hand"), (1, 40000, 40, "same"), (1, 65535, 50, "line")]


@pytest.mark.parametrize("B,N,M,kind", FPS_STREAM_CASES)
def test_fps_streamed_clouds_index_exact(ops, oracle, B, N, M, kind, monkeypatch):
    from hotrack_amd import pointnet2_hip as native
    xyz = _large_cloud(B * 31 + N, B, N, kind)
    ref = oracle.furthest_point_sample(xyz, M)
    out = torch.empty((B, M), dtype=torch.int32, device="cuda")
    native.furthest_point_sampling_wrapper(B, N, M, dev(xyz), None, out)  # no scratch buffer needed
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    np.testing.assert_array_equal(ops.furthest_point_sample(dev(xyz), M).cpu().numpy(), ref)


def test_fps_op_falls_back_to_the_scratch_kernel_when_the_streamed_kernel_is_unavailable(oracle):
    """the operator passes no scratch for 16385 .. 65536 points (fps_stream_kernel needs 144-156 KiB of dynamic LDS);
    where that kernel cannot run -- PN2_FPS_NO_STREAM here, a refused hipFuncSetAttribute elsewhere -- the C ABI answers
    PN2_ESCRATCH and the operator retries with the reference's temp buffer.  Child process: the switch is read once."""
    import subprocess
    B, N, M = 1, 20000, 40
    xyz = cloud(11, B, N, "uniform")
    ref = oracle.furthest_point_sample(xyz, M)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_fps_child_in.npy")
    np.save(path, xyz)
    try:
        code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from hotrack_amd import pointnet2_utils as ops;"
                "x = torch.from_numpy(np.load(%r)).cuda(); print(','.join(str(int(v)) for v in ops.furthest_point_sample(x, %d)[0].cpu()))"
                % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path, M))
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PN2_FPS_NO_STREAM="1"), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        got = np.array([int(v) for v in out.stdout.strip().splitlines()[-1].split(",")], dtype=np.int32)
    finally:
        os.remove(path)
    np.testing.assert_array_equal(got, ref[0])


def test_fps_large_needs_temp(oracle):
    from hotrack_amd import pointnet2_hip as native
    B, N, M = 1, 70000, 24
    xyz = cloud(5, B, N, "uniform")
    out = torch.empty((B, M), dtype=torch.int32, device="cuda")
    with pytest.raises(native.Pn2Error):
        native.furthest_point_sampling_wrapper(B, N, M, dev(xyz), None, out)
    temp = torch.full((B, N), 1e10, dtype=torch.float32, device="cuda")
    native.furthest_point_sampling_wrapper(B, N, M, dev(xyz), temp, out)
    ref = oracle.furthest_point_sample(xyz, M)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    from hotrack_amd import pointnet2_utils
    np.testing.assert_array_equal(pointnet2_utils.furthest_point_sample(dev(xyz), M).cpu().numpy(), ref)  # the op brings its own scratch


BALL_CASES = [
    # (B, N, S, radius, nsample, kind)
    (4, 1024, 256, 0.1, 32, "hand"), (4, 256, 128, 0.2, 32, "hand"), (2, 1024, 256, 0.1, 32, "uniform"),
    (2, 1024, 256, 0.5, 32, "uniform"), (2, 1000, 100, 0.3, 1, "uniform"), (2, 777, 50, 0.25, 64, "lattice"),
    (1, 8192, 2048, 0.2, 64, "uniform"), (1, 8192, 512, 0.1, 64, "uniform"), (2, 5000, 300, 0.05, 100, "uniform"),
    (2, 64, 64, 0.01, 16, "uniform"), (2, 3, 2, 10.0, 8, "uniform"), (3, 1024, 21, 0.2, 130, "uniform"),
]


@pytest.mark.parametrize("B,N,S,radius,nsample,kind", BALL_CASES)
def test_ball_query_index_exact(ops, oracle, B, N, S, radius, nsample, kind):
    xyz = cloud(11 * N + S, B, N, kind)
    fps = oracle.furthest_point_sample(xyz, S)
    new_xyz = take_points(xyz, fps)
    if kind == "lattice":  # put centroids off-lattice too, some exactly on the sphere
        new_xyz = new_xyz + np.float32(0.25)
    ref = oracle.ball_query(radius, nsample, xyz, new_xyz)
    got = ops.ball_query(radius, nsample, dev(xyz), dev(new_xyz))
    assert got.dtype == torch.int32
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


def test_ball_query_no_hits_rows_are_zero(ops, oracle):
    xyz = cloud(3, 2, 500, "uniform")
    new_xyz = xyz[:, :40] + np.float32(100.0)
    got = ops.ball_query(0.1, 16, dev(xyz), dev(new_xyz)).cpu().numpy()
    assert (got == 0).all()
    np.testing.assert_array_equal(got, oracle.ball_query(0.1, 16, xyz, new_xyz))


NN_CASES = [(4, 256, 128), (4, 1024, 256), (2, 1000, 333), (2, 50, 2), (2, 10, 1), (2, 5000, 3000), (3, 64, 64)]


@pytest.mark.parametrize("B,n,m", NN_CASES)
@pytest.mark.parametrize("kind", ["uniform", "lattice"])
def test_three_nn(ops, oracle, B, n, m, kind):
    unknown = cloud(n, B, n, kind)
    known = cloud(m + 1, B, m, kind)
    d2, idx = oracle.three_nn(unknown, known)
    gd, gi = ops.three_nn(dev(unknown), dev(known))
    np.testing.assert_array_equal(gi.cpu().numpy(), idx)
    np.testing.assert_allclose(gd.cpu().numpy(), np.sqrt(d2), rtol=0, atol=ATOL)


KNN_CASES = [(4, 21, 1024, 16), (4, 21, 1024, 64), (2, 21, 1024, 4), (2, 21, 1024, 200), (2, 30, 1000, 50),
             (2, 17, 100, 7), (2, 5, 64, 64), (2, 9, 2048, 33), (1, 40, 3000, 20), (2, 8, 10, 16), (2, 4, 1, 1),
             (2, 128, 512, 32),
             # m > 2048: the streaming kernel (sorted LDS list), one case per list-slot instantiation + ragged last step
             (2, 21, 8192, 64), (2, 33, 4097, 16), (1, 10, 5000, 100), (1, 7, 20000, 200), (2, 9, 2049, 129)]


@pytest.mark.parametrize("B,n,m,k", KNN_CASES)
@pytest.mark.parametrize("kind", ["uniform", "lattice"])
def test_knn(ops, oracle, B, n, m, k, kind):
    unknown = cloud(n + k, B, n, kind)
    known = cloud(m, B, m, kind)
    d2, idx = oracle.knn(k, unknown, known)
    gd, gi = ops.knn(k, dev(unknown), dev(known))
    np.testing.assert_array_equal(gi.cpu().numpy(), idx)
    ref = np.sqrt(d2)
    g = gd.cpu().numpy()
    assert np.array_equal(np.isinf(g), np.isinf(ref))
    np.testing.assert_allclose(np.where(np.isinf(g), 0, g), np.where(np.isinf(ref), 0, ref), rtol=0, atol=ATOL)


GROUP_CASES = [(4, 3, 1024, 256, 32), (4, 0, 1024, 256, 32), (4, 64, 256, 128, 32), (2, 384, 1024, 21, 16),
               (2, 384, 1024, 21, 64), (2, 5, 100, 7, 3), (1, 67, 8192, 2048, 64), (2, 1, 1, 1, 1), (2, 130, 333, 10, 5),
               (8, 33, 8192, 1024, 64), (5, 30, 16384, 2048, 64),  # the last two take the LDS-staged forward (rows beyond L2)
               # long position lists -> the chunked atomics-free backward (scatter_cm.hip, round 6): odd list lengths (scalar staging, a
               # partial last chunk), one / odd channel counts (ragged channel group), more than 8192 points (two target ranges),
               # 4 and 8 targets per thread
               (2, 5, 9000, 1667, 9), (1, 1, 12000, 3000, 7), (3, 4, 4097, 700, 30), (2, 7, 3000, 2100, 8), (1, 3, 15000, 4099, 5)]


def test_group_points_grad_long_lists_with_padding_skew(ops):
    """The chunked backward on the kind of list a ball query produces at the configs[4] shape: every group's tail repeats its first
    index (padding), a few points are everybody's neighbour (thousands of contributions to one target inside one chunk: the
    per-target list walk runs long), most points receive nothing.  Against an fp64 index_add."""
    B, C, N, P, S = 2, 6, 8192, 2048, 64
    g = torch.Generator().manual_seed(5)
    idx = torch.randint(0, N, (B, P, S), generator=g)
    fill = torch.randint(1, S, (B, P, 1), generator=g)                      # valid entries per group
    pos = torch.arange(S).view(1, 1, S)
    idx = torch.where(pos < fill, idx, idx[:, :, :1].expand(-1, -1, S))     # padding = the group's first index
    hot = torch.randint(0, N, (B, 4), generator=g)
    idx[:, ::7, 0] = hot[:, :1]                                             # hot targets: ~300 groups x up to 64 repeats
    idx[:, 1::11, 0] = hot[:, 1:2]
    idx = torch.where(pos < fill, idx, idx[:, :, :1].expand(-1, -1, S)).to(torch.int32)
    feat = torch.randn(B, C, N, generator=g).cuda().requires_grad_(True)
    go = torch.randn(B, C, P, S, generator=g).cuda()
    out = ops.grouping_operation(feat, idx.cuda())
    out.backward(go)
    ref = torch.zeros(B, C, N, dtype=torch.float64)
    ref.scatter_add_(2, idx.long().view(B, 1, P * S).expand(-1, C, -1), go.cpu().double().view(B, C, P * S))
    err = float((feat.grad.cpu().double() - ref).abs().max())
    assert err <= 1e-5 * max(1.0, float(ref.abs().max())), (err, float(ref.abs().max()))  # (a hot target sums ~10^4 values)
    assert int((ref.abs().sum(1) == 0).sum()) > 0  # ... and some points received nothing


@pytest.mark.parametrize("B,C,N,P,S", GROUP_CASES)
def test_group_points_forward_backward(ops, oracle, B, C, N, P, S):
    rng = np.random.default_rng(C * 31 + N)
    feat = rng.normal(size=(B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, (B, P, S)).astype(np.int32)
    f = dev(feat).requires_grad_(True)
    out = ops.grouping_operation(f, dev(idx))
    assert tuple(out.shape) == (B, C, P, S)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), oracle.group_points(feat, idx))  # pure copy: bit exact
    if C == 0:
        return
    go = rng.normal(size=(B, C, P, S)).astype(np.float32)
    out.backward(dev(go))
    ref = oracle.group_points_grad(go, idx, N)
    np.testing.assert_allclose(f.grad.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,C,N,M", [(4, 3, 1024, 256), (4, 3, 256, 128), (2, 64, 500, 77), (2, 7, 9, 20)])
def test_gather_forward_backward(ops, oracle, B, C, N, M):
    rng = np.random.default_rng(N + M)
    feat = rng.normal(size=(B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, (B, M)).astype(np.int32)
    f = dev(feat).requires_grad_(True)
    out = ops.gather_operation(f, dev(idx))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), oracle.gather_points(feat, idx))
    go = rng.normal(size=(B, C, M)).astype(np.float32)
    out.backward(dev(go))
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.gather_points_grad(go, idx, N), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,C,M,n", [(4, 256, 128, 256), (4, 128, 256, 1024), (2, 5, 3, 10), (2, 33, 700, 999),
                                     (1, 16, 20000, 300), (3, 70, 64, 1000), (1, 9, 100, 4096)])  # last two: LDS-staged forward, ragged channel chunk / several query chunks
def test_three_interpolate_forward_backward(ops, oracle, B, C, M, n):
    rng = np.random.default_rng(M + n)
    feat = rng.normal(size=(B, C, M)).astype(np.float32)
    idx = rng.integers(0, M, (B, n, 3)).astype(np.int32)
    w = rng.random((B, n, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    f = dev(feat).requires_grad_(True)
    out = ops.three_interpolate(f, dev(idx), dev(w))
    np.testing.assert_allclose(out.detach().cpu().numpy(), oracle.three_interpolate(feat, idx, w), rtol=0, atol=ATOL)
    go = rng.normal(size=(B, C, n)).astype(np.float32)
    out.backward(dev(go))
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.three_interpolate_grad(go, idx, w, M), rtol=1e-5, atol=1e-5)


def test_grad_accumulates_into_existing_buffer(oracle):
    """The reference kernels atomicAdd into the caller's buffer; so do ours (+= semantics)."""
    from hotrack_amd import pointnet2_hip as native
    rng = np.random.default_rng(0)
    B, C, N, P, S = 2, 6, 50, 9, 4
    go = rng.normal(size=(B, C, P, S)).astype(np.float32)
    idx = rng.integers(0, N, (B, P, S)).astype(np.int32)
    init = rng.normal(size=(B, C, N)).astype(np.float32)
    buf = dev(init.copy())
    native.group_points_grad_wrapper(B, C, N, P, S, dev(go), dev(idx), buf)
    np.testing.assert_allclose(buf.cpu().numpy(), init + oracle.group_points_grad(go, idx, N), rtol=1e-5, atol=1e-5)


def test_cpu_tensor_raises(ops):
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.furthest_point_sample(torch.rand(1, 16, 3), 4)


def test_forward_ops_deterministic(ops):
    xyz = dev(cloud(1, 8, 1024, "hand"))
    a = ops.furthest_point_sample(xyz, 256)
    b = ops.furthest_point_sample(xyz, 256)
    assert torch.equal(a, b)
    new = torch.gather(xyz, 1, a.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    i1 = ops.ball_query(0.1, 32, xyz, new)
    i2 = ops.ball_query(0.1, 32, xyz, new)
    assert torch.equal(i1, i2)


def test_non_default_stream(ops, oracle):
    xyz = cloud(9, 4, 1024, "uniform")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        got = ops.furthest_point_sample(dev(xyz), 128)
    s.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy(), oracle.furthest_point_sample(xyz, 128))


def test_fps_two_level_shortcut():
    """ext.fps_two_level == FPS(xyz, m1) followed by FPS(xyz[i1], m2), bit for bit, on tie-free clouds (second pass
    skipped: idx2 = 0..m2-1 must then BE the answer) and on lattice / duplicated clouds (ties -> second pass runs)."""
    from _cases import cloud
    from hotrack_amd import ext, pointnet2_utils as ops
    from oracle import pn2_oracle as O
    skipped = ran = 0
    for seed, (B, N, m1, m2, kind) in enumerate([(4, 1024, 256, 128, "uniform"), (4, 1024, 256, 128, "hand"), (3, 1000, 256, 128, "uniform"),
                                                 (2, 1024, 256, 128, "lattice"), (2, 512, 128, 64, "dup"), (2, 343, 100, 100, "lattice"),
                                                 (2, 2560, 512, 128, "uniform"), (1, 21, 8, 4, "uniform"), (2, 64, 64, 1, "uniform")]):
        xyz = cloud(4000 + seed, B, N, kind)
        d = torch.from_numpy(xyz).cuda()
        i1, l1, i2 = ext.fps_two_level(d, m1, m2)
        r1 = ops.furthest_point_sample(d, m1)
        rl1 = ext.gather_rows(d, r1)
        r2 = ops.furthest_point_sample(rl1, m2)
        assert torch.equal(i1, r1) and torch.equal(l1, rl1) and torch.equal(i2, r2), (seed, kind)
        # and against the oracle's two passes
        o1 = O.furthest_point_sample(xyz, m1)
        ol1 = np.take_along_axis(xyz, o1[..., None].astype(np.int64).repeat(3, -1), 1)
        assert np.array_equal(i2.cpu().numpy(), O.furthest_point_sample(ol1, m2))
        ident = torch.arange(m2, dtype=torch.int32, device="cuda").expand(B, m2)
        if kind in ("uniform", "hand"):
            assert torch.equal(i2, ident)      # the prefix property itself
            skipped += 1
        else:
            ran += 1
    assert skipped and ran
    # a mixed batch: cloud 0 generic, cloud 1 a lattice -> per-cloud decision
    xyz = np.concatenate([cloud(1, 1, 1024, "uniform"), cloud(2, 1, 1024, "lattice")])
    d = torch.from_numpy(xyz).cuda()
    i1, l1, i2 = ext.fps_two_level(d, 256, 128)
    assert torch.equal(i2, ops.furthest_point_sample(ext.gather_rows(d, ops.furthest_point_sample(d, 256)), 128))


def test_ball_query_picks_matches_gather_then_query():
    from _cases import cloud
    from hotrack_amd import ext, pointnet2_utils as ops
    for seed, (B, N, S, r, K, kind) in enumerate([(4, 1024, 256, 0.1, 32, "hand"), (3, 256, 128, 0.2, 32, "hand"), (2, 5000, 300, 0.05, 64, "uniform"),
                                                  (2, 300, 7, 0.3, 1, "lattice"), (1, 21, 21, 0.5, 16, "uniform")]):
        d = torch.from_numpy(cloud(8000 + seed, B, N, kind)).cuda()
        picks = ops.furthest_point_sample(d, S)
        idx, new_xyz = ext.ball_query_picks(r, K, d, picks)
        ref_xyz = ext.gather_rows(d, picks)
        assert torch.equal(new_xyz, ref_xyz)
        assert torch.equal(idx, ops.ball_query(r, K, d, ref_xyz))
    d = torch.from_numpy(cloud(1, 2, 1024, "uniform")).cuda()
    i1, l1, i2, idx1 = ext.fps_two_level(d, 256, 128, query=(0.1, 32))
    assert torch.equal(idx1, ops.ball_query(0.1, 32, d, l1)) and torch.equal(l1, ext.gather_rows(d, i1))


def test_ball_query_tie_check_colaunch_equals_separate_launches(monkeypatch):
    """pn2x_ball_query_picks_ties == pn2x_ball_query_picks2 + pn2x_fps_prefix_ties: neighbour lists, centroid coordinates and (through
    the second sampling level they steer) the tie flags, on tie-free and tied clouds, one and two centroids per wave."""
    from _cases import cloud
    from hotrack_amd import ext, pointnet2_utils as ops
    assert ext.BALL_TIE_COLAUNCH
    cases = [(1, 1024, 256, 128, 0.1, 32, "hand"), (16, 1024, 256, 128, 0.1, 32, "hand"), (3, 1000, 256, 128, 0.2, 16, "uniform"),
             (2, 1024, 256, 128, 0.15, 32, "lattice"), (2, 512, 128, 64, 0.3, 8, "dup"), (2, 343, 100, 100, 0.5, 64, "lattice"),
             (6, 2560, 512, 128, 0.1, 32, "uniform"), (1, 21, 8, 4, 1.0, 4, "uniform"),
             (1, 3584, 512, 128, 0.1, 32, "uniform"),   # the largest cloud of the co-launch: 42 KB tile + 21.5 KB static LDS <= 64 KB
             (64, 1024, 256, 128, 0.1, 32, "hand"), (1, 4096, 512, 128, 0.1, 32, "uniform")]  # many clouds / a larger cloud: two launches
    seen = []
    for seed, (B, N, m1, m2, r, K, kind) in enumerate(cases):
        d = torch.from_numpy(cloud(9000 + seed, B, N, kind)).cuda()
        a = ext.fps_two_level(d, m1, m2, query=(r, K))
        monkeypatch.setattr(ext, "BALL_TIE_COLAUNCH", False)
        b = ext.fps_two_level(d, m1, m2, query=(r, K))
        monkeypatch.setattr(ext, "BALL_TIE_COLAUNCH", True)
        for x, y in zip(a, b):
            assert torch.equal(x, y), (seed, kind)
        assert torch.equal(a[3], ops.ball_query(r, K, d, a[1])) and torch.equal(a[2], ops.furthest_point_sample(a[1], m2))
        seen.append(bool(ext._lib.pn2x_ball_query_picks_ties_supported(B, N, m1, m2)))
    assert seen == [True] * 9 + [False] * 2
    real = ext._lib.pn2x_ball_query_picks_ties
    assert real(1, 1024, 256, 0.1, 32, None, None, None, None, None, 0, 128, None, None, None) == -2   # NULL pointers
    assert real(1, 1024, 256, 0.1, 32, None, None, None, None, None, 0, 300, None, None, None) == -1   # m2 > m


def test_fps_knn_colaunch_equals_separate_launches(monkeypatch):
    """pn2x_fps_radii_knn: sampling level 1 and the keypoints' k-NN lists in one launch == the two launches, index for index
    (uniform / hand / lattice / duplicated clouds, ragged query counts, k2 = 0, the oracle's lists), and the entry is really
    the one that ran for the covered sizes."""
    from _cases import cloud
    from hotrack_amd import ext, pointnet2_utils as ops
    from oracle import pn2_oracle as O
    used = []
    real = ext._lib.pn2x_fps_radii_knn
    assert ext.FPS_KNN_COLAUNCH
    g = torch.Generator().manual_seed(5)
    cases = [(1, 1024, 256, 128, 21, 64, 16, "hand"), (8, 1024, 256, 128, 21, 64, 16, "uniform"), (64, 1024, 256, 128, 21, 64, 16, "hand"),
             (3, 1000, 256, 128, 21, 64, 16, "uniform"), (2, 600, 100, 50, 5, 16, 0, "uniform"), (2, 1024, 256, 128, 3, 200, 7, "lattice"),
             (2, 1024, 256, 128, 9, 64, 64, "dup"), (2, 513, 64, 64, 1, 1, 0, "uniform"),
             (2, 512, 128, 64, 21, 64, 16, "uniform"), (2, 2560, 512, 128, 21, 64, 16, "uniform")]  # last two: not covered -> own launch
    for seed, (B, N, m1, m2, nq, k, k2, kind) in enumerate(cases):
        xyz = cloud(6000 + seed, B, N, kind)
        d = torch.from_numpy(xyz).cuda()
        q = (torch.rand(B, nq, 3, generator=g) * 0.4 - 0.2).cuda()
        if kind == "dup":
            q[:, 0] = d[:, 0]  # a query ON a (duplicated) point: zero distances, index order decides
        i1, l1, i2, idx1, (gi, gi2) = ext.fps_two_level(d, m1, m2, query=(0.1, 32), knn=(q, k, k2))
        monkeypatch.setattr(ext, "FPS_KNN_COLAUNCH", False)
        r1, rl1, r2, ridx1, (rgi, rgi2) = ext.fps_two_level(d, m1, m2, query=(0.1, 32), knn=(q, k, k2))
        monkeypatch.setattr(ext, "FPS_KNN_COLAUNCH", True)
        assert torch.equal(i1, r1) and torch.equal(l1, rl1) and torch.equal(i2, r2) and torch.equal(idx1, ridx1), (seed, kind)
        assert torch.equal(gi, rgi) and torch.equal(gi, ops.knn(k, q, d)[1]), (seed, kind)
        assert (gi2 is None and rgi2 is None and k2 == 0) or torch.equal(gi2, gi[:, :, :k2].contiguous())
        assert np.array_equal(i1.cpu().numpy(), O.furthest_point_sample(xyz, m1))
        if B * nq * N <= 200000:
            assert np.array_equal(gi.cpu().numpy(), O.knn(k, q.cpu().numpy(), xyz)[1])
        used.append(bool(ext._lib.pn2x_fps_radii_knn_supported(N, nq, k)))
    assert used == [True] * 8 + [False] * 2
    assert real(1, 1024, 4, None, None, None, 21, 64, 16, None, None, None, None) == -2      # NULL pointers
    assert real(1, 1024, 4, None, None, None, 21, 16, 64, None, None, None, None) == -1      # k2 > k
    assert real(0, 1024, 4, None, None, None, 21, 64, 16, None, None, None, None) == 0       # empty batch


def test_three_nn_interpolate_one_launch_equals_two(monkeypatch):
    """pn2x_three_nn_interpolate_pm == pn2x_three_nn_weights + pn2x_three_interpolate_pm, bit for bit (ragged query counts, known sets
    that do not divide by four, lattice ties, column blocks of wider buffers untouched outside), incl. the sizes it hands back."""
    from _cases import cloud
    from hotrack_amd import ext
    g = torch.Generator().manual_seed(8)
    fused = []
    cases = [(16, 1024, 256, 128, 132, 0, "hand"), (64, 256, 128, 256, 320, 64, "hand"), (17, 1000, 100, 12, 16, 4, "uniform"),
             (48, 343, 343, 64, 64, 0, "lattice"), (300, 70, 17, 4, 8, 0, "uniform"), (1, 16384, 16, 8, 8, 0, "uniform"),
             (1, 1024, 256, 128, 132, 0, "hand"), (60, 300, 8, 16, 16, 0, "uniform"), (40, 500, 3000, 8, 8, 0, "uniform"),
             (300, 64, 64, 6, 8, 0, "uniform")]  # last four: two launches (few queries / tiny or huge known set / odd width)
    for seed, (B, n, m, C, ldo, col0, kind) in enumerate(cases):
        known = torch.from_numpy(cloud(7000 + seed, B, m, kind)).cuda()
        unknown = torch.from_numpy(cloud(7100 + seed, B, n, "lattice" if kind == "lattice" else "uniform")).cuda()
        pts = torch.randn(B, m, C, generator=g).cuda()
        buf_a, buf_b = torch.full((B, n, ldo), 5.0).cuda(), torch.full((B, n, ldo), 5.0).cuda()
        a = ext.three_nn_interpolate_pm(unknown, known, pts, buf_a[:, :, col0:col0 + C])
        w, i3 = ext.three_nn_weights(unknown, known)
        b = ext.three_interpolate_pm(pts, i3, w, buf_b[:, :, col0:col0 + C])
        assert torch.equal(buf_a, buf_b), (seed, float((a - b).abs().max()))
        assert float((buf_a[:, :, :col0] - 5.0).abs().max() if col0 else 0.0) == 0.0
        fused.append(bool(ext._lib.pn2x_three_nn_interpolate_pm_supported(B, n, m, C, C, ldo)))
    assert fused == [True] * 6 + [False] * 4
    real = ext._lib.pn2x_three_nn_interpolate_pm
    assert real(1, 8, 2, 4, None, None, None, 4, None, 4, None) == -1     # fewer than three known points
    assert real(1, 8, 16, 4, None, None, None, 4, None, 4, None) == -2    # NULL pointers
    assert real(0, 8, 16, 4, None, None, None, 4, None, 4, None) == 0


def test_knn_indices_prefix_output():
    from hotrack_amd import ext, pointnet2_utils as ops
    g = torch.Generator().manual_seed(21)
    for B, n, m, k, k2 in ((4, 21, 1024, 64, 16), (2, 21, 512, 64, 16), (3, 5, 100, 7, 7), (1, 1, 64, 64, 1), (2, 30, 2048, 200, 4),
                           (2, 21, 5000, 64, 16), (1, 7, 2049, 3, 2)):  # m > 2048: the general kernel behind the same index-only entry
        q, x = torch.rand(B, n, 3, generator=g).cuda(), torch.rand(B, m, 3, generator=g).cuda()
        ref = ops.knn(k, q, x)[1]
        idx, small = ext.knn_indices(k, q, x, k2=k2)
        assert torch.equal(idx, ref) and torch.equal(small, ref[:, :, :k2].contiguous())
        assert torch.equal(ext.knn_indices(k, q, x), ref)
