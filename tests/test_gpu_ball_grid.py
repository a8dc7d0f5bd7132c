"""GPU: the cell-grid ball query (csrc/ball_query_grid.hip, pn2x_ball_query_grid) is index-exact against the oracle's
restatement of ball_query_kernel_fast (reference ball_query_gpu.cu:9-66) -- same hit set, ascending index order, first-hit
padding, all-zero rows without a hit -- on inputs chosen to break a spatial index: exact-distance ties on lattices, points on
cell faces, degenerate (planar / collinear / single-point) clouds, centroids outside the bounding box, non-finite points,
radii far below and far above the cloud's extent, index ranges that are not a multiple of the block size."""
import ctypes

import numpy as np
import pytest
import torch

from _cases import cloud, take_points

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def grid_query(radius, nsample, xyz, new_xyz):
    """Straight through the C ABI entry (the operator API switches to it only for large problems)."""
    from hotrack_amd import pointnet2_hip as native
    lib = native._lib
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    x, q = dev(xyz), dev(new_xyz)
    idx = torch.full((B, S, nsample), -7, dtype=torch.int32, device="cuda")  # every slot must be written
    words = int(lib.pn2x_ball_query_grid_scratch_words(B, N))
    scratch = torch.empty(words, dtype=torch.int32, device="cuda")
    rc = lib.pn2x_ball_query_grid(B, N, S, ctypes.c_float(radius), nsample, q.data_ptr(), x.data_ptr(), idx.data_ptr(),
                                  scratch.data_ptr(), words, None)
    torch.cuda.synchronize()
    return rc, idx.cpu().numpy()


def check(oracle, radius, nsample, xyz, new_xyz):
    rc, got = grid_query(radius, nsample, xyz, new_xyz)
    assert rc == 0, rc
    np.testing.assert_array_equal(got, oracle.ball_query(radius, nsample, xyz, new_xyz))


@pytest.mark.parametrize("B,N,S,radius,nsample,kind", [
    (2, 8192, 2048, 0.1, 64, "uniform"),    # BASELINE configs[4] sizes: sparse neighbourhoods (~34 hits), one index block
    (2, 8192, 2048, 0.2, 64, "uniform"),    # dense neighbourhoods (~270 hits): early exit after the first blocks
    (2, 5000, 300, 0.05, 100, "uniform"),   # index range not a multiple of the block size, nsample > hits
    (1, 20000, 257, 0.08, 32, "uniform"),   # ten index blocks
    (2, 4096, 512, 0.3, 16, "hand"),        # clustered cloud: runs longer than a wave
    (2, 2048, 100, 0.25, 64, "lattice"),    # lattice of spacing 0.25: d2 == r2 exactly (strict <), points on cell faces
    (2, 3000, 64, 0.5, 200, "lattice"),
    (2, 2500, 40, 0.3, 64, "dup"),          # seven distinct points: every cell run is one repeated record
    (1, 8192, 64, 1e-4, 8, "uniform"),      # radius far below the cloud's extent: the 16-cells-per-axis cap
    (1, 8192, 64, 10.0, 64, "uniform"),     # radius above the extent: a single cell, pure early-exit scan
    (1, 2048, 2048, 0.15, 1, "uniform"),    # nsample = 1
    (3, 2049, 33, 0.2, 300, "uniform"),     # nsample beyond the bitmap word of a lane
])
def test_grid_ball_query_index_exact(oracle, B, N, S, radius, nsample, kind):
    xyz = cloud(7 * N + S, B, N, kind)
    new_xyz = take_points(xyz, oracle.furthest_point_sample(xyz, min(S, N)))
    if new_xyz.shape[1] < S:
        new_xyz = np.concatenate([new_xyz, new_xyz[:, : S - new_xyz.shape[1]]], 1)
    if kind == "lattice":  # centroids off the lattice as well, some exactly one radius away from lattice points
        new_xyz = new_xyz + np.float32(0.25) * (np.arange(S) % 2)[None, :, None].astype(np.float32)
    check(oracle, radius, nsample, xyz, new_xyz)


def test_grid_ball_query_centroids_outside_the_box(oracle):
    xyz = cloud(5, 2, 4096, "uniform")
    rng = np.random.default_rng(1)
    far = xyz[:, :50] + np.float32(100.0)                       # no hit
    near = np.stack([np.full((2, 50), -0.05, np.float32), rng.random((2, 50), dtype=np.float32), rng.random((2, 50), dtype=np.float32)], -1)
    huge = np.full((2, 4, 3), 3.0e38, np.float32)               # cell coordinate saturates
    new_xyz = np.concatenate([far, near, huge, -huge], 1)
    check(oracle, 0.12, 32, xyz, new_xyz)
    rc, got = grid_query(0.12, 32, xyz, far)
    assert rc == 0 and (got == 0).all()


@pytest.mark.parametrize("shape", ["plane", "line", "point"])
def test_grid_ball_query_degenerate_clouds(oracle, shape):
    rng = np.random.default_rng(2)
    xyz = rng.random((2, 3000, 3), dtype=np.float32)
    if shape in ("plane", "line", "point"):
        xyz[..., 2] = 0.5
    if shape in ("line", "point"):
        xyz[..., 1] = 0.25
    if shape == "point":
        xyz[..., 0] = 0.75
    new_xyz = xyz[:, ::30].copy()
    new_xyz[:, ::2] += np.float32(0.03)
    check(oracle, 0.1, 48, xyz, new_xyz)


def test_grid_ball_query_non_finite_points(oracle):
    """NaN / Inf coordinates never satisfy d2 < r2 (in the reference as here) and must not derail the bounding box or the cells."""
    xyz = cloud(9, 2, 4096, "uniform")
    xyz[0, 17] = np.nan
    xyz[0, 900, 1] = np.inf
    xyz[1, 5, 0] = -np.inf
    xyz[1, 4095] = np.nan
    new_xyz = xyz[:, 100:400].copy()
    new_xyz[0, 3] = np.nan
    new_xyz[1, 7, 2] = np.inf
    check(oracle, 0.15, 40, xyz, new_xyz)


def test_operator_api_switches_to_the_grid_and_stays_capture_safe(oracle):
    """pointnet2_utils.ball_query (the reference operator API) takes the grid for large problems -- with torch-owned scratch, so
    also inside a HIP-graph capture -- and small problems keep the scan; both equal the oracle."""
    from hotrack_amd import pointnet2_utils as ops
    xyz = cloud(3, 2, 8192, "uniform")
    new_xyz = take_points(xyz, oracle.furthest_point_sample(xyz, 1024))
    ref = oracle.ball_query(0.1, 64, xyz, new_xyz)
    x, q = dev(xyz), dev(new_xyz)
    np.testing.assert_array_equal(ops.ball_query(0.1, 64, x, q).cpu().numpy(), ref)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ops.ball_query(0.1, 64, x, q)
    g.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    # the reference-signature C entry with library-owned scratch (eager only)
    from hotrack_amd import pointnet2_hip as native
    idx = torch.empty((2, 1024, 64), dtype=torch.int32, device="cuda")
    rc = native._lib.pn2_ball_query(2, 8192, 1024, ctypes.c_float(0.1), 64, q.data_ptr(), x.data_ptr(), idx.data_ptr(), None)
    torch.cuda.synchronize()
    assert rc == 0
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)


def test_grid_rejects_what_it_does_not_cover():
    from hotrack_amd import pointnet2_hip as native
    lib = native._lib
    x = torch.rand(1, 100, 3, device="cuda")
    idx = torch.empty((1, 10, 4), dtype=torch.int32, device="cuda")
    sc = torch.empty(int(lib.pn2x_ball_query_grid_scratch_words(1, 100)), dtype=torch.int32, device="cuda")
    assert lib.pn2x_ball_query_grid(1, 100, 10, ctypes.c_float(0.1), 4, x.data_ptr(), x.data_ptr(), idx.data_ptr(), sc.data_ptr(), sc.numel(), None) == -3
    assert lib.pn2x_ball_query_grid(1, 100, 10, ctypes.c_float(0.1), 4, x.data_ptr(), x.data_ptr(), idx.data_ptr(), None, 0, None) == -2
