"""CPU: the oracle against (a) the committed golden vectors -- themselves cross-checked against
the imported reference fallback when generated (tests/golden/GOLDEN_REPORT.json) -- and
(b) independent brute-force numpy restatements and size-independent properties."""
import json
import os

import numpy as np
import pytest

from _cases import cloud, take_points

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def test_golden_report_says_oracle_agreed_with_reference_fallback():
    rep = json.load(open(os.path.join(G, "GOLDEN_REPORT.json")))
    for k in ("fps_vs_reference_fallback", "ball_rows_vs_reference_fallback", "three_nn_idx_vs_reference_fallback",
              "knn_idx_vs_reference_fallback"):
        agree, total = rep[k]
        assert total > 1000 and agree == total, (k, agree, total)
    assert rep["total_numel"] == 7919651 and rep["grad_none_numel"] == 3746944


def test_fps_golden(oracle):
    g = _load("ops_fps.npz")
    n = len([k for k in g.files if k.endswith("_xyz")])
    assert n >= 10
    for i in range(n):
        xyz, idx = g[f"fps{i}_xyz"], g[f"fps{i}_idx"]
        np.testing.assert_array_equal(oracle.furthest_point_sample(xyz, idx.shape[1]), idx)
        np.testing.assert_array_equal(oracle.furthest_point_sample(xyz, idx.shape[1], keyed=True), idx)


def test_ball_query_golden(oracle):
    g = _load("ops_ball_query.npz")
    n = len([k for k in g.files if k.endswith("_xyz")])
    for i in range(n):
        r, K = g[f"bq{i}_rk"]
        np.testing.assert_array_equal(oracle.ball_query(float(r), int(K), g[f"bq{i}_xyz"], g[f"bq{i}_new"]), g[f"bq{i}_idx"])


def test_nn_golden(oracle):
    g = _load("ops_nn.npz")
    for i in range(4):
        d2, idx = oracle.three_nn(g[f"nn{i}_u"], g[f"nn{i}_k"])
        np.testing.assert_array_equal(idx, g[f"nn{i}_idx"])
        np.testing.assert_array_equal(d2, g[f"nn{i}_d2"])
    for i in range(6):
        k = g[f"knn{i}_idx"].shape[-1]
        d2, idx = oracle.knn(k, g[f"knn{i}_u"], g[f"knn{i}_k"])
        np.testing.assert_array_equal(idx, g[f"knn{i}_idx"])
        np.testing.assert_array_equal(d2, g[f"knn{i}_d2"])


def test_group_interp_golden(oracle):
    g = _load("ops_group_interp.npz")
    for i in range(4):
        f, idx = g[f"grp{i}_f"], g[f"grp{i}_idx"]
        np.testing.assert_array_equal(oracle.group_points(f, idx), g[f"grp{i}_out"])
        np.testing.assert_allclose(oracle.group_points_grad(g[f"grp{i}_go"], idx, f.shape[2]), g[f"grp{i}_gin"], atol=1e-6)
    for i in range(2):
        f, idx, w = g[f"itp{i}_f"], g[f"itp{i}_idx"], g[f"itp{i}_w"]
        np.testing.assert_array_equal(oracle.three_interpolate(f, idx, w), g[f"itp{i}_out"])
        np.testing.assert_allclose(oracle.three_interpolate_grad(g[f"itp{i}_go"], idx, w, f.shape[2]), g[f"itp{i}_gin"], atol=1e-6)


# ---- independent restatements ------------------------------------------------------------------
def _sqd(a, b):  # same fixed contraction order, in numpy float32 with an exact fma emulated in float64
    d = a[:, None, :].astype(np.float32) - b[None, :, :].astype(np.float32)
    dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
    yy = (dy * dy).astype(np.float32)
    t = (dx.astype(np.float64) * dx.astype(np.float64) + yy.astype(np.float64)).astype(np.float32)
    return (dz.astype(np.float64) * dz.astype(np.float64) + t.astype(np.float64)).astype(np.float32)


@pytest.mark.parametrize("kind", ["uniform", "lattice"])
def test_ball_query_bruteforce(oracle, kind):
    xyz = cloud(1, 2, 400, kind)
    new = take_points(xyz, oracle.furthest_point_sample(xyz, 30)) + (np.float32(0.25) if kind == "lattice" else 0)
    r, K = 0.3, 12
    got = oracle.ball_query(r, K, xyz, new)
    r2 = np.float32(r) * np.float32(r)
    for b in range(2):
        d = _sqd(new[b], xyz[b])
        for s in range(30):
            hits = np.nonzero(d[s] < r2)[0][:K]
            exp = np.zeros(K, np.int32) if len(hits) == 0 else np.concatenate([hits, np.full(K - len(hits), hits[0])])
            np.testing.assert_array_equal(got[b, s], exp)


@pytest.mark.parametrize("kind", ["uniform", "lattice"])
def test_knn_three_nn_bruteforce(oracle, kind):
    u, k = cloud(2, 2, 50, kind), cloud(3, 2, 300, kind)
    d2, idx = oracle.knn(20, u, k)
    t2, tidx = oracle.three_nn(u, k)
    for b in range(2):
        d = _sqd(u[b], k[b])
        order = np.lexsort((np.arange(300)[None, :].repeat(50, 0), d), axis=-1)  # (d asc, index asc)
        np.testing.assert_array_equal(idx[b], order[:, :20])
        np.testing.assert_array_equal(d2[b], np.take_along_axis(d, order[:, :20], 1))
        np.testing.assert_array_equal(tidx[b], order[:, :3])
    np.testing.assert_array_equal(idx[..., :3], tidx)


def test_knn_fewer_candidates_than_k(oracle):
    u, k = cloud(4, 1, 5, "uniform"), cloud(5, 1, 3, "uniform")
    d2, idx = oracle.knn(8, u, k)
    assert np.isinf(d2[..., 3:]).all() and (idx[..., 3:] == 0).all() and np.isfinite(d2[..., :3]).all()
    t2, tidx = oracle.three_nn(u, cloud(6, 1, 2, "uniform"))
    assert np.isinf(t2[..., 2]).all() and (tidx[..., 2] == 0).all()


def test_fps_bruteforce_and_properties(oracle):
    xyz = cloud(7, 3, 500, "uniform")
    idx = oracle.furthest_point_sample(xyz, 100)
    assert (idx[:, 0] == 0).all()
    for b in range(3):
        assert len(set(idx[b].tolist())) == 100  # unique while M <= #distinct points
        t = np.full(500, 1e10, np.float32)
        old = 0
        for j in range(1, 100):
            t = np.minimum(t, _sqd(xyz[b][old:old + 1], xyz[b])[0])
            old = int(np.argmax(t))  # tie-free data: plain argmax
            assert idx[b, j] == old


def test_fps_tie_rule_examples(oracle):
    """bs=4, tie between k=1 and k=2 -> 2 wins (bit-reversed thread order), not 1."""
    pts = np.array([[[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0.5, 0, 0]]], np.float32)
    assert oracle.furthest_point_sample(pts, 2)[0, 1] == 2
    # all-duplicate cloud: every distance is 0 -> thread 0 / index 0 forever
    dup = np.ones((1, 64, 3), np.float32)
    assert (oracle.furthest_point_sample(dup, 10) == 0).all()
    assert oracle.opt_n_threads(1000) == 512 and oracle.opt_n_threads(5000) == 1024 and oracle.opt_n_threads(1) == 1


def test_group_gather_linearity_and_adjoint(oracle):
    rng = np.random.default_rng(0)
    f = rng.normal(size=(2, 5, 40)).astype(np.float32)
    idx = rng.integers(0, 40, (2, 6, 3)).astype(np.int32)
    out = oracle.group_points(f, idx)
    for b in range(2):
        np.testing.assert_array_equal(out[b], f[b][:, idx[b]])
    go = rng.normal(size=out.shape).astype(np.float32)
    gin = oracle.group_points_grad(go, idx, 40)
    # <group(f), go> == <f, group^T(go)>
    assert abs(float((out.astype(np.float64) * go).sum()) - float((f.astype(np.float64) * gin).sum())) < 1e-3
    np.testing.assert_array_equal(oracle.gather_points(f, idx[:, :, 0]), out[..., 0])
