"""CPU: the SDF-lookup oracle (oracle/sdf_oracle.c, oracle/sdf_oracle.py) against the golden vectors produced by the
IMPORTED reference (tests/golden/make_golden_sdf.py), plus domain properties."""
import os

import numpy as np
import torch

from oracle import sdf_oracle as S
from _sdf_cases import make_volume

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases(npz, prefix):
    i = 0
    while f"{prefix}{i}_vol" in npz:
        yield i
        i += 1


def test_distance_is_bit_identical_to_the_reference():
    z = np.load(os.path.join(G, "sdf_distance.npz"))
    n = 0
    for i in _cases(z, "d"):
        res, stride = z[f"d{i}_meta"]
        mine = S.distance(z[f"d{i}_V"], z[f"d{i}_vol"], float(stride))
        assert np.array_equal(mine.view(np.int32), z[f"d{i}_ref"].view(np.int32))  # gf_optimize_obj.Distance, every bit
        n += mine.size
    assert n > 10000


def test_particle_energy_matches_reference_evaluate():
    z = np.load(os.path.join(G, "sdf_distance.npz"))
    for i in _cases(z, "d"):
        _, stride = z[f"d{i}_meta"]
        mine = S.particle_energy(z[f"e{i}_pcld"], z[f"e{i}_rot"], z[f"e{i}_trans"], z[f"d{i}_vol"], float(stride))
        # tolerance: the reference's bmm / mean accumulate in an unspecified order (fp32)
        np.testing.assert_allclose(mine, z[f"e{i}_sdf_energy"], rtol=0, atol=1e-7)
        np.testing.assert_allclose(mine * np.float32(500), z[f"e{i}_energy"], rtol=1e-6, atol=1e-6)


def test_optimize_loop_matches_reference():
    z = np.load(os.path.join(G, "sdf_optimize.npz"))
    for i in _cases(z, "o"):
        _, stride = z[f"o{i}_meta"]
        trace = []
        R, t = S.obj_optimize(z[f"o{i}_pcld"], z[f"o{i}_R_init"], z[f"o{i}_t_init"], z[f"o{i}_pre"], z[f"o{i}_vol"], float(stride),
                              trace=trace)
        np.testing.assert_allclose(R, z[f"o{i}_R_ref"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(t, z[f"o{i}_t_ref"], rtol=0, atol=1e-6)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-6) and np.linalg.det(R.astype(np.float64)) > 0.999
        assert len(trace) == 10 and trace[-1]["sdf_energy"][0] < 0.5 * trace[0]["sdf_energy"][0]  # it does optimise


def test_query_sdf_and_penetration_match_reference():
    z = np.load(os.path.join(G, "sdf_query.npz"))
    for i in _cases(z, "q"):
        res, scale = z[f"q{i}_meta"]
        idx, sdf, pen = S.nearest(z[f"q{i}_hand"], z[f"q{i}_obj_r"], z[f"q{i}_obj_t"], z[f"q{i}_vol"], float(scale))
        ref = z[f"q{i}_sdf_ref"]
        # voxel choice is discontinuous in the transformed coordinate; the reference's BLAS matmul may round a
        # coordinate differently.  Measured here: identical on every element; allow 1e-4 of them to differ.
        assert (sdf != ref).mean() <= 1e-4
        assert (pen != z[f"q{i}_pen_ref"]).mean() <= 0.02
        assert idx.min() >= 0 and idx.max() < int(res) ** 3
        assert np.array_equal(z[f"q{i}_vol"][idx], sdf)


def test_div_floor_is_torch_floor_divide():
    rng = np.random.default_rng(3)
    a = np.concatenate([rng.uniform(-0.5, 0.5, 100000), np.arange(-200, 200) * 0.003, np.arange(-200, 200) * np.float32(0.003),
                        [0.0, -0.0, 1e-30, -1e-30, 0.2249999, -0.2250001]]).astype(np.float32)
    for b in (0.003, 0.002, 0.015):
        assert np.array_equal(S.div_floor(a, b), (torch.from_numpy(a) // b).numpy())


def test_trilinear_properties():
    res, stride = 41, 0.01
    vol = make_volume(res, stride, "box", np.float32)
    # at voxel centres the interpolant returns the stored value (clamped)
    ijk = np.random.default_rng(0).integers(0, res - 1, (500, 3))
    V = (ijk * np.float32(stride) + np.float32(-0.2)).astype(np.float32)
    got = S.distance(V, vol, stride)
    flat = (ijk[:, 0] * res + ijk[:, 1]) * res + ijk[:, 2]
    np.testing.assert_allclose(got, np.clip(vol[flat], -0.05, 0.05), atol=2e-6)
    # far outside the box everything is clamped to the border voxel, then to +0.05
    assert np.all(S.distance(np.full((4, 3), 5.0, np.float32), vol, stride) == np.float32(0.05))
    # fp16 and fp32 storage of the same (fp16-representable) values agree exactly
    v16 = make_volume(res, stride, "sphere", np.float16)
    Vr = np.random.default_rng(1).uniform(-0.25, 0.25, (2000, 3)).astype(np.float32)
    assert np.array_equal(S.distance(Vr, v16, stride), S.distance(Vr, v16.astype(np.float32), stride))
