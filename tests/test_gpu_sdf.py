"""GPU: the SDF-lookup kernels (hotrack_amd/csrc/sdf.hip through the C ABI of include/pn2_sdf.h) against the oracle,
the committed reference vectors, and size-independent properties at the reference's full sizes
(2048 particles x 1024 points in a 201^3 fp16 volume; 5120 x 778 in 151^3)."""
import os

import numpy as np
import pytest
import torch

from _sdf_cases import hand_particles, make_volume, object_points, particles, random_pose

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def sdf():
    from hotrack_amd import sdf as m
    return m


@pytest.fixture(scope="module")
def S():
    from oracle import sdf_oracle
    return sdf_oracle


def _d(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _cases(npz, prefix):
    i = 0
    while f"{prefix}{i}_vol" in npz:
        yield i
        i += 1


def test_distance_bit_exact_vs_reference_vectors(sdf):
    z = np.load(os.path.join(G, "sdf_distance.npz"))
    for i in _cases(z, "d"):
        _, stride = z[f"d{i}_meta"]
        got = sdf.distance(_d(z[f"d{i}_V"]), _d(z[f"d{i}_vol"]), float(stride)).cpu().numpy()
        assert np.array_equal(got.view(np.int32), z[f"d{i}_ref"].view(np.int32))  # == gf_optimize_obj.Distance, every bit


@pytest.mark.parametrize("dt", [np.float16, np.float32])
def test_distance_bit_exact_vs_oracle_random(sdf, S, dt):
    res, stride = 101, 0.004
    vol = make_volume(res, stride, "capsule", dt)
    rng = np.random.default_rng(11)
    V = np.concatenate([rng.uniform(-0.3, 0.3, (200000, 3)), rng.integers(0, res, (5000, 3)) * stride - 0.2]).astype(np.float32)
    got = sdf.distance(_d(V), _d(vol), stride).cpu().numpy()
    assert np.array_equal(got.view(np.int32), S.distance(V, vol, stride).view(np.int32))
    assert sdf.distance(torch.empty((0, 3), device="cuda"), _d(vol), stride).shape == (0,)


def test_particle_energy_vs_oracle_and_reference(sdf, S):
    z = np.load(os.path.join(G, "sdf_distance.npz"))
    for i in _cases(z, "d"):
        _, stride = z[f"d{i}_meta"]
        got = sdf.particle_energy(_d(z[f"e{i}_pcld"]), _d(z[f"e{i}_rot"]), _d(z[f"e{i}_trans"]), _d(z[f"d{i}_vol"]), float(stride))
        np.testing.assert_allclose(got.cpu().numpy(), z[f"e{i}_sdf_energy"], rtol=0, atol=1e-7)  # reference evaluate()
    # ragged n (not a multiple of the block), P = 1, fp32 volume
    res, stride = 41, 0.01
    vol = make_volume(res, stride, "box", np.float32)
    for n, P in ((1, 3), (255, 1), (257, 5), (1000, 64)):
        pc = object_points(n, n, "box")
        R0, t0 = random_pose(n)
        cam = (pc @ R0.T + t0).astype(np.float32)
        rot, tr = particles(n, P, R0, t0)
        got = sdf.particle_energy(_d(cam), _d(rot), _d(tr), _d(vol), stride).cpu().numpy()
        np.testing.assert_allclose(got, S.particle_energy(cam, rot, tr, vol, stride), rtol=2e-6, atol=1e-8)


def test_obj_optimize_vs_reference_and_oracle(sdf, S):
    z = np.load(os.path.join(G, "sdf_optimize.npz"))
    for i in _cases(z, "o"):
        _, stride = z[f"o{i}_meta"]
        R, t = sdf.obj_optimize(_d(z[f"o{i}_pcld"]), _d(z[f"o{i}_R_init"]), _d(z[f"o{i}_t_init"]), _d(z[f"o{i}_pre"]),
                                _d(z[f"o{i}_vol"]), float(stride))
        assert R.shape == (1, 3, 3) and t.shape == (1, 3, 1)
        np.testing.assert_allclose(R.cpu().numpy()[0], z[f"o{i}_R_ref"], rtol=0, atol=2e-5)   # gf_optimize_obj.optimize
        np.testing.assert_allclose(t.cpu().numpy().reshape(3), z[f"o{i}_t_ref"], rtol=0, atol=2e-6)
    # iteration by iteration against the oracle loop (same inputs, 1..4 iterations)
    i = 0
    _, stride = z[f"o{i}_meta"]
    for iters in (0, 1, 2, 4):
        Ro, to = S.obj_optimize(z[f"o{i}_pcld"], z[f"o{i}_R_init"], z[f"o{i}_t_init"], z[f"o{i}_pre"], z[f"o{i}_vol"], float(stride),
                                iterations=iters)
        R, t = sdf.obj_optimize(_d(z[f"o{i}_pcld"]), _d(z[f"o{i}_R_init"]), _d(z[f"o{i}_t_init"]), _d(z[f"o{i}_pre"]),
                                _d(z[f"o{i}_vol"]), float(stride), iterations=iters)
        np.testing.assert_allclose(R.cpu().numpy()[0], Ro, rtol=0, atol=1e-5)
        np.testing.assert_allclose(t.cpu().numpy().reshape(3), to, rtol=0, atol=1e-6)


def test_obj_optimize_no_better_particle_keeps_pose(sdf):
    """All particles identical to the current pose -> `success` False branch (optimization_obj.py:277-279, 290)."""
    res, stride = 41, 0.01
    vol = make_volume(res, stride, "sphere", np.float16)
    pc = object_points(5, 128, "sphere")
    R0, t0 = random_pose(5)
    cam = (pc @ R0.T + t0).astype(np.float32)
    pre = np.zeros((256, 6), np.float32)
    R, t = sdf.obj_optimize(_d(cam), _d(R0), _d(t0), _d(pre), _d(vol), stride, iterations=3)
    assert np.array_equal(R.cpu().numpy()[0], R0) and np.array_equal(t.cpu().numpy().reshape(3), t0)


def test_query_sdf_bit_exact(sdf, S):
    z = np.load(os.path.join(G, "sdf_query.npz"))
    for i in _cases(z, "q"):
        _, scale = z[f"q{i}_meta"]
        vol = z[f"q{i}_vol"]
        q, pen, idx = sdf.query_sdf(_d(z[f"q{i}_hand"]), _d(z[f"q{i}_obj_r"]), _d(z[f"q{i}_obj_t"]), _d(vol), float(scale),
                                    with_penetration=True, with_index=True)
        oi, osdf, open_ = S.nearest(z[f"q{i}_hand"], z[f"q{i}_obj_r"], z[f"q{i}_obj_t"], vol, float(scale))
        assert np.array_equal(idx.cpu().numpy(), oi)                                  # voxel index: bit-exact vs oracle
        assert np.array_equal(q.cpu().numpy().view(np.uint8), osdf.view(np.uint8))
        assert np.array_equal(pen.cpu().numpy().view(np.uint8), open_.view(np.uint8))
        # vs the vectors of the imported reference itself (gf_optimize_hand_pose.query_sdf / get_penetration_loss): exact
        assert np.array_equal(q.cpu().numpy(), z[f"q{i}_sdf_ref"])
        assert np.array_equal(pen.cpu().numpy(), z[f"q{i}_pen_ref"])
        only = sdf.query_sdf(_d(z[f"q{i}_hand"]), _d(z[f"q{i}_obj_r"]), _d(z[f"q{i}_obj_t"]), _d(vol), float(scale))
        assert torch.equal(only, q)


def test_full_size_properties(sdf, S):
    """Reference sizes: 201^3 fp16 volume, 2048 particles x 1024 points; 151^3, 5120 x 778."""
    res, stride = 201, 0.002
    vol = make_volume(res, stride, "box", np.float16)
    dvol = _d(vol)
    pc = object_points(42, 1024, "box")
    R0, t0 = random_pose(42)
    cam = (pc @ R0.T + t0).astype(np.float32)
    rot, tr = particles(43, 2048, R0, t0)
    e = sdf.particle_energy(_d(cam), _d(rot), _d(tr), dvol, stride).cpu().numpy()
    # (1) fused == unfused: transform on the host in the kernel's own chain, Distance kernel, mean
    sub = [0, 1, 777, 2047]
    np.testing.assert_allclose(e[sub], S.particle_energy(cam, rot[sub], tr[sub], vol, stride), rtol=2e-6, atol=1e-8)
    # (2) the true pose is (nearly) the best particle and its energy is at the sensor-noise level
    assert e[0] < 0.004 and e[0] <= np.percentile(e, 5)
    # (3) invariance: permuting the cloud changes only the summation order
    perm = np.random.default_rng(0).permutation(1024)
    e2 = sdf.particle_energy(_d(cam[perm]), _d(rot), _d(tr), dvol, stride).cpu().numpy()
    np.testing.assert_allclose(e, e2, rtol=3e-6, atol=1e-9)
    # (4) the optimiser recovers a jittered pose at full size
    dR, dt = random_pose(44, angle=0.05, trans=0.006)
    pre = np.random.default_rng(45).standard_normal((2048, 6)).astype(np.float32)
    pre[0] = 0
    Ri, ti = (R0 @ dR).astype(np.float32), (t0 + dt).astype(np.float32)
    R, t = sdf.obj_optimize(_d(cam), _d(Ri), _d(ti), _d(pre), dvol, stride)
    before = S.particle_energy(cam, Ri[None], ti[None], vol, stride)[0]
    after = S.particle_energy(cam, R.cpu().numpy(), t.cpu().numpy().reshape(1, 3), vol, stride)[0]
    assert after < 0.5 * before
    Rn = R.cpu().numpy()[0].astype(np.float64)
    assert np.allclose(Rn @ Rn.T, np.eye(3), atol=1e-6)
    # hand side
    res, scale = 151, 0.003
    vol = make_volume(res, scale, "capsule", np.float16)
    hand = hand_particles(46, 5120, 778, R0, t0, extent=0.26)
    q, pen, idx = sdf.query_sdf(_d(hand), _d(R0), _d(t0), _d(vol), scale, with_penetration=True, with_index=True)
    idx = idx.cpu().numpy()
    assert idx.min() >= 0 and idx.max() < res ** 3
    assert np.array_equal(vol[idx], q.cpu().numpy())                                  # the lookup is a pure copy
    qf = q.float()
    assert torch.equal(pen, torch.max(qf.abs() * (qf < 0), dim=-1)[0].half())        # get_penetration_loss, restated
    oi, _, _ = S.nearest(hand[:64], R0, t0, vol, scale)
    assert np.array_equal(idx[:64], oi)


def test_validation(sdf):
    vol = torch.zeros(27, dtype=torch.float16, device="cuda")
    with pytest.raises(RuntimeError, match="no CPU path"):
        sdf.distance(torch.zeros(4, 3, device="cuda"), torch.zeros(27, dtype=torch.float16), 0.1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        sdf.distance(torch.zeros(4, 3), vol, 0.1)
    with pytest.raises(ValueError):
        sdf.distance(torch.zeros(4, 3, device="cuda"), torch.zeros(28, dtype=torch.float16, device="cuda"), 0.1)
    with pytest.raises(TypeError):
        sdf.distance(torch.zeros(4, 3, device="cuda"), vol.double(), 0.1)
    with pytest.raises(Exception):
        sdf.query_sdf(torch.zeros(2, 4, 3, device="cuda"), torch.eye(3).cuda(), torch.zeros(3).cuda(),
                      torch.zeros(64, dtype=torch.float16, device="cuda"), 0.1)   # even res


def test_reference_class_surface(sdf):
    """network/models/optimization_obj.py / optimization_hand.py: same method names and results as the reference classes."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "network"))
    from models.optimization_hand import gf_optimize_hand_pose
    from models.optimization_obj import gf_optimize_obj

    z = np.load(os.path.join(G, "sdf_optimize.npz"))
    res, stride = z["o1_meta"]
    o = gf_optimize_obj({"device": "cuda"})
    o.load_volume(_d(z["o1_vol"]).view(int(res), int(res), int(res)), float(stride))
    o.pre_sampled_particle = _d(z["o1_pre"])
    ret = o.optimize(_d(z["o1_pcld"])[None], {"rotation": _d(z["o1_R_init"])[None], "translation": _d(z["o1_t_init"]).view(1, 3, 1)},
                     "cat", "file", {"w": [640], "h": [480]})
    np.testing.assert_allclose(ret["rotation"].cpu().numpy()[0], z["o1_R_ref"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(ret["translation"].cpu().numpy().reshape(3), z["o1_t_ref"], rtol=0, atol=2e-6)
    e, se = o.evaluate(_d(z["o1_pcld"])[None], ret["rotation"], ret["translation"])
    assert e.shape == (1,) and torch.allclose(e, se * 500)
    assert o.Distance(_d(z["o1_pcld"])).shape == (z["o1_pcld"].shape[0],)
    assert o.update_seach_size(se[0], torch.ones(1, 6, device="cuda")).shape == (1, 6)

    q = np.load(os.path.join(G, "sdf_query.npz"))
    res, scale = q["q0_meta"]
    h = gf_optimize_hand_pose({"device": "cuda"})
    h.load_volume(_d(q["q0_vol"]).view(int(res), int(res), int(res)), float(scale))
    h.set_obj_pose({"rotation": _d(q["q0_obj_r"]), "translation": _d(q["q0_obj_t"])})
    qs = h.query_sdf(_d(q["q0_hand"]))
    assert (qs.cpu().numpy() != q["q0_sdf_ref"]).mean() <= 1e-4
    pen = h.get_penetration_loss(qs)
    qs2, pen2 = h.query_sdf_and_penetration(_d(q["q0_hand"]))
    assert torch.equal(qs, qs2) and torch.equal(pen, pen2)


@pytest.mark.parametrize("dt", [np.float16, np.float32])
def test_corner_layout_is_bit_identical(sdf, S, dt):
    """pn2s_build_corner_volume: lookups through the corner layout == lookups through the linear volume, every bit,
    including the reference's flat-index wrap at the top faces and the clamp at the last element."""
    for res, stride in ((2, 0.2), (5, 0.08), (41, 0.01), (101, 0.004)):
        rng = np.random.default_rng(res)
        vol = rng.uniform(-0.1, 0.1, res ** 3).astype(dt)      # rough volume: any indexing slip shows
        dv = _d(vol)
        cv = sdf.CornerVolume(dv)
        ext = res * stride
        V = np.concatenate([rng.uniform(-0.2 - 0.2 * ext, -0.2 + 1.2 * ext, (50000, 3)),
                            -0.2 + rng.integers(0, res, (4000, 3)) * stride,
                            np.full((8, 3), -0.2 + (res - 1) * stride)]).astype(np.float32)
        a = sdf.distance(_d(V), dv, stride).cpu().numpy()
        b = sdf.distance(_d(V), cv, stride).cpu().numpy()
        assert np.array_equal(a.view(np.int32), b.view(np.int32))
        assert np.array_equal(a.view(np.int32), S.distance(V, vol, stride).view(np.int32))
        # the cells themselves: the eight values the reference indexes for every base index
        cells = cv.data.cpu().numpy()
        R, last = res, res ** 3 - 1
        i = np.arange(res ** 3)
        for k, off in enumerate((0, 1, R, 1 + R, R * R, 1 + R * R, R + R * R, 1 + R + R * R)):
            assert np.array_equal(cells[:, k], vol[np.minimum(i + off, last)])
    # fused entries accept it too and agree exactly with the linear path (same arithmetic, same reduction order)
    z = np.load(os.path.join(G, "sdf_optimize.npz"))
    _, stride = z["o0_meta"]
    dv = _d(z["o0_vol"].astype(dt))
    cv = sdf.CornerVolume(dv)
    args = (_d(z["o0_pcld"]), _d(z["o0_R_init"]), _d(z["o0_t_init"]), _d(z["o0_pre"]))
    Ra, ta = sdf.obj_optimize(*args, dv, float(stride))
    Rb, tb = sdf.obj_optimize(*args, cv, float(stride))
    assert torch.equal(Ra, Rb) and torch.equal(ta, tb)


def test_query_sdf_voxel_boundaries_exact(sdf, S):
    """Coordinates on and one ulp either side of every voxel face, huge and tiny values: the kernel's division-based
    floor must give the voxel torch's fmod-based `//` gives (oracle = literal restatement of c10::div_floor_floating)."""
    for res, scale in ((151, 0.003), (31, 0.015), (201, 0.002), (5, 0.1)):
        half = res // 2
        k = np.arange(-half - 3, half + 4, dtype=np.float64)
        base = (k * np.float64(np.float32(scale))).astype(np.float32)          # nearest floats to k*b
        alt = (k.astype(np.float32) * np.float32(scale)).astype(np.float32)    # fp32 products k*b
        c = np.concatenate([base, np.nextafter(base, np.float32(np.inf)), np.nextafter(base, np.float32(-np.inf)),
                            alt, np.nextafter(alt, np.float32(np.inf)), np.nextafter(alt, np.float32(-np.inf)),
                            np.float32([0.0, -0.0, 1e-38, -1e-38, 1e-45, -1e-45, 3e4, -3e4, 1e10, -1e10, 3.0e38, -3.0e38])]).astype(np.float32)
        rng = np.random.default_rng(res)
        pts = np.stack([c, rng.permutation(c), rng.permutation(c)], axis=-1)[None]   # (1, M, 3), identity pose: q == hand
        vol = rng.uniform(-0.1, 0.1, res ** 3).astype(np.float16)
        eye, zero = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
        _, idx = sdf.query_sdf(_d(pts), _d(eye), _d(zero), _d(vol), scale, with_index=True)
        oi, _, _ = S.nearest(pts, eye, zero, vol, scale)
        assert np.array_equal(idx.cpu().numpy(), oi), (res, scale)
        # and against torch's own floor division on the host
        want = (torch.clamp(torch.from_numpy(pts[0]) // scale, -half, half).long() + half)
        flat = (want[:, 0] * res + want[:, 1]) * res + want[:, 2]
        assert np.array_equal(idx.cpu().numpy()[0], flat.numpy().astype(np.int32))


def test_object_tracking_sequence(sdf):
    """gf_optimize_obj over a synthetic sequence the way the reference's tracker drives it (track_network.py:365: each
    frame starts from the previous frame's estimate): a box moving 4 mm / 1.4 degrees per frame for 20 frames must stay
    tracked to the sensor-noise level, without any host synchronisation inside optimize()."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "network"))
    from models.optimization_obj import gf_optimize_obj
    from _sdf_cases import _axis_angle

    res, stride = 201, 0.002
    o = gf_optimize_obj({"device": "cuda"}, seed=0)
    o.load_volume(_d(make_volume(res, stride, "box", np.float16)).view(res, res, res), stride)
    pts = object_points(3, 1024, "box")
    R, t = random_pose(3)
    R, t = R.astype(np.float64), t.astype(np.float64)
    est = {"rotation": _d(R.astype(np.float32))[None], "translation": _d(t.astype(np.float32)).view(1, 3, 1)}
    rng = np.random.default_rng(9)
    worst_t = worst_r = 0.0
    for frame in range(20):
        R = R @ _axis_angle(rng.standard_normal(3), 0.025)
        v = rng.standard_normal(3)
        t = t + 0.004 * v / np.linalg.norm(v)
        cam = (pts @ R.T + t + rng.normal(0, 0.0005, pts.shape)).astype(np.float32)
        est = o.optimize(_d(cam)[None], est, "box", "frame%d" % frame, {"w": [640], "h": [480]})
        Re = est["rotation"].cpu().numpy()[0].astype(np.float64)
        te = est["translation"].cpu().numpy().reshape(3).astype(np.float64)
        ang = np.arccos(np.clip((np.trace(Re.T @ R) - 1) / 2, -1, 1))
        # the box is symmetric only under 180-degree flips, far from the 1.4-degree steps: pose error is well defined
        worst_t, worst_r = max(worst_t, np.linalg.norm(te - t)), max(worst_r, ang)
    assert worst_t < 0.004 and worst_r < 0.05, (worst_t, worst_r)
