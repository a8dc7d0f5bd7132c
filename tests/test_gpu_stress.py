"""GPU, BASELINE.json configs[4] at its per-GPU size (64 clouds x N=8192, npoint=2048, nsample=64, C=64):
size-independent properties over the whole batch instead of a full oracle run, plus an oracle comparison on a
2-cloud slice (the oracle's FPS of one 8192 -> 2048 cloud takes ~0.1 s, the ball query ~0.2 s)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
B, N, S, K, C = 64, 8192, 2048, 64, 64


@pytest.fixture(scope="module")
def data():
    from hotrack_amd import pointnet2_utils as ops
    g = torch.Generator(device="cuda").manual_seed(0)
    xyz = torch.rand(B, N, 3, device="cuda", generator=g)
    feat = torch.randn(B, C, N, device="cuda", generator=g)
    fps = ops.furthest_point_sample(xyz, S)
    new_xyz = torch.gather(xyz, 1, fps.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    return ops, xyz, feat, fps, new_xyz


def test_fps_full_size_properties(data, oracle):
    ops, xyz, feat, fps, new_xyz = data
    f = fps.cpu().numpy()
    assert (f[:, 0] == 0).all() and f.min() >= 0 and f.max() < N
    for b in range(B):
        assert len(set(f[b].tolist())) == S  # distinct points are never re-selected while M <= N
    # greedy property: every selected point was, when chosen, at least as far from the chosen set as any other point
    x = xyz[0].cpu().numpy().astype(np.float64)
    chosen = x[f[0, :200]]
    d = np.full(N, np.inf)
    for j in range(199):
        d = np.minimum(d, ((x - chosen[j]) ** 2).sum(1))
        assert d[f[0, j + 1]] >= d.max() * (1 - 1e-6)
    ref = oracle.furthest_point_sample(xyz[:2].cpu().numpy(), S)  # exact, 2 clouds
    np.testing.assert_array_equal(f[:2], ref)


@pytest.mark.parametrize("radius", [0.2, 0.1])
def test_ball_query_full_size_properties(data, oracle, radius):
    ops, xyz, feat, fps, new_xyz = data
    idx = ops.ball_query(radius, K, xyz, new_xyz)
    i = idx.long()
    assert int(i.min()) >= 0 and int(i.max()) < N
    pts = torch.gather(xyz.unsqueeze(1).expand(-1, S, -1, -1), 2, i.unsqueeze(-1).expand(-1, -1, -1, 3))
    d2 = ((pts - new_xyz.unsqueeze(2)) ** 2).sum(-1)
    assert float(d2.max()) < radius * radius * (1 + 1e-5)  # every listed point is inside the ball (centroid itself always is)
    # rows are ascending until the first repeat of the first hit (padding), never descending before that
    diff = i[:, :, 1:] - i[:, :, :-1]
    pad = i[:, :, 1:] == i[:, :, :1]
    assert bool(((diff > 0) | pad).all())
    np.testing.assert_array_equal(idx[:2].cpu().numpy(), oracle.ball_query(radius, K, xyz[:2].cpu().numpy(), new_xyz[:2].cpu().numpy()))


def test_group_and_fused_sa_full_size(data):
    ops, xyz, feat, fps, new_xyz = data
    from hotrack_amd import ext
    idx = ops.ball_query(0.2, K, xyz, new_xyz)
    grouped = ops.grouping_operation(feat, idx)  # (B,C,S,K)
    ref = torch.gather(feat.unsqueeze(2).expand(-1, -1, S, -1), 3, idx.long().unsqueeze(1).expand(-1, C, -1, -1))
    assert torch.equal(grouped, ref)  # pure copy: bit exact at full size
    # linearity of the backward (adjoint identity) at full size: <group(f), g> == <f, group^T(g)>
    gsel = torch.randn_like(grouped)
    f2 = feat.clone().requires_grad_(True)
    (ops.grouping_operation(f2, idx) * gsel).sum().backward()
    lhs = float((grouped.double() * gsel.double()).sum())
    rhs = float((feat.double() * f2.grad.double()).sum())
    assert abs(lhs - rhs) < 5e-6 * max(1.0, abs(lhs))  # 5.4e8 fp32 products on each side, ~1000 adds per gradient element
    # fused SA scale [C+3 -> 64 -> 64 -> 128] vs unfused torch on the materialised group
    g = torch.Generator(device="cuda").manual_seed(1)
    W1 = torch.randn(64, C + 3, device="cuda", generator=g) * 0.2
    b1 = torch.randn(64, device="cuda", generator=g) * 0.1
    W2 = torch.randn(64, 64, device="cuda", generator=g) * 0.2
    b2 = torch.randn(64, device="cuda", generator=g) * 0.1
    W3 = torch.randn(128, 64, device="cuda", generator=g) * 0.2
    b3 = torch.randn(128, device="cuda", generator=g) * 0.1
    a1f = torch.matmul(feat.transpose(1, 2), W1[:, :C].t().contiguous())
    out = ext.sa_mlp_max(idx, W2, b2, W3, b3, a1f=a1f, xyz=xyz, cxyz=new_xyz, wx=W1[:, C:].contiguous(), b1=b1)
    gx = torch.gather(xyz.transpose(1, 2).unsqueeze(2).expand(-1, -1, S, -1), 3, idx.long().unsqueeze(1).expand(-1, 3, -1, -1))
    # 2 clouds through the unfused composition in fp64 (the ground truth: an fp32 einsum carries its own summation-order error),
    # held to the bound the small shapes are held to (tests/test_gpu_fused.py::test_fused_sa_scale_matches_unfused: 2e-5 of the
    # output scale; VERDICT r5: the full-size test allowed 1e-3 without a reason)
    x = torch.cat([grouped, gx - new_xyz.transpose(1, 2).unsqueeze(-1)], 1)[:2].double()
    d = lambda t: t.double()
    h = torch.relu(torch.einsum("oc,bcsk->bosk", d(W1), x) + d(b1)[None, :, None, None])
    h = torch.relu(torch.einsum("oc,bcsk->bosk", d(W2), h) + d(b2)[None, :, None, None])
    h = torch.relu(torch.einsum("oc,bcsk->bosk", d(W3), h) + d(b3)[None, :, None, None]).max(-1)[0]
    err, scale = float((out[:2].double() - h).abs().max()), float(h.abs().max())
    assert err <= 2e-5 * max(1.0, scale), (err, scale)
