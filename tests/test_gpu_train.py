"""GPU: BASELINE.json configs[2] -- one HandTrackNet TRAINING step (train-mode BatchNorm, losses, backward) through the
HIP forward AND backward operators, against the golden vectors of the imported reference's own train step
(tests/golden/handtracknet_reference.npz, make_golden.py:179-246): total loss, pred_kp, the grad-is-None mask
(30 tensors / 3,746,944 parameters never receive a gradient) and the per-parameter gradient norms of the reference's
fp64 re-run.  Also asserts that the C-ABI backward entry points really ran (no torch-indexing substitute)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "network"))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fast", [False, True])
def test_train_step_hip_operators_match_reference_golden(monkeypatch, fast):
    """fast=False: module path (channel-major, the ten reference operators forward + backward);
    fast=True: point-major training path (models/fast_train.py, hotrack_amd/train_ops.py) -- the default on the GPU."""
    import test_network as tn
    from models.hand_network import HandTrackNet
    monkeypatch.setattr(HandTrackNet, "_force_fast_train", fast, raising=False)
    from hotrack_amd import pointnet2_hip
    calls = {}
    for name in pointnet2_hip.EXPORTED:
        fn = getattr(pointnet2_hip, name)

        def counted(*a, _fn=fn, _name=name, **k):
            calls[_name] = calls.get(_name, 0) + 1
            return _fn(*a, **k)
        monkeypatch.setattr(pointnet2_hip, name, counted)
    model, ret, total = tn._train_step("cuda", True)
    gold = tn.GOLD
    assert abs(float(total) - float(gold["train_total_loss"])) < 2e-4 * abs(float(gold["train_total_loss"]))
    np.testing.assert_allclose(ret["pred_kp"].detach().cpu().numpy(), gold["train_pred_kp"], atol=5e-4)
    none_mask = np.array([p.grad is None for _, p in model.named_parameters()])
    np.testing.assert_array_equal(none_mask, gold["param_grad_is_none"])
    assert int(none_mask.sum()) == 30
    assert sum(p.numel() for p in model.parameters() if p.grad is None) == 3746944
    gn = np.array([0.0 if p.grad is None else float(p.grad.norm()) for _, p in model.named_parameters()])
    truth = gold["param_grad_norm_f64"]
    live = truth > 1e-6 * truth.max()
    np.testing.assert_allclose(gn[live], truth[live], rtol=1e-2, atol=5e-4)
    assert bool(model._ftrain) == fast
    # the HIP operators (forward and backward) carried the step
    names = ("furthest_point_sampling_wrapper", "ball_query_wrapper") if fast else (
        "furthest_point_sampling_wrapper", "ball_query_wrapper", "knn_wrapper", "three_nn_wrapper", "three_interpolate_wrapper",
        "three_interpolate_grad_wrapper", "group_points_wrapper", "group_points_grad_wrapper")
    for name in names:
        assert calls.get(name, 0) > 0, (name, calls)


def test_train_step_is_run_to_run_stable():
    """Two identical steps: same loss to fp32 round-off (the LDS-slab backward kernels use no global atomics)."""
    import test_network as tn
    a = float(tn._train_step("cuda", True)[2])
    b = float(tn._train_step("cuda", True)[2])
    assert abs(a - b) <= 1e-6 * abs(a)


# ---- the point-major training operators one by one ----------------------------------------------------------------------
@pytest.mark.parametrize("R,C", [(4096, 32), (1000, 64), (21 * 16 * 3, 128), (777, 192), (5000, 384), (333, 512), (7, 4)])
@pytest.mark.parametrize("relu", [True, False])
def test_bn_relu_matches_torch_batchnorm(R, C, relu):
    from hotrack_amd.train_ops import Workspace, bn_relu
    g = torch.Generator(device="cuda").manual_seed(R + C)
    y0 = torch.randn(R, C, device="cuda", generator=g) * 2 + 0.5
    bias = torch.randn(C, device="cuda", generator=g)
    go = torch.randn(R, C, device="cuda", generator=g)
    bn_a, bn_b = torch.nn.BatchNorm1d(C).cuda().train(), torch.nn.BatchNorm1d(C).cuda().train()
    with torch.no_grad():
        for bn in (bn_a, bn_b):
            bn.weight.copy_(1 + 0.3 * torch.randn(C, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)))
            bn.bias.copy_(0.2 * torch.randn(C, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2)))
            bn.momentum = 0.05
    ws = Workspace("cuda")
    ya = y0.clone().requires_grad_(True)
    ba = bias.clone().requires_grad_(True)
    ha = bn_relu(ya, bn_a, ws, ba, relu=relu)
    ha.backward(go)
    yb = y0.clone().requires_grad_(True)
    bb = bias.clone().requires_grad_(True)
    hb = bn_b(yb + bb)
    hb = torch.relu(hb) if relu else hb
    hb.backward(go)
    torch.testing.assert_close(ha, hb, rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(bn_a.running_mean, bn_b.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn_a.running_var, bn_b.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn_a.num_batches_tracked) == int(bn_b.num_batches_tracked) == 1
    scale = float(yb.grad.abs().max())
    torch.testing.assert_close(ya.grad, yb.grad, rtol=1e-4, atol=2e-5 * max(scale, 1.0))
    torch.testing.assert_close(bn_a.weight.grad, bn_b.weight.grad, rtol=1e-4, atol=1e-4 * float(bn_b.weight.grad.abs().max()))
    torch.testing.assert_close(bn_a.bias.grad, bn_b.bias.grad, rtol=1e-4, atol=1e-4 * float(bn_b.bias.grad.abs().max()))
    assert ba.grad is not None and float(ba.grad.abs().max()) == 0.0      # analytically zero
    assert float(bb.grad.abs().max()) < 1e-3 * float(bn_b.bias.grad.abs().max() + 1)   # torch: round-off only


@pytest.mark.parametrize("with_feat,with_centre,Ks", [(False, False, [32]), (True, False, [32]), (True, True, [16, 64]), (True, False, [16, 64])])
def test_sa_layer1_matches_grouped_reference(with_feat, with_centre, Ks, oracle):
    from hotrack_amd.train_ops import sa_layer1
    B, N, S, D, C1 = 3, 300, 21, 24, 32
    g = torch.Generator(device="cuda").manual_seed(5)
    xyz = torch.rand(B, N, 3, device="cuda", generator=g)
    cxyz = torch.rand(B, S, 3, device="cuda", generator=g)
    feat = torch.randn(B, N, D, device="cuda", generator=g)
    cfeat = torch.randn(B, S, D, device="cuda", generator=g)
    idxs = [torch.randint(0, N, (B, S, K), device="cuda", generator=g, dtype=torch.int32) for K in Ks]
    Cin = (D if with_feat else 0) + 3 + (D if with_centre else 0)
    Ws = [torch.randn(C1, Cin, device="cuda", generator=g).requires_grad_(True) for _ in Ks]
    Df = D if with_feat else 0

    def run(fused):
        for W in Ws:
            W.grad = None
        f = feat.clone().requires_grad_(True)
        cf = cfeat.clone().requires_grad_(True)
        if fused:
            a1f = torch.nn.functional.linear(f.view(B * N, D), torch.cat([W[:, :Df] for W in Ws])).view(B, N, -1) if with_feat else None
            cadd = torch.nn.functional.linear(cf.view(B * S, D), torch.cat([W[:, Df + 3:] for W in Ws])).view(B, S, -1) if with_centre else None
            outs = sa_layer1(a1f, cadd, xyz, cxyz, idxs, [W[:, Df:Df + 3] for W in Ws])
        else:
            outs = []
            for W, idx in zip(Ws, idxs):
                K = idx.shape[2]
                ii = idx.long().view(B, S * K)
                parts = []
                if with_feat:
                    parts.append(torch.gather(f, 1, ii.unsqueeze(-1).expand(-1, -1, D)))
                parts.append(torch.gather(xyz, 1, ii.unsqueeze(-1).expand(-1, -1, 3)) - cxyz.repeat_interleave(K, dim=1))
                if with_centre:
                    parts.append(cf.repeat_interleave(K, dim=1))
                outs.append(torch.nn.functional.linear(torch.cat(parts, dim=2), W))
        loss = sum((o * torch.cos(o.detach() * 0.1 + i)).sum() for i, o in enumerate(outs))
        loss.backward()
        return [o.detach() for o in outs], f.grad, cf.grad, [W.grad.clone() for W in Ws]

    oa, fa, ca, wa = run(True)
    ob, fb, cb, wb = run(False)
    for a, b in zip(oa, ob):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-4)
    if with_feat:
        torch.testing.assert_close(fa, fb, rtol=1e-4, atol=1e-3)
    if with_centre:
        torch.testing.assert_close(ca, cb, rtol=1e-4, atol=1e-3)
    for a, b in zip(wa, wb):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))


@pytest.mark.parametrize("C,weighted", [(64, False), (128, True), (256, False), (32, False), (192, False)])
def test_rows_segment_sum_with_long_skewed_segments(C, weighted):
    """The owner-computes row scatter (backward of the layer-1 gather / of three-NN interpolation) where some destinations
    receive hundreds of rows (ball-query padding repeats a ball's first index): the split-segment kernel (C / 4 dividing 256,
    >= 4 entries per destination on average) and the one-thread-per-(destination, quad) kernel vs index_add in fp64."""
    from hotrack_amd.train_ops import inverse_index, rows_segment_sum
    B, n_dst, M = 3, 100, 4096
    g = torch.Generator(device="cuda").manual_seed(C)
    t = 3 if weighted else 1
    idx = torch.randint(0, n_dst, (B, M * t), device="cuda", generator=g, dtype=torch.int32)
    idx[:, : M * t // 2] = idx[:, : M * t // 2] % 3          # half of all entries land on three destinations
    idx[0, :] = 7                                             # one cloud: a single destination receives everything
    dout = torch.randn(B, M, C, device="cuda", generator=g)
    w = torch.rand(B, M * t, device="cuda", generator=g) if weighted else None
    din = torch.full((B, n_dst, C), float("nan"), device="cuda")
    rows_segment_sum(dout, inverse_index(idx, n_dst), n_dst, din, weight=w)
    ref = torch.zeros(B, n_dst, C, device="cuda", dtype=torch.float64)
    src = dout.double().repeat_interleave(t, dim=1)
    if weighted:
        src = src * w.double().unsqueeze(-1)
    for b in range(B):
        ref[b].index_add_(0, idx[b].long(), src[b])
    torch.testing.assert_close(din.double(), ref, rtol=1e-5, atol=1e-3)


def test_interpolate_rows_matches_operator_api():
    from hotrack_amd import pointnet2_utils as ops
    from hotrack_amd.train_ops import interpolate_rows
    B, M, n, C = 3, 128, 500, 64
    g = torch.Generator(device="cuda").manual_seed(9)
    pts = torch.randn(B, M, C, device="cuda", generator=g)
    idx = torch.randint(0, M, (B, n, 3), device="cuda", generator=g, dtype=torch.int32)
    w = torch.rand(B, n, 3, device="cuda", generator=g)
    w = w / w.sum(-1, keepdim=True)
    go = torch.randn(B, n, C, device="cuda", generator=g)
    a = pts.clone().requires_grad_(True)
    oa = interpolate_rows(a, idx, w)
    oa.backward(go)
    b = pts.clone().requires_grad_(True)
    ob = ops.three_interpolate(b.transpose(1, 2).contiguous(), idx, w)  # channel-major operator (reference API)
    ob.backward(go.transpose(1, 2).contiguous())
    torch.testing.assert_close(oa, ob.transpose(1, 2), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(a.grad, b.grad, rtol=1e-4, atol=1e-4)


def test_fast_train_path_equals_module_path():
    """Same weights, same batch: point-major training path vs the channel-major module path.  The forward must agree
    tightly (loss, BatchNorm running statistics, counters).  Gradients of two fp32 implementations agree only to the
    per-cent level element-wise (train-mode BatchNorm chains amplify round-off), so each path is judged against the
    ELEMENT-WISE fp64 gradients of the imported reference's train step (golden `g64/*`, make_golden.py) and the
    point-major path must not be further from that truth than the module path."""
    import test_network as tn
    from models.hand_network import HandTrackNet
    res = {}
    for fast in (False, True):
        HandTrackNet._force_fast_train = fast
        try:
            model, ret, total = tn._train_step("cuda", True)
        finally:
            del HandTrackNet._force_fast_train
        assert bool(model._ftrain) == fast
        res[fast] = (model, float(total))
    (ma, la), (mb, lb) = res[False], res[True]
    assert abs(la - lb) < 2e-5 * abs(la), (la, lb)
    pa, pb = dict(ma.named_parameters()), dict(mb.named_parameters())
    err = {False: [], True: []}
    for k in pa:
        assert (pa[k].grad is None) == (pb[k].grad is None), k
        if pa[k].grad is None:
            continue
        t = torch.from_numpy(tn.GOLD["g64/" + k]).cuda()
        scale = float(t.abs().max())
        if scale < 1e-6:  # analytically zero (biases in front of a train-mode BatchNorm): round-off in A, exact zeros in B
            assert float(pb[k].grad.abs().max()) <= 1e-5, k
            continue
        for fast, p in ((False, pa[k]), (True, pb[k])):
            e = float((p.grad.flatten()[:1024].double() - t).abs().max()) / scale
            assert e < 0.25, (k, fast, e)  # worst single element of the noisiest (smallest) gradients; the mean is what is compared
            err[fast].append(e)
    mean = {f: sum(v) / len(v) for f, v in err.items()}
    assert mean[True] <= 1.25 * mean[False] + 1e-3, mean
    ba, bb = dict(ma.named_buffers()), dict(mb.named_buffers())
    for k in ba:
        if k.endswith("num_batches_tracked"):
            assert int(ba[k]) == int(bb[k]), k
        elif k.endswith("running_mean") or k.endswith("running_var"):
            # (atol 3e-5: the two paths canonicalise the cloud with differently rounded arithmetic -- torch matmul vs the
            # hand-frame kernel -- which moves near-zero running means of the last layers by ~1e-5)
            torch.testing.assert_close(bb[k], ba[k], rtol=1e-4, atol=3e-5, msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("C1,K", [(32, 32), (64, 32), (128, 16), (128, 64), (12, 7)])
def test_sa_layer1_statistics_in_the_same_launch(C1, K):
    """sa_layer1(ws=...): same outputs bit for bit, and the BatchNorm statistics it leaves in the workspace slice are what
    pn2x_bn_stats computes from the output (fp64 accumulators of fp32 partial sums: equal to round-off); mlp_stack then takes
    them instead of running its own pass (C1 = 12: channel quads that do not divide the workgroup -> the two-launch form inside)."""
    import ctypes
    from hotrack_amd import train_ops as T
    B, N, S = 5, 700, 37
    g = torch.Generator(device="cuda").manual_seed(C1 + K)
    xyz, cxyz = torch.rand(B, N, 3, device="cuda", generator=g), torch.rand(B, S, 3, device="cuda", generator=g)
    a1f, cadd = torch.randn(B, N, C1, device="cuda", generator=g), torch.randn(B, S, C1, device="cuda", generator=g)
    idx = torch.randint(0, N, (B, S, K), device="cuda", generator=g, dtype=torch.int32)
    wx = torch.randn(C1, 3, device="cuda", generator=g)
    plain, = T.sa_layer1(a1f, cadd, xyz, cxyz, [idx], [wx])
    ws, aux = T.Workspace("cuda"), {}
    fused, = T.sa_layer1(a1f, cadd, xyz, cxyz, [idx], [wx], aux=aux, ws=ws)
    assert torch.equal(plain, fused) and set(aux["sums"]) == {0}
    ref = torch.zeros(T._lib.pn2x_bn_sums_doubles(C1), dtype=torch.float64, device="cuda")
    T._native._check(T._lib.pn2x_bn_stats(B * S * K, C1, plain.data_ptr(), C1, ref.data_ptr(), T._native._stream(plain)), "bn_stats")
    rep = ref.numel() // (2 * C1)
    tot = lambda t: t.view(rep, 2, C1).sum(0)
    y = plain.double().view(-1, C1)
    exact = torch.stack([y.sum(0), (y * y).sum(0)])
    scale = torch.stack([y.abs().sum(0), (y * y).sum(0)])  # fp32 partial sums of ~10 rows each, then fp64: ~1e-7 of the absolute sums
    for got in (tot(aux["sums"][0]), tot(ref)):
        assert float(((got - exact).abs() / scale).max()) < 2e-6, float(((got - exact).abs() / scale).max())


def test_gather_rows_gradient_is_the_segment_sum():
    """train_ops.gather_rows == torch.index_select along the rows, forward (bit-exact) and backward (a repeated row receives the
    sum of its slots' gradients, an unused row exact zeros), incl. the static permutation of the rearrange modules."""
    from hotrack_amd.train_ops import gather_rows, inverse_index
    g = torch.Generator(device="cuda").manual_seed(4)
    for B, N, M, C in ((32, 21, 63, 384), (3, 1024, 336, 128), (2, 7, 40, 4), (1, 5, 1, 8)):
        src = torch.randn(B, N, C, device="cuda", generator=g)
        idx = torch.randint(0, N, (B, M), device="cuda", generator=g, dtype=torch.int32)
        if N > 3:
            idx[idx == 2] = 3  # row 2 is never read
        wide = torch.randn(B, M, C + 8, device="cuda", generator=g)
        go = wide[:, :, 4:4 + C]  # a column block of a wider gradient (half of a concatenation's): read in place, no copy launch
        a = src.clone().requires_grad_(True)
        out = gather_rows(a, idx, inverse_index(idx, N))
        out.backward(go)
        b = src.clone().double().requires_grad_(True)
        ref = torch.gather(b, 1, idx.long().unsqueeze(-1).expand(B, M, C))
        ref.backward(go.double())
        assert torch.equal(out.detach().double(), ref.detach())
        assert torch.allclose(a.grad.double(), b.grad, rtol=1e-6, atol=1e-6)
        if N > 3:
            assert float(a.grad[:, 2].abs().max()) == 0.0


@pytest.mark.parametrize("G,K,C,ties", [(300, 32, 64, False), (21 * 5, 16, 192, False), (64, 128, 512, False), (7, 3, 4, False),
                                         (32 * 21, 64, 192, True), (32, 128, 512, True), (9000, 32, 64, True), (50, 24, 128, True)])
def test_bn_relu_max_matches_unfused(G, K, C, ties):
    """Fused BatchNorm + ReLU + max over K rows (forward, arg-max routing of the gradient, running statistics) vs
    bn_relu followed by torch.max -- through the one-thread-per-(group, quad) kernel (many groups) and the one that splits a
    group's rows over lanes (few groups); `ties`: pre-activations on a coarse lattice, so equal maxima are common and the
    FIRST row must win in both."""
    from hotrack_amd.train_ops import Workspace, bn_relu, bn_relu_max
    g = torch.Generator(device="cuda").manual_seed(G + K)
    y0 = torch.randn(G * K, C, device="cuda", generator=g) * 1.5 - 0.2
    if ties:
        y0 = torch.round(y0 * 2) / 2
    go = torch.randn(G, C, device="cuda", generator=g)
    bias = torch.randn(C, device="cuda", generator=g)
    res = []
    for fused in (True, False):
        bn = torch.nn.BatchNorm1d(C).cuda().train()
        with torch.no_grad():
            bn.weight.copy_(1 + 0.3 * torch.randn(C, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)))
            bn.bias.copy_(0.2 * torch.randn(C, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2)))
        ws = Workspace("cuda")
        y = y0.clone().requires_grad_(True)
        b = bias.clone().requires_grad_(True)
        out = bn_relu_max(y, K, bn, ws, b) if fused else bn_relu(y, bn, ws, b).view(G, K, C).max(dim=1)[0]
        out.backward(go)
        res.append((out.detach(), y.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone()))
    a, b_ = res
    assert torch.equal(a[0], b_[0])
    torch.testing.assert_close(a[1], b_[1], rtol=1e-5, atol=1e-6)   # same arg-max routing (ties: both take the first row)
    torch.testing.assert_close(a[2], b_[2], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(a[3], b_[3], rtol=1e-5, atol=1e-5)
    assert torch.equal(a[4], b_[4]) and torch.equal(a[5], b_[5])


def test_graph_captured_data_parallel_step_two_ranks_one_gpu():
    """Trainer's "flat" data-parallel mode under torch.distributed.run: forward+backward HIP graph | one all-reduce of the flat
    gradient buffer | optimiser HIP graph.  Two ranks share this GPU over gloo (RCCL refuses two ranks on one device; the
    collective call site is the same), launched the way the driver launches bench.py."""
    import json
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PN2_DIST_BACKEND="gloo", HOTRACK_DATA_ROOT="/tmp/hotrack_test_dp")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "scripts", "bench_train.py"), "--steps", "6", "--warmup", "4",
                          "--batch", "8", "--graph"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, out.stdout[-2000:]
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["graph_step"] is True and d["dp_mode"] == "flat" and d["loss"] == d["loss"]


# ---- fused BatchNorm GEMM stacks (csrc/train_gemm.hip, hotrack_amd/train_stack.py) ---------------------------------------------
def _stack_draw(R, widths, K, seed):
    """Inputs, modules and the fp64 reference of one seeded draw of a stack case, plus the reference's KINK MARGIN: the smallest
    |pre-activation| in front of any ReLU.  An element whose fp32 pre-activation has the other sign than the fp64 one flips one
    mask bit, and the gradient of a ReLU is discontinuous there: one flipped bit moves a row of dW by ~ 1 / sqrt(R) (percents)
    and a row of dY_1 by as much -- a property of comparing fp32 with fp64, not of the kernel under test."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    C1 = widths[0]
    y1 = (torch.randn(R, C1, device="cuda", generator=g) * 1.5 + 0.3)
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)  # Conv1d draws its weights from the global CPU generator
        convs = [torch.nn.Conv1d(a, b, 1).cuda() for a, b in zip(widths[:-1], widths[1:])]
    bns = [torch.nn.BatchNorm1d(c).cuda().train() for c in widths]
    bias1 = torch.randn(C1, device="cuda", generator=g).requires_grad_(True)
    with torch.no_grad():
        for i, bn in enumerate(bns):
            bn.weight.copy_(1 + 0.3 * torch.randn(bn.weight.shape, device="cuda", generator=g))
            bn.bias.copy_(0.2 * torch.randn(bn.bias.shape, device="cuda", generator=g))
            bn.momentum = 0.07
    ref_convs = [torch.nn.Conv1d(a, b, 1).cuda().double() for a, b in zip(widths[:-1], widths[1:])]
    ref_bns = [torch.nn.BatchNorm1d(c).cuda().double().train() for c in widths]
    for c, rc in zip(convs, ref_convs):
        rc.load_state_dict({k: v.double() for k, v in c.state_dict().items()})
    for b, rb in zip(bns, ref_bns):
        rb.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in b.state_dict().items()})
        rb.momentum = b.momentum
    # reference forward (fp64 torch modules; the layer-1 bias enters in front of its BatchNorm as in the network)
    yb = y1.double().clone().requires_grad_(True)
    b1 = bias1.detach().double().clone().requires_grad_(True)
    pre = ref_bns[0](yb + b1)
    margin = float(pre.detach().abs().min())
    h = torch.relu(pre)
    for rc, rb in zip(ref_convs, ref_bns[1:]):
        pre = rb(rc(h.t().unsqueeze(0)).squeeze(0).t())
        margin = min(margin, float(pre.detach().abs().min()))
        h = torch.relu(pre)
    ref = h.view(R // K, K, -1).max(dim=1)[0] if K else h
    return g, y1, convs, bns, bias1, ref_convs, ref_bns, yb, ref, margin


KINK_MARGIN = 2e-6  # ~ 5x the fp32 round-off of a BatchNorm output of O(1) behind a 128-term dot product


@pytest.mark.parametrize("R,widths,K", [
    (32 * 64, [32, 32, 64], 32),        # sa1 widths: 32-column tiles, wgrad with four row groups per workgroup
    (4000, [64, 64, 128], 0),           # ragged last row tile (4000 = 31 * 128 + 32), dense top
    (21 * 16 * 5, [128, 128, 192], 16), # keypoint-query widths: 192 = three 64-column tiles, max over K = 16
    (21 * 64 * 2, [128, 128, 192], 64),
    (1500, [128, 128, 512], 0),         # four 128-column tiles
    (999, [256, 256], 0),               # two-layer stack (feature propagation), K = 256: eight reduction chunks
    (2048, [128, 128, 384], 0),         # fp1 + conv1
    (128 * 3, [128, 128, 512], 128),    # sa3: max over all 128 points of a cloud
])
def test_mlp_stack_matches_torch(R, widths, K, seed_offset=0):
    """train_stack.mlp_stack (fused BatchNorm GEMMs) vs the same stack as torch modules in fp64: forward, running statistics,
    and every gradient (dY_1, weights, gamma / beta; conv biases get exact zeros).

    Every input is seeded, and the draw is the first one whose fp64 reference keeps all ~10^6 ReLU inputs at least KINK_MARGIN
    away from zero (decided from the reference alone, see _stack_draw): with unseeded conv weights this test used to fail about
    once in 12 suite runs on a single flipped mask bit (scripts/probes/stack_flake.py, stack_determinism.py: the fused path
    itself is bit-for-bit repeatable)."""
    from hotrack_amd import train_stack
    from hotrack_amd.train_ops import Workspace
    C1 = widths[0]
    for attempt in range(400):
        seed = R + sum(widths) + K + 7919 * seed_offset + 104729 * attempt
        g, y1, convs, bns, bias1, ref_convs, ref_bns, yb, ref, margin = _stack_draw(R, widths, K, seed)
        if margin >= KINK_MARGIN:
            break
    else:
        raise AssertionError("no draw with a kink margin >= %g" % KINK_MARGIN)
    assert train_stack.stack_supported(C1, widths[1:])
    import copy
    convs_c, bns_c = copy.deepcopy(convs), copy.deepcopy(bns)  # for the round-2 baseline below
    # fused
    ws = Workspace("cuda")
    ya = y1.clone().requires_grad_(True)
    layers = [train_stack.Layer(None, bns[0], bias1)] + [train_stack.Layer(c.weight.view(c.weight.shape[0], -1), bn, c.bias)
                                                         for c, bn in zip(convs, bns[1:])]
    out = train_stack.mlp_stack(ya, layers, ws, max_over=K)
    go = torch.randn(out.shape, device="cuda", generator=g)
    out.backward(go)
    ref.backward(go.double())
    torch.testing.assert_close(out.double(), ref, rtol=2e-4, atol=2e-4)
    for b, rb in zip(bns, ref_bns):
        torch.testing.assert_close(b.running_mean.double(), rb.running_mean, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(b.running_var.double(), rb.running_var, rtol=1e-4, atol=1e-5)
        assert int(b.num_batches_tracked) == 1

    # the same stack through the round-2 kernels (library GEMMs + streaming BatchNorm): train-mode BatchNorm chains amplify
    # fp32 round-off, so the bar for the fused path is "as close to fp64 as the unfused fp32 path", not an absolute number
    from hotrack_amd.train_ops import bn_relu, bn_relu_max
    yc = y1.clone().requires_grad_(True)
    x = bn_relu(yc, bns_c[0], ws, bias1.detach().clone().requires_grad_(True))
    for i, (c, bn) in enumerate(zip(convs_c, bns_c[1:])):
        yy = torch.nn.functional.linear(x, c.weight.view(c.weight.shape[0], -1))
        x = bn_relu_max(yy, K, bn, ws, c.bias) if (K and i == len(convs) - 1) else bn_relu(yy, bn, ws, c.bias)
    x.backward(go)
    base = float((yc.grad.double() - yb.grad).abs().max()) / (float(yb.grad.abs().max()) + 1e-12)

    def close(a, b, what, tol=2e-3):
        scale = float(b.abs().max()) + 1e-12
        err = float((a.double() - b).abs().max()) / scale
        assert err < tol, (what, err, tol)

    close(ya.grad, yb.grad, "dy1", max(2e-3, 3 * base))
    for i, (c, rc) in enumerate(zip(convs, ref_convs)):
        close(c.weight.grad, rc.weight.grad, f"dW{i + 2}")
        assert c.bias.grad is not None and float(c.bias.grad.abs().max()) == 0.0  # bias in front of a BatchNorm: exactly zero
    for i, (b, rb) in enumerate(zip(bns, ref_bns)):
        close(b.weight.grad, rb.weight.grad, f"dgamma{i + 1}")
        close(b.bias.grad, rb.bias.grad, f"dbeta{i + 1}")
    assert bias1.grad is not None and float(bias1.grad.abs().max()) == 0.0


@pytest.mark.parametrize("R,widths,K", [
    (32 * 256 * 32, [32, 32, 64], 32),      # sa1 at configs[2]'s per-GPU size: 262144 rows
    (32 * 128 * 32, [64, 64, 128], 32),     # sa2: 131072 rows
    (32 * 21 * 64, [128, 128, 192], 64),    # a keypoint-query scale, K = 64: 43008 rows
    (32 * 21 * 16, [128, 128, 192], 16),    # ... K = 16: 10752 rows
    (32 * 1024, [128, 128, 384], 0),        # fp1 + conv1: 32768 rows
])
def test_mlp_stack_matches_fp64_at_the_per_gpu_sizes(R, widths, K):
    """VERDICT r5 (weak 1.ii): the whole-step test at 32 x 1024 has to use per-cent bounds (BatchNorm chains amplify round-off through
    the network), so a 1 % systematic error in ONE fused stack could pass it.  Here every fused stack shape of the step, at the row
    count the step runs it with, alone against the same stack as fp64 torch modules.  At 10^7 - 10^8 ReLU inputs some fp32
    pre-activations fall on the other side of zero than their fp64 twins (each flips one mask bit: a property of comparing fp32
    with fp64), so the comparison is in the L2 norm: ~25 flipped bits among sa2's 3.3e7 ReLU inputs move a gradient by
    sqrt(25 / (128 x 131072)) ~ 1e-3 of its norm (measured 1.0 - 1.2e-3 on every tensor of that case); a systematic 1 % error in a
    stack moves it by 1e-2.  Bound: 4e-3."""
    from hotrack_amd import train_stack
    from hotrack_amd.train_ops import Workspace
    g, y1, convs, bns, bias1, ref_convs, ref_bns, yb, ref, _margin = _stack_draw(R, widths, K, 1234 + R + K)
    assert train_stack.stack_supported(widths[0], widths[1:])
    ws = Workspace("cuda")
    ya = y1.clone().requires_grad_(True)
    layers = [train_stack.Layer(None, bns[0], bias1)] + [train_stack.Layer(c.weight.view(c.weight.shape[0], -1), bn, c.bias)
                                                         for c, bn in zip(convs, bns[1:])]
    out = train_stack.mlp_stack(ya, layers, ws, max_over=K)
    go = torch.randn(out.shape, device="cuda", generator=g)
    out.backward(go)
    ref.backward(go.double())

    def rel(a, b):
        return float((a.detach().double() - b.detach()).norm()) / (float(b.detach().norm()) + 1e-30)

    assert rel(out, ref) < 2e-5, ("forward", rel(out, ref))
    errs = {"dy1": rel(ya.grad, yb.grad)}
    for i, (c, rc) in enumerate(zip(convs, ref_convs)):
        errs[f"dW{i + 2}"] = rel(c.weight.grad, rc.weight.grad)
        assert float(c.bias.grad.abs().max()) == 0.0
    for i, (b, rb) in enumerate(zip(bns, ref_bns)):
        errs[f"dgamma{i + 1}"] = rel(b.weight.grad, rb.weight.grad)
        errs[f"dbeta{i + 1}"] = rel(b.bias.grad, rb.bias.grad)
        torch.testing.assert_close(b.running_mean.double(), rb.running_mean, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(b.running_var.double(), rb.running_var, rtol=1e-4, atol=1e-5)
    assert max(errs.values()) < 4e-3, errs


def test_mlp_stack_second_backward_and_stale_workspace():
    """the fp64 backward accumulators are this forward's workspace slices only for its FIRST backward and only
    while no later forward reset the workspace; otherwise fresh zeros are used -- gradients never double-count."""
    from hotrack_amd import train_stack
    from hotrack_amd.train_ops import Workspace
    g = torch.Generator(device="cuda").manual_seed(3)
    R, widths = 1024, [32, 32, 64]
    convs = [torch.nn.Conv1d(a, b, 1).cuda() for a, b in zip(widths[:-1], widths[1:])]
    bns = [torch.nn.BatchNorm1d(c).cuda().train() for c in widths]
    ws = Workspace("cuda")

    def run(y):
        layers = [train_stack.Layer(None, bns[0])] + [train_stack.Layer(c.weight.view(c.weight.shape[0], -1), bn, c.bias)
                                                     for c, bn in zip(convs, bns[1:])]
        return train_stack.mlp_stack(y, layers, ws, max_over=32)

    y = torch.randn(R, 32, device="cuda", generator=g).requires_grad_(True)
    ws.reset()
    out = run(y)
    go = torch.randn(out.shape, device="cuda", generator=g)
    (g1,) = torch.autograd.grad(out, y, go, retain_graph=True)
    (g2,) = torch.autograd.grad(out, y, go, retain_graph=True)   # second backward through the same graph
    torch.testing.assert_close(g1, g2, rtol=1e-5, atol=1e-6)
    ws.reset()
    out_b = run(y.detach().clone().requires_grad_(True))           # a later forward resets the workspace ...
    (g3,) = torch.autograd.grad(out, y, go)                        # ... and the first graph's backward still gives the same
    torch.testing.assert_close(g1, g3, rtol=1e-5, atol=1e-6)
    assert out_b.shape == out.shape


@pytest.mark.parametrize("R,widths,K", [
    (16 * 4403, [64, 64, 128], 16),      # 1101 row tiles (the last one ragged) on <= 512 persistent workgroups
    (32 * 1200, [32, 32, 64], 32),
    (16 * 2000 + 16, [128, 128, 192], 16),
    (8 * 777, [128, 128, 128, 64, 32], 8),
    (24 * 301, [64, 64, 128], 24),       # groups of 24 rows (not a power of two) straddle the 64-row tiles
    (64 * 150, [128, 128, 192], 64),
    (5000, [128, 128, 384], 0),          # dense top (mask recomputed on load), 384 = two column slices of 192
    (128 * 40, [128, 128, 512], 128),    # routed top over four column slices of 128
    (3000, [64, 64, 128], 0),
])
def test_mlp_stack_one_kernel_layer_backward_equals_two_kernel_backward(R, widths, K):
    """csrc/train_bwd.hip (data + weight gradient of a layer from one pass) vs the tg_dgrad / tg_wgrad pair on the same
    forward: same formulas, different summation orders."""
    from hotrack_amd import train_stack
    from hotrack_amd.train_ops import Workspace
    g = torch.Generator(device="cuda").manual_seed(R)
    convs = [torch.nn.Conv1d(a, b, 1).cuda() for a, b in zip(widths[:-1], widths[1:])]
    bns = [torch.nn.BatchNorm1d(c).cuda().train() for c in widths]
    with torch.no_grad():
        for bn in bns:
            bn.weight.copy_(1 + 0.3 * torch.randn(bn.weight.shape, device="cuda", generator=g))
            bn.bias.copy_(0.2 * torch.randn(bn.bias.shape, device="cuda", generator=g))
    params = [p for m in convs + bns for p in m.parameters()]
    y1 = torch.randn(R, widths[0], device="cuda", generator=g) * 1.3 + 0.2
    go = torch.randn(R // K if K else R, widths[-1], device="cuda", generator=g)
    ws = Workspace("cuda")
    for a, b in zip(widths[:-1], widths[1:]):
        assert train_stack._bwd_slices(a, b)

    def run(fused):
        old = train_stack.FUSED_BWD
        train_stack.FUSED_BWD = fused
        try:
            for p in params:
                p.grad = None
            ws.reset()
            y = y1.clone().requires_grad_(True)
            layers = [train_stack.Layer(None, bns[0])] + [train_stack.Layer(c.weight, bn, c.bias) for c, bn in zip(convs, bns[1:])]
            train_stack.mlp_stack(y, layers, ws, max_over=K).backward(go)
            return [y.grad.clone()] + [None if p.grad is None else p.grad.clone() for p in params]
        finally:
            train_stack.FUSED_BWD = old

    one, two = run(True), run(False)
    for a, b in zip(one, two):
        assert (a is None) == (b is None)
        if a is not None:
            scale = max(1.0, float(b.abs().max()))
            torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5 * scale)


@pytest.mark.parametrize("R,widths,K", [
    (16 * 4403, [64, 64, 128], 16),       # C_{i-1} = 64: four 16-column slices x two row halves; routed top, ragged last tile
    (64 * 150 + 64, [128, 128, 192], 64),  # C_{i-1} = 128: eight slices, C_i = 192 (48 registers of W_i per lane)
    (5000, [128, 128, 384], 0),           # dense top, two column slices of 192 chained through the partial data gradient
    (128 * 40, [128, 128, 512], 128),     # routed top over four column slices of 128
    (3000, [64, 64, 64], 0),
    (21 * 16 * 7, [128, 128, 192], 16),   # one tile per workgroup and fewer tiles than compute units
])
def test_layer_backward_with_register_resident_w_equals_round4_kernel(R, widths, K):
    """Round 5: tg_bwd2 (W_i register-resident, 16x16x4 data-gradient tiles; csrc/train_bwd.hip) against the round-4 kernel on the
    same forward: the same dY / mask / sums expressions, another summation order inside the data gradient only."""
    from hotrack_amd import train_stack
    from hotrack_amd.train_ops import Workspace
    g = torch.Generator(device="cuda").manual_seed(R + 1)
    convs = [torch.nn.Conv1d(a, b, 1).cuda() for a, b in zip(widths[:-1], widths[1:])]
    bns = [torch.nn.BatchNorm1d(c).cuda().train() for c in widths]
    with torch.no_grad():
        for bn in bns:
            bn.weight.copy_(1 + 0.3 * torch.randn(bn.weight.shape, device="cuda", generator=g))
            bn.bias.copy_(0.2 * torch.randn(bn.bias.shape, device="cuda", generator=g))
    params = [p for m in convs + bns for p in m.parameters()]
    y1 = torch.randn(R, widths[0], device="cuda", generator=g) * 1.3 + 0.2
    go = torch.randn(R // K if K else R, widths[-1], device="cuda", generator=g)
    ws = Workspace("cuda")

    def run(v2):
        train_stack.set_bwd_kernel_variant(v2)
        try:
            for p in params:
                p.grad = None
            ws.reset()
            y = y1.clone().requires_grad_(True)
            layers = [train_stack.Layer(None, bns[0])] + [train_stack.Layer(c.weight, bn, c.bias) for c, bn in zip(convs, bns[1:])]
            train_stack.mlp_stack(y, layers, ws, max_over=K).backward(go)
            torch.cuda.synchronize()
            return [y.grad.clone()] + [None if p.grad is None else p.grad.clone() for p in params]
        finally:
            train_stack.set_bwd_kernel_variant(True)

    new, old = run(True), run(False)
    for a, b in zip(new, old):
        assert (a is None) == (b is None)
        if a is not None:
            scale = max(1.0, float(b.abs().max()))
            torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5 * scale)


def test_mlp_stack_deferred_weight_gradient_sums():
    """The weight-gradient reductions of all stacks run as ONE launch at the end of the autograd pass (train_stack._defer):
    same gradients as the immediate reductions; more layers in a pass than one kernel-argument pack holds; a second pass
    that ACCUMULATES into existing .grad falls back to immediate reductions (autograd would read the tensors too early)."""
    from hotrack_amd import train_stack
    from hotrack_amd.train_ops import Workspace
    g = torch.Generator(device="cuda").manual_seed(5)
    R, widths, n_stacks = 2048, [32, 64, 32, 64], 10          # 30 fused layers in one pass (> kRmMax = 24)
    stacks = [([torch.nn.Conv1d(a, b, 1).cuda() for a, b in zip(widths[:-1], widths[1:])],
               [torch.nn.BatchNorm1d(c).cuda().train() for c in widths]) for _ in range(n_stacks)]
    params = [p for convs, bns in stacks for m in convs + bns for p in m.parameters()]
    ws = Workspace("cuda")
    ys = [torch.randn(R, 32, device="cuda", generator=g) for _ in range(n_stacks)]
    gos = [torch.randn(R // 16, 64, device="cuda", generator=g) for _ in range(n_stacks)]

    def run_pass():
        ws.reset()
        total = 0
        for (convs, bns), y, go in zip(stacks, ys, gos):
            layers = [train_stack.Layer(None, bns[0])] + [train_stack.Layer(c.weight, bn, c.bias) for c, bn in zip(convs, bns[1:])]
            total = total + (train_stack.mlp_stack(y, layers, ws, max_over=16) * go).sum()
        total.backward()

    def grads():
        return [None if p.grad is None else p.grad.clone() for p in params]

    def zero():
        for p in params:
            p.grad = None

    old = train_stack.DEFER_REDUCE
    try:
        train_stack.DEFER_REDUCE = False
        zero(); run_pass(); ref = grads()
        train_stack.DEFER_REDUCE = True
        seen = []
        flush = train_stack._flush_reductions
        train_stack._flush_reductions = lambda: (seen.append(len(train_stack._pending)), flush())
        try:
            zero(); run_pass(); got = grads()
            assert seen == [n_stacks * (len(widths) - 1)] and not train_stack._pending   # ONE flush over all 30 layers
            for a, b in zip(got, ref):
                assert (a is None) == (b is None)
                if a is not None:   # atomics: the order of the partial sums is free
                    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * max(1.0, float(b.abs().max())))
            run_pass()                                  # accumulate a second pass into the existing gradients: nothing deferred
            assert len(seen) == 1
            for p, b in zip(params, ref):
                if b is not None:
                    torch.testing.assert_close(p.grad, 2 * b, rtol=1e-4, atol=2e-5 * max(1.0, float(b.abs().max())))
        finally:
            train_stack._flush_reductions = flush
        # ---- a parameter that feeds TWO stack nodes of one pass (weight tying / a module called twice before one
        # backward()): autograd sums the two gradients when the second node returns, so the first node's pending reduction must
        # have run by then; and parameters with tensor hooks are never deferred (the hook would see an unreduced tensor)
        convs, bns = stacks[0]
        shared = [train_stack.Layer(None, bns[0])] + [train_stack.Layer(c.weight, bn, c.bias) for c, bn in zip(convs, bns[1:])]
        tied = [p_ for m in convs + bns for p_ in m.parameters()]

        def run_tied():
            ws.reset()
            for p_ in tied:
                p_.grad = None
            a = train_stack.mlp_stack(ys[0], shared, ws, max_over=16)
            b = train_stack.mlp_stack(ys[1], shared, ws, max_over=16)
            ((a * gos[0]).sum() + (b * gos[1]).sum()).backward()
            return [None if p_.grad is None else p_.grad.clone() for p_ in tied]

        train_stack.DEFER_REDUCE = False
        ref_tied = run_tied()
        train_stack.DEFER_REDUCE = True
        for _ in range(3):  # unreduced tiles are uninitialised memory: repeat so that a stale-but-equal buffer cannot pass
            got_tied = run_tied()
            assert not train_stack._pending and not train_stack._pending_params
            for a, b in zip(got_tied, ref_tied):
                assert (a is None) == (b is None)
                if a is not None:
                    torch.testing.assert_close(a, b, rtol=1e-4, atol=2e-5 * max(1.0, float(b.abs().max())))
        seen_in_hook = []
        h = convs[0].weight.register_hook(lambda gr: seen_in_hook.append(gr.clone()))
        try:
            zero(); run_pass()
        finally:
            h.remove()
        torch.testing.assert_close(seen_in_hook[0], ref[params.index(convs[0].weight)], rtol=1e-4,
                                   atol=1e-5 * max(1.0, float(seen_in_hook[0].abs().max())))
    finally:
        train_stack.DEFER_REDUCE = old


# ---- multi-tensor Adam (csrc/adam.hip, hotrack_amd/optim.py) ---------------------------------------------------------------------
@pytest.mark.parametrize("wd", [0.0, 1e-4])
def test_fused_adam_matches_torch_adam(wd):
    """hotrack_amd.optim.FusedAdam vs torch.optim.Adam over 7 steps: 150 tensors (three kernel-argument packs), sizes that
    are not multiples of 4 / of the 8192-element chunk, parameters that never receive a gradient, weight decay; and the
    state dicts are interchangeable."""
    from hotrack_amd.optim import FusedAdam
    g = torch.Generator(device="cuda").manual_seed(11)
    sizes = [(1,), (3,), (5, 7), (8192,), (8193,), (100000,), (384, 256), (17, 3, 3)] + [(33 + i,) for i in range(142)]
    pa = [torch.randn(*s, device="cuda", generator=g).requires_grad_(True) for s in sizes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa = FusedAdam(pa, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    ob = torch.optim.Adam(pb, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    for it in range(7):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i % 10 == 9:   # never used: grad stays None, parameter and state untouched
                continue
            gr = torch.randn(a.shape, device="cuda", generator=g) * (1 + it)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    for i, (a, b) in enumerate(zip(pa, pb)):
        torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-7, msg=lambda m: f"tensor {i} {tuple(a.shape)}: {m}")
        if i % 10 == 9:
            assert a not in oa.state or len(oa.state[a]) == 0
        else:
            assert float(oa.state[a]["step"]) == 7.0
            torch.testing.assert_close(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"], rtol=2e-6, atol=1e-12)
    # torch's optimiser continues from our state dict and vice versa
    oc = torch.optim.Adam([p.detach().clone().requires_grad_(True) for p in pa], lr=1e-3, weight_decay=wd)
    oc.load_state_dict(oa.state_dict())
    od = FusedAdam([p.detach().clone().requires_grad_(True) for p in pa], lr=1e-3, weight_decay=wd)
    od.load_state_dict(ob.state_dict())
    pc, pd = oc.param_groups[0]["params"], od.param_groups[0]["params"]
    for i, (c, d) in enumerate(zip(pc, pd)):
        if i % 10 != 9:
            gr = torch.randn(c.shape, device="cuda", generator=g)
            c.grad, d.grad = gr.clone(), gr.clone()
    oc.step()
    od.step()
    for c, d in zip(pc, pd):
        torch.testing.assert_close(c, d, rtol=2e-6, atol=2e-7)
    # load_state_dict on an optimiser that HAS stepped re-homes every counter once, eagerly, into a fresh shared
    # buffer (no overflow, no host sync left for a later step): the next step is one advance launch and can be captured
    od.load_state_dict(ob.state_dict())
    live = [d for i, d in enumerate(pd) if i % 10 != 9]
    buf, used = od._step_bufs[live[0].device]
    assert used == len(live) <= buf.numel()
    assert sorted(od.state[d]["step"].data_ptr() for d in live) == [buf.data_ptr() + 4 * i for i in range(used)]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        od.step()   # warm-up on the side stream, then the same step under capture (a host read of a counter would raise)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            od.step()
    graph.replay()
    torch.cuda.synchronize()
    assert all(float(od.state[d]["step"]) == 9.0 for d in live)   # 7 loaded + the warm-up step + one replay (capture does not run)


def test_fused_hand_losses_match_the_torch_composition():
    """ext.HandLosses (two launches) vs HandTrackNet.compute_loss's torch composition (the reference's expressions,
    hand_network.py:159-221): all nine dictionary entries and the gradient of the weighted total w.r.t. pred_kp_handframe
    (keypoint L1 + the closed-form Kabsch gradient of the rotation / translation L1 terms)."""
    from _netinit import make_cfg
    from models.hand_network import HandTrackNet
    from models import pointnet_utils
    from hotrack_amd import pointnet2_utils
    pointnet_utils.set_operator_backend(pointnet2_utils)
    model = HandTrackNet(make_cfg("cuda")).cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    B = 37
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    q = r(B, 4)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    Rc = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                      2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).view(B, 3, 3)
    canon = {"scale": 0.2 * torch.ones(1, device="cuda"), "rotation": Rc, "translation": r(B, 3, 1) * 0.1 + torch.tensor([0, 0, 0.5], device="cuda").view(1, 3, 1)}
    gt_hf = r(B, 3, 21) * 0.4
    gt_kp = (0.2 * (Rc @ gt_hf) + canon["translation"]).transpose(1, 2).contiguous()
    base = (gt_hf + 0.05 * r(B, 3, 21))
    init_hf = gt_hf + 0.08 * r(B, 3, 21)
    data = {"gt_hand_kp": gt_kp, "gt_hand_pose": {"palm_template": (gt_hf[:1, :, [0, 1, 5, 9, 13, 17]].transpose(1, 2) * 0.2).contiguous()}}
    flags = {"track_flag": False, "test_flag": False, "save_flag": False, "IKNet_flag": False}
    weights = {"hand_pred_kp_loss": 10.0, "hand_pred_r_loss": 1.0, "hand_pred_t_loss": 1.0}
    res = {}
    for fused_on in (True, False):
        model.use_fused_losses = fused_on
        p = base.clone().requires_grad_(True)
        ret = {"canon_pose": canon, "pred_kp_handframe": p, "init_kp_handframe": init_hf,
               "pred_kp": (0.2 * (Rc @ p) + canon["translation"]).transpose(1, 2)}
        loss, _ = model.compute_loss(data, ret, dict(flags))
        assert (getattr(loss, "fused_values", None) is not None) == fused_on
        total = sum(loss[k] * w_ for k, w_ in weights.items())
        (gp,) = torch.autograd.grad(total, p)
        res[fused_on] = ({k: float(v) for k, v in loss.items()}, gp)
    la, lb = res[True][0], res[False][0]
    assert list(la) == list(lb)  # same keys, same order as the reference's dictionary
    for k in lb:
        assert abs(la[k] - lb[k]) <= 2e-5 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
    ga, gb = res[True][1], res[False][1]
    assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max()) + 1e-7, float((ga - gb).abs().max())
    # the weighted total formed inside the loss kernel (what the Trainer uses once it has handed its weights to the model)
    from hotrack_amd.ext import HAND_LOSS_NAMES
    model.use_fused_losses = True
    model.fused_loss_weights = torch.tensor([weights.get(k, 0.0) for k in HAND_LOSS_NAMES], device="cuda")
    p = base.clone().requires_grad_(True)
    ret = {"canon_pose": canon, "pred_kp_handframe": p, "init_kp_handframe": init_hf,
           "pred_kp": (0.2 * (Rc @ p) + canon["translation"]).transpose(1, 2)}
    loss, _ = model.compute_loss(data, ret, dict(flags))
    assert loss.fused_total is not None and loss.fused_total_weights is model.fused_loss_weights
    ref_total = sum(lb[k] * w_ for k, w_ in weights.items())
    assert abs(float(loss.fused_total) - ref_total) <= 2e-5 * max(1.0, abs(ref_total))
    (gt_,) = torch.autograd.grad(loss.fused_total, p)
    assert float((gt_ - gb).abs().max()) <= 2e-5 * float(gb.abs().max()) + 1e-7


# ---- the 21-token tail in training mode (csrc/tail_train.hip, hotrack_amd/tail_train.py) ----------------------------------------
@pytest.mark.parametrize("rows,C,two,with_y", [(672, 384, True, False), (672, 384, True, True), (21, 384, False, True), (100, 256, True, True), (7, 64, False, False)])
def test_tail_ln_matches_torch(rows, C, two, with_y):
    """LN_b(LN_a(x + y + bias)) forward / backward (dropout off) vs torch.nn.LayerNorm in fp64."""
    from hotrack_amd import tail_train as T
    g = torch.Generator(device="cuda").manual_seed(rows + C)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    na, nb = torch.nn.LayerNorm(C).cuda(), torch.nn.LayerNorm(C).cuda()
    with torch.no_grad():
        for n in (na, nb):
            n.weight.copy_(1 + 0.3 * r(C))
            n.bias.copy_(0.2 * r(C))
    x, y, bias = r(rows, C).requires_grad_(True), r(rows, C).requires_grad_(True), r(C).requires_grad_(True)
    grads = T.TailGrads("cuda", 8 * C)
    out = T.ln(x, na, nb if two else None, grads, y=y if with_y else None, bias=bias if with_y else None)
    go = r(rows, C)
    out.backward(go)
    got = [t.grad.clone() if t.grad is not None else None for t in (x, y, bias, na.weight, na.bias, nb.weight, nb.bias)]
    for t in (x, y, bias, na.weight, na.bias, nb.weight, nb.bias):
        t.grad = None
    xd, yd, bd = x.detach().double().requires_grad_(True), y.detach().double().requires_grad_(True), bias.detach().double().requires_grad_(True)
    nad, nbd = torch.nn.LayerNorm(C).cuda().double(), torch.nn.LayerNorm(C).cuda().double()
    nad.load_state_dict({k: v.double() for k, v in na.state_dict().items()})
    nbd.load_state_dict({k: v.double() for k, v in nb.state_dict().items()})
    u = xd + (yd + bd if with_y else 0)
    ref = nad(u)
    if two:
        ref = nbd(ref)
    ref.backward(go.double())
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-5)
    want = [xd.grad, yd.grad if with_y else None, bd.grad if with_y else None, nad.weight.grad, nad.bias.grad,
            nbd.weight.grad if two else None, nbd.bias.grad if two else None]
    for a, b, name in zip(got, want, ("dx", "dy", "dbias", "dga", "dba", "dgb", "dbb")):
        if b is None:
            assert a is None or name in ("dy", "dbias"), name
            continue
        assert a is not None, name
        scale = float(b.abs().max()) + 1e-12
        assert float((a.double() - b).abs().max()) / scale < 1e-4, (name, float((a.double() - b).abs().max()) / scale)


def test_tail_relu_dropout_mask_is_consistent_and_unbiased():
    """dropout(relu(z + bias)): p = 0 equals torch exactly; with p > 0 the keep rate is 1 - p, kept elements are scaled by
    1 / (1 - p), the backward regenerates the same mask, and two forwards with different seeds draw different masks."""
    from hotrack_amd import tail_train as T
    g = torch.Generator(device="cuda").manual_seed(3)
    rows, C = 672, 1024
    z = torch.randn(rows, C, device="cuda", generator=g).requires_grad_(True)
    bias = torch.randn(C, device="cuda", generator=g).requires_grad_(True)
    grads = T.TailGrads("cuda", 4 * C)
    h0 = T.relu_dropout(z, bias, 0.0, 1, None, grads)
    assert torch.equal(h0, torch.relu(z + bias))
    h0.backward(torch.ones_like(h0))
    assert torch.equal(z.grad, (z + bias > 0).float()) and torch.allclose(bias.grad, (z + bias > 0).float().sum(0), rtol=1e-5)
    z.grad = bias.grad = None
    seeds = [torch.tensor([s], dtype=torch.int64, device="cuda") for s in (5, 6)]
    outs = []
    for sd in seeds:
        h = T.relu_dropout(z, bias, 0.1, 3, sd, grads)
        pos = (z + bias > 0)
        kept = (h != 0) & pos
        rate = float(kept.sum()) / float(pos.sum())
        assert abs(rate - 0.9) < 0.01, rate
        assert torch.allclose(h[kept], (z + bias)[kept] / 0.9, rtol=1e-6)
        (gz,) = torch.autograd.grad(h, z, torch.ones_like(h))
        assert torch.equal(gz != 0, kept) and torch.allclose(gz[kept], torch.full_like(gz[kept], 1 / 0.9))
        outs.append(kept)
    assert float((outs[0] ^ outs[1]).float().mean()) > 0.05  # different seeds, different masks


def test_fast_tail_equals_module_tail():
    """FastTail (token-major, fused element-wise runs) vs the module composition TransT -> c3 -> final_mlp on the same
    weights with dropout off: outputs 1e-5, every parameter gradient 1e-4 relative."""
    from _netinit import deterministic_init, make_cfg
    from models.hand_network import HandTrackNet
    from models.fast_train import FastTail
    from models.hand_utils import decanonicalize
    import torch.nn.functional as F
    torch.manual_seed(0)
    net = HandTrackNet(make_cfg("cuda"))
    deterministic_init(net)
    net = net.cuda().train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    g = torch.Generator(device="cuda").manual_seed(1)
    B, J, C = 5, 21, 384
    rows = torch.randn(B * J, C, device="cuda", generator=g)
    xyz1 = torch.randn(B, 3, J, device="cuda", generator=g)
    rot = torch.linalg.qr(torch.randn(B, 3, 3, device="cuda", generator=g))[0]
    canon = {"scale": 0.2 * torch.ones(1, device="cuda"), "rotation": rot, "translation": torch.randn(B, 3, 1, device="cuda", generator=g)}
    assert FastTail.supported(net)
    res = {}
    for fast in (True, False):
        net.zero_grad()
        r_ = rows.clone().requires_grad_(True)
        if fast:
            hf, kp = FastTail(net).forward(r_, xyz1, canon)
        else:
            f14 = r_.view(B, J, C).transpose(1, 2)
            f15, f251 = net.transt(src1=f14, pos1=None, src2=None, pos2=None, attn=False, elide_dead=True, need_result2=False)
            fused = net.c3(f15, None, f251, None, attn=False, elide_dead=True)
            lin = lambda conv, x: F.linear(x.transpose(1, 2), conv.weight.squeeze(-1), conv.bias).transpose(1, 2)
            hf = lin(net.final_mlp[2], F.relu(lin(net.final_mlp[0], fused))) + xyz1
            kp = decanonicalize(hf, canon).transpose(2, 1)
        ((hf * torch.cos(torch.arange(hf.numel(), device="cuda").view_as(hf) * 0.1)).sum()
         + (kp * torch.sin(torch.arange(kp.numel(), device="cuda").view_as(kp) * 0.3)).sum()).backward()
        res[fast] = (hf.detach(), kp.detach(), r_.grad, {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    a, b = res[True], res[False]
    torch.testing.assert_close(a[0], b[0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(a[1], b[1], rtol=1e-5, atol=1e-5)
    assert set(a[3]) == set(b[3]) and len(a[3]) > 20
    for k in b[3]:
        scale = float(b[3][k].abs().max()) + 1e-12
        assert float((a[3][k] - b[3][k]).abs().max()) / scale < 1e-4, k
    assert float((a[2] - b[2]).abs().max()) / float(b[2].abs().max()) < 1e-4


@pytest.mark.parametrize("widths,Ka,Kb,G", [([128, 128, 192], 16, 64, 21 * 32), ([128, 128, 192], 16, 64, 21 * 3), ([64, 64, 128], 32, 32, 40),
                                            ([128, 128, 128], 16, 32, 50)])
def test_mlp_stack_pair_equals_two_stacks(widths, Ka, Kb, G):
    """train_stack.mlp_stack_pair -- the fused launches of two sibling stacks grouped into pair launches (csrc/train_fwd.hip
    tgf_pair_kernel, train_bwd.hip tg_bwd_pair_kernel: the two neighbourhood sizes of a keypoint-query module, reference
    pointnet_utils.py:566-581) -- against the same two stacks run one after the other: outputs, arg-max routing, every
    gradient, running statistics.  The 64-channel case has no pair kernel (lockstep single launches), the last one pairs 128 ->
    128 layers of different group sizes."""
    from hotrack_amd import train_stack
    from hotrack_amd.train_ops import Workspace
    g = torch.Generator(device="cuda").manual_seed(G + Ka)

    def make():
        gg = torch.Generator(device="cuda").manual_seed(7)
        convs = [torch.nn.Conv1d(a, b, 1).cuda() for a, b in zip(widths[:-1], widths[1:])]
        bns = [torch.nn.BatchNorm1d(c).cuda().train() for c in widths]
        with torch.no_grad():
            for bn in bns:
                bn.weight.copy_(1 + 0.3 * torch.randn(bn.weight.shape, device="cuda", generator=gg))
                bn.bias.copy_(0.2 * torch.randn(bn.bias.shape, device="cuda", generator=gg))
        return convs, bns

    ya0 = torch.randn(G * Ka, widths[0], device="cuda", generator=g) * 1.5 + 0.3
    yb0 = torch.randn(G * Kb, widths[0], device="cuda", generator=g) * 0.7 - 0.2
    goa = torch.randn(G, widths[-1], device="cuda", generator=g)
    gob = torch.randn(G, widths[-1], device="cuda", generator=g)
    res = {}
    for paired in (False, True):
        torch.manual_seed(3)
        (ca, ba), (cb, bb) = make(), make()
        with torch.no_grad():  # the two scales have DIFFERENT weights
            for c in cb:
                c.weight.mul_(0.5).add_(0.01)
        la = [train_stack.Layer(None, ba[0], None)] + [train_stack.Layer(c.weight, bn, c.bias) for c, bn in zip(ca, ba[1:])]
        lb = [train_stack.Layer(None, bb[0], None)] + [train_stack.Layer(c.weight, bn, c.bias) for c, bn in zip(cb, bb[1:])]
        ws = Workspace("cuda")
        ya, yb = ya0.clone().requires_grad_(True), yb0.clone().requires_grad_(True)
        if paired:
            oa, ob = train_stack.mlp_stack_pair(ya, yb, la, lb, ws, Ka, Kb)
        else:
            oa, ob = train_stack.mlp_stack(ya, la, ws, Ka), train_stack.mlp_stack(yb, lb, ws, Kb)
        ((oa * goa).sum() + (ob * gob).sum()).backward()
        params = [p for m in ca + ba + cb + bb for p in m.parameters()]
        bufs = [b_ for m in ba + bb for b_ in (m.running_mean, m.running_var)]
        res[paired] = ([oa.detach(), ob.detach(), ya.grad, yb.grad] + [p.grad for p in params], bufs)
    for i, (a, b) in enumerate(zip(res[False][0], res[True][0])):
        assert (a is None) == (b is None), i
        if a is not None:  # same kernels, same per-tile arithmetic; only the order of the partial sums differs
            torch.testing.assert_close(b, a, rtol=2e-4, atol=2e-5 * max(1.0, float(a.abs().max())), msg=lambda m: f"tensor {i}: {m}")
    for a, b in zip(res[False][1], res[True][1]):
        torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("with_centre", [False, True])
def test_query_module_pair_launches_equal_single_launches(monkeypatch, with_centre):
    """One keypoint-query module (two neighbourhood sizes, reference pointnet_utils.py:566-581) through train_ops.sa_layer1 +
    train_stack.mlp_stack_pair with every two-problem launch on (layer-1 assembly + statistics, BatchNorm + ReLU + max, the
    routed reduction, the first-layer BatchNorm backward + d(W_xyz), the two row scatters: csrc/train_ops.hip *_pair) against the
    same module with one launch per scale: the streaming kernels run the same per-element code, so everything they produce is
    bit-equal; what passes through the fused GEMM kernels agrees to round-off."""
    from hotrack_amd import train_ops, train_stack
    from hotrack_amd.train_ops import Workspace, inverse_index, sa_layer1
    B, N, S, C = 3, 512, 21, 128
    Ks = (16, 64)
    g = torch.Generator(device="cuda").manual_seed(5)
    xyz = torch.rand(B, N, 3, device="cuda", generator=g)
    cxyz = torch.rand(B, S, 3, device="cuda", generator=g)
    idxs = [torch.randint(0, N, (B, S, K), device="cuda", generator=g, dtype=torch.int32) for K in Ks]
    invs = [inverse_index(i.view(B, -1), N) for i in idxs]
    a1f0 = torch.randn(B, N, 2 * C, device="cuda", generator=g)
    cadd0 = torch.randn(B, S, 2 * C, device="cuda", generator=g) if with_centre else None
    go = [torch.randn(B * S, 192, device="cuda", generator=g) for _ in Ks]
    res = {}
    for paired in (False, True):
        monkeypatch.setattr(train_ops, "PAIR_SCALES", paired)
        monkeypatch.setattr(train_stack, "PAIR_LAUNCH", paired)
        torch.manual_seed(11)
        wxs = [torch.randn(C, 3, device="cuda").requires_grad_(True) for _ in Ks]
        stacks = []
        for _ in Ks:
            convs = [torch.nn.Conv1d(128, 128, 1).cuda(), torch.nn.Conv1d(128, 192, 1).cuda()]
            bns = [torch.nn.BatchNorm1d(c).cuda().train() for c in (128, 128, 192)]
            stacks.append((convs, bns))
        ws = Workspace("cuda")
        a1f = a1f0.clone().requires_grad_(True)
        cadd = cadd0.clone().requires_grad_(True) if with_centre else None
        aux = {}
        y1s = sa_layer1(a1f, cadd, xyz, cxyz, idxs, wxs, invs=invs, aux=aux, ws=ws)
        layers = [[train_stack.Layer(None, bns[0], None)] + [train_stack.Layer(c.weight, bn, c.bias) for c, bn in zip(convs, bns[1:])]
                  for convs, bns in stacks]
        oa, ob = train_stack.mlp_stack_pair(y1s[0].view(-1, C), y1s[1].view(-1, C), layers[0], layers[1], ws, Ks[0], Ks[1],
                                            aux_a=(aux, 0), aux_b=(aux, 1))
        ((oa * go[0]).sum() + (ob * go[1]).sum()).backward()
        params = [p for convs, bns in stacks for m in convs + bns for p in m.parameters()]
        res[paired] = ([y.detach() for y in y1s], [oa.detach(), ob.detach(), a1f.grad] + ([cadd.grad] if with_centre else [])
                       + [w.grad for w in wxs] + [p.grad for p in params])
    for a, b in zip(res[False][0], res[True][0]):
        assert torch.equal(a, b)  # layer-1 pre-activations: the same expressions
    for i, (a, b) in enumerate(zip(res[False][1], res[True][1])):
        assert (a is None) == (b is None), i
        if a is not None:
            torch.testing.assert_close(b, a, rtol=2e-4, atol=2e-5 * max(1.0, float(a.abs().max())), msg=lambda m: f"tensor {i}: {m}")


def test_copy_multi_equals_tensor_copies():
    """pn2x_copy_multi (the batch hand-over of the captured step as one launch): mixed dtypes, sizes around the 16 KB chunk, a
    1-element tensor, more tensors than one launch takes."""
    from hotrack_amd import ext
    g = torch.Generator(device="cuda").manual_seed(3)
    shapes = [(32, 1024, 3), (32, 21, 3), (1,), (4097,), (16384 // 4,), (5, 7, 11)] + [(37 + i,) for i in range(30)]
    srcs = []
    for i, sh in enumerate(shapes):
        t = torch.randn(sh, device="cuda", generator=g)
        srcs.append(t if i % 3 else (t * 1000).to(torch.int32))
    dsts = [torch.zeros_like(s) for s in srcs]
    ext.copy_multi(dsts, srcs)
    for d, s in zip(dsts, srcs):
        assert torch.equal(d, s)
    with pytest.raises(ValueError):
        ext.copy_multi([dsts[0]], [srcs[1]])
