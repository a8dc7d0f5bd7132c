"""GPU: the grouped end-of-pass weight-gradient launch (csrc/train_wgrad.hip, hotrack_amd/linear_dw.py) -- the products
`grad_out^T . input` of the plain linear layers of the training path (layer 1 of every stack over the un-grouped points,
rearrange linears, 21-token tail: reference pointnet_utils.py:399-403,460-462,504-506,577-581, blocks.py:226-239,
transformer.py:72-82) against torch: the kernel alone on the shapes of a training step (incl. K = 131 rows that are not
16-byte loadable, column blocks of a wider weight, single-split and split problems), and through autograd against
torch.nn.functional.linear (deferred, not deferrable, accumulated into an existing gradient; run-to-run bit-equal)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run_multi(problems, dev):
    """problems: list of (g, x, dw2d view to be written)."""
    from hotrack_amd import train_stack as ts
    items = [ts.WgradItem(g, x, dw, dw.data_ptr(), dw.stride(0), g.shape[1], x.shape[1], torch.cuda.current_stream().cuda_stream)
             for g, x, dw in problems]
    ts.wgrad_multi(items)
    torch.cuda.synchronize()


def _check(g, x, dw, scale=3.0):
    ref64 = g.double().t() @ x.double()
    err = float((dw.double() - ref64).abs().max())
    lib = float(((g.t() @ x).double() - ref64).abs().max())  # the library's fp32 product as the yardstick
    tol = max(scale * lib, 1e-6 * float(ref64.abs().max()) + 1e-7)
    assert err <= tol, (tuple(g.shape), tuple(x.shape), err, lib)


# (rows, N, K) of one training step at 32 x 1024 (per-GPU batch of BASELINE configs[2]) + edge shapes
STEP = [(32768, 128, 384)] * 4 + [(32768, 128, 131), (8192, 256, 384), (4096, 256, 640), (4096, 128, 131), (8192, 64, 64),
                                  (672, 384, 1536), (672, 384, 1536), (672, 1024, 384), (672, 384, 1024), (672, 1024, 384),
                                  (672, 384, 1024), (672, 256, 384)]


def test_wgrad_multi_matches_torch_on_the_training_step_shapes():
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(1)
    probs = []
    for r, n, k in STEP:
        g = torch.randn((r, n), device=dev, generator=gen)
        x = torch.randn((r, k), device=dev, generator=gen)
        probs.append((g, x, torch.full((n, k), float("nan"), device=dev)))
    _run_multi(probs, dev)
    for g, x, dw in probs:
        _check(g, x, dw)


@pytest.mark.parametrize("r,n,k", [(1, 4, 4), (33, 5, 7), (100, 130, 129), (2500, 96, 200), (40000, 32, 3), (9000, 257, 64)])
def test_wgrad_multi_edge_shapes(r, n, k):
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(r + n + k)
    g = torch.randn((r, n), device=dev, generator=gen)
    x = torch.randn((r, k), device=dev, generator=gen)
    dw = torch.full((n, k), float("nan"), device=dev)
    # a second, long problem in the same launch so that the small one is cut with the launch's rows-per-split
    g2 = torch.randn((30000, 128), device=dev, generator=gen)
    x2 = torch.randn((30000, 128), device=dev, generator=gen)
    dw2 = torch.empty((128, 128), device=dev)
    _run_multi([(g, x, dw), (g2, x2, dw2)], dev)
    _check(g, x, dw)
    _check(g2, x2, dw2)


def test_wgrad_multi_strided_operands_and_column_block_output():
    """g a column block of a wider gradient, x with padded rows, dW the feature block of a [feature | xyz | centre] weight:
    the columns next to the block must stay untouched."""
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(7)
    gw = torch.randn((20000, 256), device=dev, generator=gen)
    xw = torch.randn((20000, 400), device=dev, generator=gen)
    full = torch.full((128, 771), 5.0, device=dev)
    g, x = gw[:, 128:256], xw[:, :384]
    _run_multi([(g, x, full[:, :384])], dev)
    _check(g, x, full[:, :384])
    assert bool((full[:, 384:] == 5.0).all())
    # single split (direct store), unaligned output rows (lddw = 387)
    full2 = torch.full((128, 387), 5.0, device=dev)
    _run_multi([(g[:600], x[:600], full2[:, :384])], dev)
    _check(g[:600], x[:600], full2[:, :384])
    assert bool((full2[:, 384:] == 5.0).all())


def test_wgrad_multi_more_problems_than_one_launch_holds_and_is_deterministic():
    from hotrack_amd import train_stack as ts
    dev = torch.device("cuda")
    cap = int(ts._lib.pn2x_wgrad_multi_max())
    gen = torch.Generator(device=dev).manual_seed(3)
    probs = []
    for i in range(cap + 5):
        r, n, k = 700 + 331 * i, 32 + 8 * (i % 5), 48 + 4 * (i % 7)
        probs.append((torch.randn((r, n), device=dev, generator=gen), torch.randn((r, k), device=dev, generator=gen),
                      torch.empty((n, k), device=dev)))
    _run_multi(probs, dev)
    first = [dw.clone() for _, _, dw in probs]
    for g, x, dw in probs:
        _check(g, x, dw)
        dw.fill_(0)
    _run_multi(probs, dev)
    for a, (_, _, dw) in zip(first, probs):
        assert torch.equal(a, dw)


def test_wgrad_multi_rejects_bad_arguments():
    from hotrack_amd import train_stack as ts
    lib = ts._lib
    one = (ctypes.c_int * 1)
    assert lib.pn2x_wgrad_multi_scratch_floats(1, one(0), one(4), one(4)) == -1
    assert lib.pn2x_wgrad_multi_scratch_floats(int(lib.pn2x_wgrad_multi_max()) + 1, one(1), one(4), one(4)) == -1
    dev = torch.device("cuda")
    g, x, dw = torch.ones((4096 * 40, 8), device=dev), torch.ones((4096 * 40, 8), device=dev), torch.empty((8, 8), device=dev)
    vp = (ctypes.c_void_p * 1)
    rc = lib.pn2x_wgrad_multi(1, vp(g.data_ptr()), one(8), vp(x.data_ptr()), one(8), one(g.shape[0]), one(8), one(8), vp(dw.data_ptr()),
                              one(8), None, 0, None)
    assert rc != 0  # a split problem without scratch


class _Net(torch.nn.Module):
    def __init__(self, D=64, C=32, Dc=16):
        super().__init__()
        self.conv_a = torch.nn.Conv2d(D + 3, C, 1)            # first layer [feature | xyz]
        self.conv_b = torch.nn.Conv2d(D + 3 + Dc, C, 1)       # first layer [feature | xyz | centre]
        self.conv_c = torch.nn.Conv2d(D + 3 + Dc, C, 1)
        self.lin = torch.nn.Linear(2 * C, 24)
        self.conv1 = torch.nn.Conv1d(24, 16, 1)
        self.D = D


def _net_forward(net, x, deferred):
    """A small graph with every kind of deferred product: two modules' per-point first layers on shared rows (one of them with
    two scales and centre blocks), a Linear with bias, a 1x1 Conv1d weight."""
    import torch.nn.functional as F
    D = net.D
    if deferred:
        from hotrack_amd.linear_dw import linear, per_point_first_layer
        (a, bc), blocks, _share = per_point_first_layer(x, [[net.conv_a.weight], [net.conv_b.weight, net.conv_c.weight]], D)
        wx = [b[0] for mod in blocks for b in mod]
        wc = [b[1] for mod in blocks for b in mod if b[1] is not None]
    else:
        w2 = lambda c: c.weight.view(c.weight.shape[0], -1)
        a = F.linear(x, w2(net.conv_a)[:, :D])
        bc = F.linear(x, torch.cat([w2(net.conv_b)[:, :D], w2(net.conv_c)[:, :D]], 0))
        wx = [w2(c)[:, D:D + 3] for c in (net.conv_a, net.conv_b, net.conv_c)]
        wc = [w2(c)[:, D + 3:] for c in (net.conv_b, net.conv_c)]
    extra = sum((w * w).sum() for w in wx) + sum(w.sum() * 0.5 for w in wc)  # the xyz / centre blocks get a gradient of their own
    h = torch.relu(torch.cat([a, bc[:, :a.shape[1]] * bc[:, a.shape[1]:]], 1))
    if deferred:
        y = linear(h, net.lin.weight, net.lin.bias)
        z = linear(torch.tanh(y), net.conv1.weight)
    else:
        y = F.linear(h, net.lin.weight, net.lin.bias)
        z = F.linear(torch.tanh(y), net.conv1.weight.squeeze(-1))
    return (z * z).mean() + extra


@pytest.mark.parametrize("mode", ["defer", "accumulate", "no_defer"])
def test_deferred_weight_gradients_equal_autograd(mode):
    from hotrack_amd import train_stack as ts
    dev = torch.device("cuda")
    torch.manual_seed(0)
    net = _Net().to(dev)
    x = torch.randn((5000, net.D), device=dev, requires_grad=True)
    loss = _net_forward(net, x, False)
    loss.backward()
    ref = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}  # (the conv biases are not used)
    ref_x = x.grad.clone()
    x.grad = None
    old = ts.DEFER_REDUCE
    try:
        if mode == "no_defer":
            ts.DEFER_REDUCE = False
        if mode != "accumulate":
            for p in net.parameters():
                p.grad = None  # ("accumulate": the reference gradients stay in place -> nothing may be deferred, result = 2 x)
        loss2 = _net_forward(net, x, True)
        assert abs(float(loss2) - float(loss)) <= 1e-6 * abs(float(loss))
        loss2.backward()
        torch.cuda.synchronize()
    finally:
        ts.DEFER_REDUCE = old
    assert not ts._pending
    k = 2.0 if mode == "accumulate" else 1.0
    for n, p in net.named_parameters():
        if n not in ref:
            assert p.grad is None, n
            continue
        want = k * ref[n]
        tol = 2e-5 * float(want.abs().max()) + 1e-7
        assert float((p.grad - want).abs().max()) <= tol, (mode, n)
    assert float((x.grad - ref_x).abs().max()) <= 2e-5 * float(ref_x.abs().max())


def test_deferred_weight_gradients_are_run_to_run_bit_equal_and_graph_capturable():
    dev = torch.device("cuda")
    torch.manual_seed(1)
    net = _Net().to(dev)
    x = torch.randn((20000, net.D), device=dev)

    def step():
        for p in net.parameters():
            p.grad = None
        _net_forward(net, x, True).backward()
        return [p.grad for p in net.parameters() if p.grad is not None]

    a = [g.clone() for g in step()]
    b = [g.clone() for g in step()]
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    # the same pass captured into a HIP graph and replayed (the end-of-pass launch is part of the capture)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        grads = step()
    for g in grads:
        g.zero_()
    graph.replay()
    torch.cuda.synchronize()
    for u, v in zip(a, grads):
        assert torch.equal(u, v)


@pytest.mark.parametrize("mode", ["defer", "no_defer"])
def test_linear_blocks_equals_linear_of_the_concatenation(mode):
    """linear_dw.linear_blocks -- layer 1 over [skip | interpolated | per-cloud feature broadcast over the cloud's points] as the sum
    of its column blocks' products (reference pointnet_utils.py:437-438, 451-453: repeat + cat + Conv1d) -- against
    torch.nn.functional.linear of the materialised concatenation: values, every input gradient (the broadcast block's summed over
    the rows it covers), the weight gradient written block by block with a column offset."""
    import torch.nn.functional as F
    from hotrack_amd import train_stack as ts
    from hotrack_amd.linear_dw import linear_blocks
    dev = torch.device("cuda")
    torch.manual_seed(2)
    B, n, ka, kb, kc, N = 6, 50, 3, 40, 24, 72
    xa = torch.randn(B * n, ka, device=dev)                       # e.g. coordinates: no gradient
    xb = torch.randn(B * n, kb, device=dev, requires_grad=True)
    xc = torch.randn(B, kc, device=dev, requires_grad=True)       # one row per cloud
    conv = torch.nn.Conv1d(ka + kb + kc, N, 1).to(dev)
    go = torch.randn(B * n, N, device=dev)
    full = torch.cat([xa, xb, xc.repeat_interleave(n, dim=0)], dim=1)
    y0 = F.linear(full, conv.weight.squeeze(-1))
    (y0 * go).sum().backward()
    ref = (y0.detach(), xb.grad.clone(), xc.grad.clone(), conv.weight.grad.clone())
    xb.grad = xc.grad = conv.weight.grad = None
    old = ts.DEFER_REDUCE
    try:
        if mode == "no_defer":
            ts.DEFER_REDUCE = False
        y1 = linear_blocks([xa, xb, (xc, n)], conv.weight)
        (y1 * go).sum().backward()
        torch.cuda.synchronize()
    finally:
        ts.DEFER_REDUCE = old
    assert not ts._pending
    for got, want in zip((y1.detach(), xb.grad, xc.grad, conv.weight.grad), ref):
        assert got.shape == want.shape
        assert float((got - want).abs().max()) <= 3e-5 * float(want.abs().max()) + 1e-6


@pytest.mark.parametrize("order", ["product_first", "tap_first"])
def test_twice_used_tensor_gradient_sum_inside_the_product(order):
    """linear_dw.tap / GradStash: a tensor read by a product AND by a later consumer gets ONE input gradient, parked by the later
    consumer and summed by the product's addmm -- equal to autograd's own sum (FUSE_GRAD_SUMS = False), in the intended order of
    the two backwards and in the other one (the tap created before the product: the product then finds nothing parked and the
    tap returns its gradient the ordinary way)."""
    from hotrack_amd import linear_dw as L
    dev = torch.device("cuda")
    torch.manual_seed(1)
    lin1, lin2 = torch.nn.Linear(48, 40).to(dev), torch.nn.Linear(7, 24).to(dev)
    w3 = torch.nn.Parameter(torch.randn(16, 48 + 7, device=dev))
    x0 = torch.randn(300, 48, device=dev)
    z0 = torch.randn(300, 7, device=dev)

    def run(fuse):
        L.FUSE_GRAD_SUMS = fuse
        for p in (*lin1.parameters(), *lin2.parameters(), w3):
            p.grad = None
        x = (x0 * 1.0).requires_grad_(True)
        z = (z0 * 1.0).requires_grad_(True)
        xs = x * 1.5  # (non-leaf: both consumers send it a gradient)
        zs = z * 0.5
        st, st2 = L.GradStash(), L.GradStash()
        if order == "product_first":
            a = L.linear(xs, lin1.weight, lin1.bias, stash=st)
            b = L.linear_blocks([xs, zs], w3, stashes=[None, st2])
            later = torch.tanh(L.tap(xs, st)).sum() + (L.tap(zs, st2) ** 2).sum()
        else:
            tx, tz = L.tap(xs, st), L.tap(zs, st2)
            later = torch.tanh(tx).sum() + (tz ** 2).sum()
            a = L.linear(xs, lin1.weight, lin1.bias, stash=st)
            b = L.linear_blocks([xs, zs], w3, stashes=[None, st2])
        loss = (a * a).mean() + b.sum() * 0.01 + later * 0.1 + L.linear(zs, lin2.weight, lin2.bias).sum()
        loss.backward()
        torch.cuda.synchronize()
        return [x.grad.clone(), z.grad.clone()] + [p.grad.clone() for p in (*lin1.parameters(), *lin2.parameters(), w3)]
    try:
        got, ref = run(True), run(False)
    finally:
        L.FUSE_GRAD_SUMS = True
    for g, r in zip(got, ref):
        torch.testing.assert_close(g, r, rtol=1e-5, atol=1e-6)
