"""GPU: BASELINE.json configs[2] at its PER-GPU size -- one HandTrackNet training step over 32 clouds x 1024 points (what each
of the 8 ranks of the 256-cloud data-parallel job runs; reference composition pointnet_utils.py:399-403,460-462,504-506,577-581,
trainer.py:278-302).  The golden step of the imported reference is 4 clouds (tests/golden/make_golden.py:186) and the graph
test 6 x 512; here the size the bench times -- 262,144-row layer tiles, 2048-tile grids -- is compared:
  * point-major training path (fused BatchNorm GEMMs, one-kernel layer backward) vs the channel-major module path (the ten
    reference operators + torch modules): loss 2e-5, the grad-is-None mask (30 tensors / 3,746,944 parameters), per-parameter
    gradient norms within 1 %, BatchNorm running statistics;
  * the same step replayed as a HIP graph (Trainer graph_step) vs launched eagerly."""
import argparse
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "network"))
pytestmark = pytest.mark.gpu

B, N = 32, 1024  # configs[2]: batch 256 over 8 GPUs


def _step(fast):
    import test_network as tn
    from models.hand_network import HandTrackNet
    from netinit import synthetic_frames
    HandTrackNet._force_fast_train = fast
    try:
        model = tn._build("cuda", True).train()
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        data = synthetic_frames(3000, B, N)
        data = {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in data.items()}
        flags = dict(tn.FLAGS, test_flag=False)
        ret = model(data, flags)
        loss, ret = model.compute_loss(data, ret, flags)
        total = 10 * loss["hand_pred_kp_loss"] + loss["hand_pred_r_loss"] + loss["hand_pred_t_loss"]
        total.backward()
        torch.cuda.synchronize()
    finally:
        del HandTrackNet._force_fast_train
    assert bool(model._ftrain) == fast
    return model, float(total), ret["pred_kp"].detach()


def test_fast_training_path_equals_module_path_at_the_per_gpu_size():
    """Both fp32 paths against the fp64 re-run of the IMPORTED reference's train step on the same 32 x 1024 batch
    (tests/golden/handtracknet_train32_f64.npz, make_golden_train32.py).  Two fp32 implementations of this network agree with
    each other only to ~1e-3 in the train-mode outputs (BatchNorm chains amplify round-off), so each is judged against the
    truth and the fused path must not be further from it than the module path."""
    import numpy as np
    gold = np.load(os.path.join(ROOT, "tests", "golden", "handtracknet_train32_f64.npz"))
    assert tuple(gold["meta"]) == (B, N, 3000)
    truth_loss, truth_kp = float(gold["train_total_loss_f64"]), torch.from_numpy(gold["train_pred_kp_f64"]).cuda()
    (ma, la, ka), (mb, lb, kb) = _step(False), _step(True)
    assert abs(la - lb) <= 2e-5 * abs(la), (la, lb)
    for l in (la, lb):
        assert abs(l - truth_loss) <= 5e-5 * abs(truth_loss), (la, lb, truth_loss)
    ea, eb = float((ka.double() - truth_kp).abs().max()), float((kb.double() - truth_kp).abs().max())
    ma_, mb_ = float((ka.double() - truth_kp).abs().mean()), float((kb.double() - truth_kp).abs().mean())
    assert ea < 5e-3 and eb < 5e-3, (ea, eb)                      # both within fp32 noise of the truth (coordinates are O(1))
    assert mb_ <= 1.25 * ma_ + 1e-5 and eb <= 2.0 * ea + 1e-4, (ma_, mb_, ea, eb)   # and the fused path is no further from it
    pa, pb = dict(ma.named_parameters()), dict(mb.named_parameters())
    names = list(gold["param_names"])
    assert names == list(pa) == list(pb)
    for params in (pa, pb):
        none_mask = np.array([params[k].grad is None for k in names])
        np.testing.assert_array_equal(none_mask, gold["param_grad_is_none"])
    assert int(gold["param_grad_is_none"].sum()) == 30
    assert sum(pa[k].numel() for k in names if pa[k].grad is None) == 3746944
    # per-parameter gradient norms within 1 % of the fp64 truth (analytically zero gradients -- biases in front of a
    # train-mode BatchNorm -- only bounded); element-wise, the fused path is on average no further from the truth
    truth = gold["param_grad_norm_f64"]
    live = truth > 1e-6 * truth.max()
    err, dev = {False: [], True: []}, {}
    for fast, params in ((False, pa), (True, pb)):
        gn = np.array([0.0 if params[k].grad is None else float(params[k].grad.norm()) for k in names])
        dev[fast] = np.abs(gn[live] - truth[live]) / truth[live]
        assert (gn[~live] < 2e-3 * truth.max()).all()
    # the module path (torch modules + the ten operators) is itself up to ~2 % off the fp64 norms at this batch size; the fused
    # path must be within 1 % of the truth, or at least no further from it than the module path is, parameter by parameter
    assert dev[False].max() < 3e-2, dev[False].max()
    ok = (dev[True] <= 1e-2) | (dev[True] <= 1.25 * dev[False] + 1e-3)
    assert ok.all(), [(names[i], float(dev[True][j]), float(dev[False][j])) for j, i in enumerate(np.flatnonzero(live)) if not ok[j]]
    for fast, params in ((False, pa), (True, pb)):
        for k, is_live in zip(names, live):
            if not is_live or params[k].grad is None:
                continue
            t = torch.from_numpy(gold["g64/" + k]).cuda()
            e = float((params[k].grad.flatten()[:256].double() - t).abs().max()) / max(float(t.abs().max()), 1e-30)
            assert e < 0.25, (k, fast, e)
            err[fast].append(e)
    mean = {f: sum(v) / len(v) for f, v in err.items()}
    assert mean[True] <= 1.25 * mean[False] + 1e-3, mean
    # BatchNorm running statistics: the two paths against each other (same fp32 statistics) and against the truth
    ba, bb = dict(ma.named_buffers()), dict(mb.named_buffers())
    for k in ba:
        if k.endswith("num_batches_tracked"):
            assert int(ba[k]) == int(bb[k]), k
        elif k.endswith("running_mean") or k.endswith("running_var"):
            torch.testing.assert_close(bb[k], ba[k], rtol=1e-4, atol=3e-5, msg=lambda m: f"{k}: {m}")
            torch.testing.assert_close(bb[k].double(), torch.from_numpy(gold["buf/" + k]).cuda(), rtol=2e-4, atol=5e-5, msg=lambda m: f"{k} vs fp64: {m}")


def test_graph_captured_step_equals_eager_step_at_the_per_gpu_size(tmp_path, monkeypatch):
    monkeypatch.setenv("HOTRACK_DATA_ROOT", str(tmp_path))
    from configs.config import get_config
    from datasets.synthetic import make_frame
    from parse_args import add_args
    from trainer import Trainer

    def build(graph):
        a = add_args(argparse.ArgumentParser()).parse_args(["--config", "handtracknet_train_SimGrasp.yml"])
        a.num_points, a.batch_size = N, B
        cfg = get_config(a, save=False)
        cfg["graph_step"] = graph
        torch.manual_seed(0)
        tr = Trainer(cfg)
        tr.step_epoch()
        for m in tr.model.modules():  # the FFN dropouts draw different masks in two runs: off, to compare
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        return tr

    batch = torch.utils.data.default_collate([make_frame(7000 + i, N, 0.02) for i in range(B)])
    batch = {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in batch.items()}
    eager, graph = build(False), build(True)
    graph.load_state_dict(eager.state_dict())
    le = eager.update(batch)["total_loss"].item()
    lg = graph.update(batch)["total_loss"].item()
    assert graph.graph_step, "capture fell back to eager"
    assert abs(le - lg) <= 2e-5 * max(1.0, abs(le)), (le, lg)
    pe, pg = dict(eager.model.named_parameters()), dict(graph.model.named_parameters())
    checked = 0
    for k in pe:
        st = eager.optimizer.state.get(pe[k])
        if not st or float(st["exp_avg_sq"].max()) <= 1e-9:
            continue
        sg = graph.optimizer.state[pg[k]]
        # the first Adam moment is 0.1 g: the two launches of the same kernels differ by atomics order only
        ne = float(st["exp_avg"].norm())
        assert abs(ne - float(sg["exp_avg"].norm())) <= 1e-2 * ne + 1e-7, k
        real = st["exp_avg_sq"] > 1e-9
        assert float(((pe[k] - pg[k]).abs() * real).max()) < 2e-5, k   # a lost / doubled / stale step would be 1e-4 (= lr)
        checked += 1
    assert checked > 50
    assert sum(p.grad is None for p in graph.model.parameters()) == sum(p.grad is None for p in eager.model.parameters()) == 30
    # the replayed step keeps training on the same batch
    for _ in range(3):
        lg2 = graph.update(batch)["total_loss"].item()
    assert lg2 == lg2 and lg2 < lg


def test_geometry_prefetch_equals_inline_geometry(tmp_path, monkeypatch):
    """update(data, next_data=...) -- the next batch's geometry graph replayed on a second stream beside this batch's dense
    step (Trainer._geometry_for) -- against the same captured step with the geometry stage run in line: same losses step by
    step over a rotation of three different batches (a stale / swapped geometry pack would show at once), including a step
    whose announced next batch is NOT the one that follows."""
    monkeypatch.setenv("HOTRACK_DATA_ROOT", str(tmp_path))
    from configs.config import get_config
    from datasets.synthetic import make_frame
    from parse_args import add_args
    from trainer import Trainer

    def build(prefetch):
        a = add_args(argparse.ArgumentParser()).parse_args(["--config", "handtracknet_train_SimGrasp.yml"])
        a.num_points, a.batch_size = 512, 8
        cfg = get_config(a, save=False)
        cfg["graph_step"], cfg["prefetch_geometry"] = True, prefetch
        cfg["learning_rate"] = 0.0  # frozen parameters: a step's loss is a function of its batch (and its geometry) only
        torch.manual_seed(0)
        tr = Trainer(cfg)
        tr.step_epoch()
        for m in tr.model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        return tr

    batches = []
    for j in range(3):
        b = torch.utils.data.default_collate([make_frame(900 + 40 * j + i, 512, 0.02) for i in range(8)])
        batches.append({k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in b.items()})
    a, b = build(True), build(False)
    b.load_state_dict(a.state_dict())
    order = [0, 1, 2, 0, 2, 1, 1, 0]
    la, lb = [], []
    for s, i in enumerate(order):
        nxt = batches[order[s + 1]] if s + 1 < len(order) else None
        if s == 3:
            nxt = batches[1]  # a wrong announcement: the step after this one gets batches[2], not the prefetched batch
        la.append(a.update(batches[i], next_data=nxt)["total_loss"].item())
        lb.append(b.update(batches[i])["total_loss"].item())
    assert a.graph_step and b.graph_step and a._geo_graph is not None and b._geo_graph is None
    for s, (x, y) in enumerate(zip(la, lb)):
        assert abs(x - y) <= 2e-5 * max(1.0, abs(y)), (s, la, lb)
    # the same batch gives the same loss whenever it comes round, and different batches give different losses (so a stale or
    # swapped geometry pack could not hide)
    by_batch = {}
    for i, x in zip(order, la):
        by_batch.setdefault(i, []).append(x)
    assert all(max(v) - min(v) <= 2e-5 * max(v) for v in by_batch.values()), by_batch
    firsts = sorted(v[0] for v in by_batch.values())
    assert all(b_ - a_ > 1e-3 * a_ for a_, b_ in zip(firsts, firsts[1:])), by_batch


def test_geometry_prefetch_survives_recapture_and_shape_changes(tmp_path, monkeypatch):
    """The transitions around the prefetch: a re-capture while a prefetch is in flight (epoch boundary: learning rate and BN
    momentum are baked into the graphs), a batch of another shape announced as `next_data` (ignored: that batch re-captures and
    runs its geometry in line), and back.  Parameters frozen, so every step's loss must equal the loss of the same batch in a
    trainer without prefetch."""
    monkeypatch.setenv("HOTRACK_DATA_ROOT", str(tmp_path))
    from configs.config import get_config
    from datasets.synthetic import make_frame
    from parse_args import add_args
    from trainer import Trainer

    def build(prefetch):
        a = add_args(argparse.ArgumentParser()).parse_args(["--config", "handtracknet_train_SimGrasp.yml"])
        a.num_points, a.batch_size = 512, 8
        cfg = get_config(a, save=False)
        cfg["graph_step"], cfg["prefetch_geometry"], cfg["learning_rate"] = True, prefetch, 0.0
        torch.manual_seed(0)
        tr = Trainer(cfg)
        tr.step_epoch()
        for m in tr.model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        return tr

    def batch(seed, n):
        b = torch.utils.data.default_collate([make_frame(seed + i, 512, 0.02) for i in range(n)])
        return {k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in b.items()}

    big = [batch(100, 8), batch(200, 8), batch(300, 8)]
    small = batch(400, 5)
    a, b = build(True), build(False)
    b.load_state_dict(a.state_dict())
    seq = [big[0], big[1], "epoch", big[2], big[0], small, big[1], big[2]]
    steps = [x for x in seq if not isinstance(x, str)]
    la, lb, k = [], [], 0
    for x in seq:
        if isinstance(x, str):   # epoch boundary: both trainers drop their graphs (the prefetch of big[2] is in flight in `a`)
            a.step_epoch(); b.step_epoch()
            continue
        nxt = steps[k + 1] if k + 1 < len(steps) else None
        la.append(a.update(x, next_data=nxt)["total_loss"].item())
        lb.append(b.update(x)["total_loss"].item())
        k += 1
    torch.cuda.synchronize()
    assert a.graph_step and b.graph_step
    for i, (x, y) in enumerate(zip(la, lb)):
        assert abs(x - y) <= 2e-5 * max(1.0, abs(y)), (i, la, lb)
    assert abs(la[0] - la[3]) <= 2e-5 * abs(la[0]) and abs(la[1] - la[5]) <= 2e-5 * abs(la[1])   # big[0], big[1] come round again


@pytest.mark.parametrize("segments", [1, 2])
def test_flat_exchange_graph_step_is_zero_copy_and_equals_the_single_process_step(tmp_path, segments):
    """dp = flat as HIP graphs with a one-rank RCCL group (the exchange path end to end on one GPU): the gradients' producers write
    into the flat exchange buffers (hotrack_amd.train_stack.grad_buffer), so the per-step copy into them moves only the small
    leftovers (< 10 % of the buffer; rounds 4-5 packed and scattered all 16.7 MB), every .grad the optimiser reads IS a view of a
    flat buffer, and four steps follow the losses / parameters of the single-process captured step."""
    import json
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HOTRACK_DATA_ROOT=str(tmp_path))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_flat_worker.py"), str(segments), str(port)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["graph_step"] and r["graphs"] == [True, segments == 2, True]
    assert r["grads_in_flat"] and r["n_none"] == 30
    assert len(r["flat_floats"]) == segments
    assert sum(r["moved_floats"]) <= 0.10 * sum(r["flat_floats"]), r
    # step 0 sees the same parameters and batch: equal; from there two runs of the SAME configuration drift apart through the order
    # of the BatchNorm sums' atomics (two single-process runs differ by 1e-3 in the fourth loss: Adam's first steps move every
    # parameter by ~lr whatever its gradient's size).  A gradient that did not reach the optimiser (stale / misplaced slice of the
    # flat buffer) leaves the losses of the first replays unchanged or sends them off by >= 1e-1.
    tol = [2e-5, 1e-4, 5e-3, 5e-3]
    for a, b, t in zip(r["l_solo"], r["l_dp"], tol):
        assert abs(a - b) <= t * max(1.0, abs(a)), r
    assert r["l_dp"][1] < r["l_dp"][0] and r["param_diff"] < 2e-3, r
