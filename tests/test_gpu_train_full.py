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
    (ma, la, ka), (mb, lb, kb) = _step(False), _step(True)
    assert abs(la - lb) <= 2e-5 * abs(la), (la, lb)
    torch.testing.assert_close(kb, ka, rtol=0, atol=2e-4)
    pa, pb = dict(ma.named_parameters()), dict(mb.named_parameters())
    none_a = [k for k in pa if pa[k].grad is None]
    assert none_a == [k for k in pb if pb[k].grad is None]
    assert len(none_a) == 30 and sum(pa[k].numel() for k in none_a) == 3746944
    # gradient norms: every parameter with a live gradient within 1 % (biases in front of a train-mode BatchNorm have an
    # analytically zero gradient: round-off in the module path, exact zeros in the fused one)
    gmax = max(float(p.grad.norm()) for p in pa.values() if p.grad is not None)
    worst = ("", 0.0)
    for k in pa:
        if pa[k].grad is None:
            continue
        na, nb = float(pa[k].grad.norm()), float(pb[k].grad.norm())
        if na < 1e-6 * gmax:
            assert nb <= 1e-5 * gmax, (k, na, nb)
            continue
        rel = abs(na - nb) / na
        worst = max(worst, (k, rel), key=lambda t: t[1])
        assert rel < 1e-2, (k, na, nb)
        # and direction: the two gradients point the same way (a permuted / mis-strided tile keeps the norm)
        cos = float(torch.dot(pa[k].grad.flatten().double(), pb[k].grad.flatten().double())) / (na * nb)
        assert cos > 0.98, (k, cos)
    ba, bb = dict(ma.named_buffers()), dict(mb.named_buffers())
    for k in ba:
        if k.endswith("num_batches_tracked"):
            assert int(ba[k]) == int(bb[k]), k
        elif k.endswith("running_mean") or k.endswith("running_var"):
            torch.testing.assert_close(bb[k], ba[k], rtol=1e-4, atol=3e-5, msg=lambda m: f"{k}: {m}")


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
