"""GPU: the tracking loop (network/test.py path): HIP-graph replay per frame == eager per frame, and the
test.py entry point runs end to end on synthetic sequences."""
import argparse
import copy
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "network"))
pytestmark = pytest.mark.gpu


def _cfg():
    from _netinit import make_cfg
    cfg = make_cfg(torch.device("cuda", 0))
    cfg.update(num_points=1024, hand_jitter_cfg={"rand_scale": 0.01}, track="hand", use_optimization=False)
    return cfg


def test_graph_tracking_equals_eager_tracking():
    from _netinit import deterministic_init
    from datasets.synthetic import SyntheticSequences
    from hotrack_amd import fused, pointnet2_utils
    from models import pointnet_utils
    from models.hand_network import HandTrackNet
    from models.track_network import HandTrackModel
    pointnet_utils.set_operator_backend(pointnet2_utils)
    cfg = _cfg()
    model = HandTrackModel(cfg, handnet=HandTrackNet)
    deterministic_init(model)
    model = model.cuda().eval()
    seq = SyntheticSequences(cfg, 1, 6)[0]
    flags = {"track_flag": True, "test_flag": True, "save_flag": False, "IKNet_flag": False}
    try:
        pointnet_utils.set_fused_backend(fused)
        with torch.no_grad():
            model.use_graph = True
            a = model(copy.deepcopy(seq), dict(flags))
            assert len(model._graphs) == 1
            b2 = model(copy.deepcopy(seq), dict(flags))  # second sequence reuses the captured graph
            model.use_graph = False
            b = model(copy.deepcopy(seq), dict(flags))
    finally:
        pointnet_utils.set_fused_backend(None)
    for ra, rb, rc in zip(a, b, b2):
        assert torch.equal(ra["pred_kp"], rc["pred_kp"])
        assert torch.allclose(ra["pred_kp"], rb["pred_kp"], atol=1e-5)
    loss, _ = model.compute_loss(seq, a, dict(flags))
    assert all(v == v for v in loss.values())  # finite


def test_test_py_entry_point_on_gpu(tmp_path, monkeypatch):
    monkeypatch.setenv("HOTRACK_DATA_ROOT", str(tmp_path))
    import test as test_entry
    from parse_args import add_args
    p = add_args(argparse.ArgumentParser())
    p.add_argument("--mode_name", default="test")
    a = p.parse_args(["--config", "handtracknet_test_SimGrasp.yml"])
    a.num_points, a.synthetic_frames = 1024, 5
    test_entry.main(a)


def test_loader_side_fps_batched_equals_per_cloud(oracle):
    import numpy as np
    from datasets.data_utils import farthest_point_sample_batch
    rng = np.random.default_rng(0)
    clouds = [rng.random((n, 3)).astype(np.float32) for n in (6000, 9000, 2000, 5120, 700)]
    npoint = 512

    class Fixed:  # deterministic "random" permutation so both calls see the same pre-subsample
        def __init__(self):
            self.r = np.random.default_rng(1)

        def permutation(self, n):
            return self.r.permutation(n)
    batched = farthest_point_sample_batch(clouds, npoint, "cuda", rng=Fixed())
    f = Fixed()
    single = [farthest_point_sample_batch([c], npoint, "cuda", rng=f)[0] for c in clouds]
    for b, s, c in zip(batched, single, clouds):
        assert len(b) == npoint and len(set(b.tolist())) == npoint and b.max() < len(c)
        np.testing.assert_array_equal(b, s)
    # the sampled subset is exactly the oracle's FPS of the pre-subsampled cloud
    f = Fixed()
    keep = f.permutation(6000)[:5 * npoint]
    ref = oracle.furthest_point_sample(clouds[0][keep][None], npoint)[0]
    np.testing.assert_array_equal(batched[0], keep[ref])


def test_graph_step_trains_like_eager(tmp_path, monkeypatch):
    """Trainer(graph_step=True): forward + loss + backward + Adam replayed as one HIP graph must follow the eager
    trajectory (same batches, same init): losses and parameters after a few steps, BN statistics, Adam step count."""
    import argparse
    import torch
    monkeypatch.setenv("HOTRACK_DATA_ROOT", str(tmp_path))
    from configs.config import get_config
    from datasets.synthetic import make_frame
    from parse_args import add_args
    from trainer import Trainer

    def build(graph):
        a = add_args(argparse.ArgumentParser()).parse_args(["--config", "handtracknet_train_SimGrasp.yml"])
        a.num_points, a.batch_size = 512, 6
        cfg = get_config(a, save=False)
        cfg["graph_step"] = graph
        torch.manual_seed(0)
        tr = Trainer(cfg)
        tr.step_epoch()
        for m in tr.model.modules():  # the FFN dropouts draw different masks in the two runs: switch them off to compare
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        return tr

    batches = []
    for j in range(2):
        b = torch.utils.data.default_collate([make_frame(50 * j + i, 512, 0.02) for i in range(6)])
        batches.append({k: (v.cuda() if torch.is_tensor(v) else {kk: vv.cuda() for kk, vv in v.items()}) for k, v in b.items()})
    eager, graph = build(False), build(True)
    graph.load_state_dict(eager.state_dict())
    pe, pg = dict(eager.model.named_parameters()), dict(graph.model.named_parameters())
    # ---- one update: same loss (the forward is the same computation), same step for every parameter that receives a
    # real gradient.  (Adam's first steps are lr * sign(g): for the biases BatchNorm cancels, g is ~1e-9 round-off and
    # the sign is random in any two runs -- and after a few such steps two runs drift apart at the 1e-3 level, which is
    # why only the first update is compared tightly.)
    le = eager.update(batches[0])["total_loss"].item()
    lg = graph.update(batches[0])["total_loss"].item()
    assert graph.graph_step, "capture fell back to eager"
    assert abs(le - lg) <= 1e-4 * max(1.0, abs(le)), (le, lg)
    checked, worst = 0, 0.0
    for k in pe:
        st = eager.optimizer.state.get(pe[k])
        if st and float(st["exp_avg_sq"].max()) > 1e-9:
            real = st["exp_avg_sq"] > 1e-9
            worst = max(worst, float(((pe[k] - pg[k]).abs() * real).max()))
            sg_ = graph.optimizer.state[pg[k]]
            # BN-cancellation noise in g is ~1e-4 between ANY two runs (atomics order in the statistics / the partial sums; the
            # small batch of this test amplifies it): once in ~40 full-suite runs an element exceeded 5e-5 in exp_avg = 0.1 g
            torch.testing.assert_close(sg_["exp_avg"] * real, st["exp_avg"] * real, rtol=5e-2, atol=2e-4, msg=lambda m: f"{k}: {m}")
            checked += 1
    assert checked > 50 and worst < 2e-5, (checked, worst)   # a lost / doubled / stale Adam step would be 1e-4
    # ---- four more: the replayed step keeps training
    first = lg
    for step in range(1, 5):
        eager.update(batches[step % 2])
        lg = graph.update(batches[step % 2])["total_loss"].item()
        assert graph.graph_step and lg == lg
    assert lg < 0.8 * first, (first, lg)
    be, bg = dict(eager.model.named_buffers()), dict(graph.model.named_buffers())
    for k in be:
        if k.endswith("num_batches_tracked"):
            assert int(be[k]) == int(bg[k]) == 5
        elif k.endswith("running_mean"):
            torch.testing.assert_close(bg[k], be[k], rtol=5e-2, atol=5e-3)  # five drifting steps apart, same statistics
    se = next(iter(eager.optimizer.state.values()))["step"]
    sg = next(iter(graph.optimizer.state.values()))["step"]
    assert float(se) == float(sg) == 5.0
    # parameters the reference never uses keep grad None (DDP contract) under capture too
    assert sum(p.grad is None for p in graph.model.parameters()) == sum(p.grad is None for p in eager.model.parameters()) > 0


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the fields the driver and the judge read (short run, no CPU leg)."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--min-time", "0.5", "--no-cpu-baseline",
                          "--train-steps", "6"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "frames/s" and d["dtype"] == "f32" and d["vs_baseline"] is None and "workload" in d["config"]
    assert abs(d["value"] - d["config"]["per_gpu_batch"] * 1e3 / d["ms_per_step"]) <= 0.01 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    # the replays of the serving loop were compared with the eager forward at the headline size (64 x 1024, every stream)
    rc = d["replay_check"]
    assert rc["batches"] == 5 and rc["streams"] == d["config"]["batches_in_flight"] and rc["max_abs_diff_pred_kp"] <= 1e-5
    assert d["gemm_table"] == "applied", d["config"].get("gemm_table_detail")
    # the secondary legs (child processes after the headline): configs[2] per GPU, configs[4] per GPU, B = 1 latency
    tr, st_, lat = d["train"], d["stress"], d["latency_b1"]
    assert "error" not in tr and tr["graph_step"] is True and tr["per_gpu_batch"] == 32 and 0 < tr["ms_per_step"] < 20, tr
    assert tr["launches"] is None or 50 < tr["launches"] < 600, tr
    assert abs(tr["mfma_frac"] - tr["tflops"] / 157.3) < 1e-3 and 0 < tr["mfma_frac"] < 1, tr
    for k in ("fps_ms", "ball_ms_r01", "ball_ms_r02", "sa_ms", "level_ms", "level_ms_pipelined"):
        assert "error" not in st_ and 0 < st_[k] < 100, (k, st_)
    assert st_["pipelined_equals_serial"] is True and 0 < st_["sa_mfma_frac"] < 1
    assert "error" not in lat and 0 < lat["graph_ms"] < 5 and lat["max_abs_diff_vs_eager"] <= 1e-5, lat
