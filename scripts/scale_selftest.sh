#!/bin/bash
# Multi-rank plumbing self-test on ONE GPU (no 8-GPU node is available to the builder; the driver runs the real scaling
# bench).  Launches bench.py and network/train.py exactly as the driver / a user would -- torch.distributed.run, one process
# per rank, 127.0.0.1 rendezvous -- but with the gloo backend so that N ranks can share device 0 (RCCL refuses two ranks on
# one device).  Checks: the JSON line says n_gpus == N == world_size, value aggregates all ranks, training runs N ranks and
# every rank ends with identical parameters.   usage: scripts/scale_selftest.sh [N ...]   (default: 2 4)
set -euo pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
Ns=${@:-2 4}
export HOTRACK_DATA_ROOT=${HOTRACK_DATA_ROOT:-/tmp/hotrack_selftest}
one=$(python bench.py --gpus 1 --steps 5 --warmup 2 --min-time 0.2 --no-cpu-baseline | grep '^{')
for N in $Ns; do
  port=$((29600 + N))
  line=$(PN2_BENCH_BACKEND=gloo PN2_BENCH_LEG_TIMEOUT=240 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
         --master-port $port bench.py --gpus $N --steps 5 --warmup 2 --min-time 0.2 --train-steps 5 --no-cpu-baseline | grep '^{')
  python - "$N" "$line" "$one" <<'PY'
import json, sys
n, d, one = int(sys.argv[1]), json.loads(sys.argv[2]), json.loads(sys.argv[3])
assert d["n_gpus"] == n and d["config"]["world_size"] == n, d
assert d["config"]["global_batch"] == n * d["config"]["per_gpu_batch"] and d["scaling"] == "weak"
assert abs(d["value"] - d["config"]["global_batch"] * 1e3 / d["ms_per_step"]) < 0.01 * d["value"]
# the multi-GPU extras: per-rank spread of the headline and the data-parallel training leg that shows the gradient all-reduce
pr, tr = d["per_rank_frames_per_s"], d["train"]
assert "error" not in tr, tr
assert len(pr["ranks"]) == n and pr["min"] <= pr["max"]
assert d["world_size_seen_by_backend"] == n, d
assert tr["n_gpus"] == n and tr["world_size_seen_by_backend"] == n and tr["dp_mode"] == "flat" and tr["graph_step"] is True, tr
assert tr["backend"] == "gloo" and tr["bytes"] > 16e6 and tr["allreduce_us"] > 0 and tr["ms_per_step"] > 0, tr
assert d["replay_check"]["max_abs_diff_pred_kp"] <= 1e-5 and d["gemm_table"] in ("applied", "stale", "off")
# N ranks time-share one GPU here, so the aggregate stays near the 1-rank number (it must NOT be N times smaller or larger)
print(f"bench --gpus {n}: ok  n_gpus={d['n_gpus']} world_size={d['config']['world_size']} value={d['value']:.0f} frames/s "
      f"(1 rank on the same GPU: {one['value']:.0f})")
PY
done
# a training leg that dies on ONE rank (its peers then wait for it until the leg's timeout): the headline must survive, the
# line must carry train.error, and the process group of the parents must still shut down cleanly (exit code 0)
line=$(PN2_BENCH_BACKEND=gloo PN2_BENCH_FAIL_TRAIN_LEG=1 PN2_BENCH_LEG_TIMEOUT=90 python -m torch.distributed.run --nnodes=1 \
       --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29650 bench.py --gpus 2 --steps 5 --warmup 2 --min-time 0.2 \
       --train-steps 5 --no-cpu-baseline | grep '^{')
python - "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[1])
assert d["n_gpus"] == 2 and d["value"] > 0 and d["replay_check"]["max_abs_diff_pred_kp"] <= 1e-5, d
assert "error" in d["train"], d["train"]
print("bench --gpus 2 with a failing training leg: headline survives (%.0f frames/s), train = %s" % (d["value"], d["train"]))
PY
for N in 2; do
  PN2_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
      --master-port $((29700 + N)) network/train.py --config handtracknet_train_SimGrasp.yml --num_points 512 --batch_size 4 \
      --total_epoch 1 --synthetic_frames 32 --max_iters 3 2>&1 | tee /tmp/selftest_train_$N.log | grep -E "world_size|Train total_loss" || true
  grep -q "world_size $N, 3 iterations" /tmp/selftest_train_$N.log || { echo "train.py did not run $N ranks"; tail -20 /tmp/selftest_train_$N.log; exit 1; }
  echo "train.py x$N ranks: ok"
done
# the graph-captured data-parallel training step (forward+backward graph | flat all-reduce | Adam graph), 2 ranks on one GPU
PN2_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29801 \
    scripts/bench_train.py --steps 10 --warmup 5 --batch 8 --graph 2>&1 | grep '^{' | tee /tmp/selftest_bench_train.json
python - <<'PY'
import json
d = json.loads(open("/tmp/selftest_bench_train.json").read())
assert d["n_gpus"] == 2 and d["graph_step"] is True and d["dp_mode"] == "flat", d
print("bench_train --graph x2 ranks: ok (%.2f ms/step on a shared GPU)" % d["ms_per_step"])
PY
echo "scale_selftest: all ok"
