#!/bin/bash
# usage (GPU box, repo root): scripts/evidence_round.sh rNN  -> gpurun_out/rNN_*: the round's measurements in one call
#   pytest -m gpu summary, profile_round.sh (bench line + rocprofv3 kernel stats + PMC traffic / MfmaUtil), per-operator,
#   stress, latency, SDF and training benches, the training step kernel by kernel, the multi-rank self-test.
set -u
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/${tag}_pytest_gpu.log
bash scripts/profile_round.sh $tag > $O/${tag}_profile_round.log 2>&1
python scripts/bench_ops.py --out $O/${tag}_bench_ops.json > /dev/null 2>&1 || python scripts/bench_ops.py > $O/${tag}_bench_ops.json 2>/dev/null
python scripts/bench_stress.py --out $O/${tag}_stress.json > $O/${tag}_stress.log 2>&1
python scripts/bench_latency.py > $O/${tag}_bench_latency.json 2>/dev/null
python scripts/bench_train.py --graph > $O/${tag}_bench_train_graph.json 2>/dev/null
HOTRACK_FUSED_STACKS=0 python scripts/bench_train.py --graph > $O/${tag}_bench_train_graph_unfused_stacks.json 2>/dev/null
python scripts/bench_train.py > $O/${tag}_bench_train_eager.json 2>/dev/null
python scripts/probes/tg_bench.py --iters 10 > $O/${tag}_tg_bench.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ts && rocprofv3 --kernel-trace --output-format csv -d /tmp/ts -o t -- python $R/scripts/bench_train.py --graph --steps 12 --warmup 4 > /dev/null 2>&1
python $R/scripts/trace_one_step.py $(find /tmp/ts -name "*kernel_trace.csv" | head -1) > $O/${tag}_train_one_step.csv
python $R/scripts/trace_window.py $(find /tmp/ts -name "*kernel_trace.csv" | head -1) --frac 0.3 --steps-in-window 0 > $O/${tag}_train_graph_window.csv
rm -rf /tmp/tst && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tst -o t -- python $R/scripts/bench_stress.py > /dev/null 2>&1
cp $(find /tmp/tst -name "*kernel_stats.csv" | head -1) $O/${tag}_stress_kernel_stats.csv
rm -rf /tmp/tl && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $R/scripts/bench_legs.py latency > /dev/null 2>&1
python $R/scripts/trace_graph_replays.py $(find /tmp/tl -name "*kernel_trace.csv" | head -1) --marker hand_frame_kernel > $O/${tag}_latency_b1_replay.csv 2>&1
cd $R
python scripts/probes/hand_frame_timing.py 2>/dev/null | grep hand_frame > $O/${tag}_hand_frame_timing.txt
python scripts/probes/two_graph_overlap.py 2>/dev/null | grep "ms / step" > $O/${tag}_two_graph_overlap.txt
python scripts/probes/train_host_timing.py 2>/dev/null | grep "step " > $O/${tag}_train_prefetch_vs_inline.txt
HOTRACK_STACK_PAIR_LAUNCH=0 python scripts/bench_train.py --graph > $O/${tag}_bench_train_graph_no_pair_launch.json 2>/dev/null
HOTRACK_PREFETCH_GEOMETRY=0 python scripts/bench_train.py --graph > $O/${tag}_bench_train_graph_no_prefetch.json 2>/dev/null
python scripts/bench_legs.py stress > $O/${tag}_stress_leg.json 2>/dev/null
python scripts/bench_legs.py latency > $O/${tag}_latency_leg.json 2>/dev/null
python scripts/kernel_resources.py > $O/${tag}_kernel_resources.txt 2>&1
bash scripts/scale_selftest.sh 2 > $O/${tag}_scale_selftest.log 2>&1
# dp = flat with a one-rank RCCL group: what the exchange mechanism costs beside the single-graph step
for i in 1 2 3; do
python scripts/bench_train.py --graph 2>/dev/null | grep '^{' > $O/${tag}_dp_none_$i.json
python scripts/bench_train.py --graph --dp-selftest 2>/dev/null | grep '^{' > $O/${tag}_dp_seg1_$i.json
done
python scripts/bench_train.py --graph --dp-selftest --segments 2 --no-overlap 2>/dev/null | grep '^{' > $O/${tag}_dp_seg2_inorder.json
python scripts/bench_train.py --graph --dp-selftest --segments 2 2>/dev/null | grep '^{' > $O/${tag}_dp_seg2_overlap.json
python scripts/probes/train_step_parts.py 2>/dev/null | grep "ms / iteration\|full step" > $O/${tag}_train_step_parts.txt
python scripts/probes/train_step_parts.py --dp 2>/dev/null | grep "ms / iteration\|full step" >> $O/${tag}_train_step_parts.txt
python scripts/probes/scatter_cm_bench.py > $O/${tag}_scatter_cm_bench.txt 2>/dev/null
tail -3 $O/${tag}_pytest_gpu.log; cat $O/${tag}_bench_train_graph.json; tail -2 $O/${tag}_scale_selftest.log
