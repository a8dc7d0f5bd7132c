# after the last training commits: refresh the training numbers of the evidence set (the inference path did not change)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
for i in 1 2 3; do python scripts/bench_train.py --graph 2>/dev/null | grep '^{' > $O/r06_dp_none_$i.json; python scripts/bench_train.py --graph --dp-selftest 2>/dev/null | grep '^{' > $O/r06_dp_seg1_$i.json; done
cp $O/r06_dp_none_1.json $O/r06_bench_train_graph.json
python scripts/probes/train_step_parts.py 2>/dev/null | grep "ms / iteration\|full step" > $O/r06_train_step_parts.txt
python scripts/probes/train_step_parts.py --dp 2>/dev/null | grep "ms / iteration\|full step" >> $O/r06_train_step_parts.txt
python bench.py > $O/r06_bench.json 2> $O/r06_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ts && rocprofv3 --kernel-trace --output-format csv -d /tmp/ts -o t -- python $R/scripts/bench_train.py --graph --steps 12 --warmup 4 > /dev/null 2>&1
python $R/scripts/trace_one_step.py $(find /tmp/ts -name "*kernel_trace.csv" | head -1) > $O/r06_train_one_step.csv
python $R/scripts/trace_window.py $(find /tmp/ts -name "*kernel_trace.csv" | head -1) --frac 0.3 --steps-in-window 0 > $O/r06_train_graph_window.csv
head -1 $O/r06_train_one_step.csv; grep -h -o '"ms_per_step": [0-9.]*' $O/r06_dp_none_*.json $O/r06_dp_seg1_*.json
