# round 6, first call: the state of round 5 on today's box (training step, dp = flat mechanism at one rank, one-step listing)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R
python scripts/bench_train.py --graph > $O/train_graph.json 2>$O/train_graph.err; cat $O/train_graph.json
python scripts/bench_train.py --graph --dp-selftest --segments 1 2>/dev/null | grep '^{' > $O/dp_seg1.json; cut -c1-200 $O/dp_seg1.json
python scripts/bench_train.py --graph --dp-selftest 2>/dev/null | grep '^{' > $O/dp_seg2.json; cut -c1-200 $O/dp_seg2.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ts && rocprofv3 --kernel-trace --output-format csv -d /tmp/ts -o t -- python $R/scripts/bench_train.py --graph --steps 12 --warmup 4 > /dev/null 2>&1
python $R/scripts/trace_one_step.py $(find /tmp/ts -name "*kernel_trace.csv" | head -1) > $O/train_one_step.csv
wc -l $O/train_one_step.csv
python -c "import torch; print(torch.__version__); import torch.distributed as d; print(hasattr(d,'ProcessGroupNCCL'))"
