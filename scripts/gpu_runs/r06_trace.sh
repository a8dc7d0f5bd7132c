R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r06e}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ts && rocprofv3 --kernel-trace --output-format csv -d /tmp/ts -o t -- python $R/scripts/bench_train.py --graph --steps 12 --warmup 4 > /dev/null 2>&1
python $R/scripts/trace_one_step.py $(find /tmp/ts -name "*kernel_trace.csv" | head -1) > $O/train_one_step.csv
head -1 $O/train_one_step.csv
cd $R; for i in 1 2; do python scripts/bench_train.py --graph 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
