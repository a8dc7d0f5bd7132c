cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
export PN2_TGB2_TWO_PER_CU=$v
echo "two per CU = $v: $(python $GRAFT_REPO_ROOT/scripts/bench_train.py --graph 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
rm -rf /tmp/ts && rocprofv3 --kernel-trace --output-format csv -d /tmp/ts -o t -- python $GRAFT_REPO_ROOT/scripts/bench_train.py --graph --steps 12 --warmup 4 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/trace_one_step.py $(find /tmp/ts -name "*kernel_trace.csv" | head -1) | grep -E "tg_bwd2_kernel<2|tg_reduce_multi"
done
