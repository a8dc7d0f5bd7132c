export HOTRACK_DATA_ROOT=/tmp/hotrack_data
(timeout 600 python -m pytest tests/test_gpu_train.py -q -k "adam or Adam or optim") 2>&1 | tail -3
python scripts/bench_train.py --graph 2>/dev/null | grep '^{' | cut -c100-260
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ts && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ts -o t -- python $GRAFT_REPO_ROOT/scripts/bench_train.py --graph --steps 12 --warmup 4 > /dev/null 2>&1; grep "adam" $(find /tmp/ts -name '*kernel_stats.csv' | head -1) | cut -c1-120
