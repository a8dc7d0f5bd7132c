export HOTRACK_DATA_ROOT=/tmp/hotrack_data
for v in 32 64 128 16; do echo "slice $v: $(PN2_TGR_SLICE=$v python scripts/bench_train.py --graph 2>/dev/null | grep '^{' | cut -c100-260)"; done
cd /tmp && export TMPDIR=/tmp
for v in 32 128; do rm -rf /tmp/ts && PN2_TGR_SLICE=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ts -o t -- python $GRAFT_REPO_ROOT/scripts/bench_train.py --graph --steps 12 --warmup 4 > /dev/null 2>&1; echo "slice $v: $(grep tg_reduce_multi $(find /tmp/ts -name '*kernel_stats.csv' | head -1) | cut -c1-120)"; done
