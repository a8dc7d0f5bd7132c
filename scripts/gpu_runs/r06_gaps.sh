# round 6: where the training step's stream sits idle (graph-launch boundaries), three configurations
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name, env..., -- args
  name=$1; shift
  rm -rf /tmp/ts_$name
  env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/ts_$name -o t -- python $R/scripts/bench_train.py --graph --steps 12 --warmup 4 $EXTRA > /tmp/$name.log 2>&1
  f=$(find /tmp/ts_$name -name "*kernel_trace.csv" | head -1)
  echo "== $name: $(grep -o '"ms_per_step": [0-9.]*' /tmp/$name.log)"
  python $R/scripts/trace_gaps.py $f --steps 3 > $O/gaps_$name.txt 2>&1; cat $O/gaps_$name.txt
}
EXTRA="" run prefetch A=1
EXTRA="" run inline HOTRACK_PREFETCH_GEOMETRY=0
EXTRA="--dp-selftest --segments 1" run dp_seg1 A=1
EXTRA="--dp-selftest" run dp_seg2 A=1
