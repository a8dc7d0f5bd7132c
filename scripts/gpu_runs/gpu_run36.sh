# fused forward stack kernel with the peeled, branch-free tile loop: parity, step time, per-kernel time
python -m pytest tests/test_gpu_train.py tests/test_gpu_train_full.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do python scripts/bench_train.py --graph 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tk && rocprofv3 --kernel-trace --output-format csv -d /tmp/tk -o t -- python $GRAFT_REPO_ROOT/scripts/bench_train.py --graph --steps 10 > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/tk/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    agg[n[:70]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
tot=sum(sum(v) for v in agg.values())
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:24]:
    print('%-72s n=%5d total %9.1f us  avg %7.1f  %.1f%%'%(k,len(v),sum(v),sum(v)/len(v),100*sum(v)/tot))
PY
