# headline: one graph per batch in flight vs geometry prefix as its own graph on a high-priority stream
one() { python bench.py --no-cpu-baseline --no-legs "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['config'].get('geometry_on_priority_stream'), d.get('replay_check'))"; }
one
one --split-geometry
one --split-geometry --sa-cus 240
one --split-geometry --sa-cus 224
one
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tk && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tk -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-legs --split-geometry > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/tk/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('fps_knn','hand_frame','ball_tie','fps_kernel')): print(r['Name'][:40], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
