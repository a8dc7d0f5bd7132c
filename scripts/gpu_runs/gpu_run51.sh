cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tk && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tk -o t -- python $GRAFT_REPO_ROOT/scripts/bench_latency.py > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/tk/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('ln_linear','linear_small','add_layernorm','MT16x16x64','pose_head')): print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
