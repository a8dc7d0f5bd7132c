cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tk && rocprofv3 --kernel-trace --output-format csv -d /tmp/tk -o t -- python $GRAFT_REPO_ROOT/scripts/probes/linear_k128_bench.py > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/tk/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'linear_k128' in n or 'Cijk' in n:
        agg[(n[:60], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
for k,v in sorted(agg.items()):
    v=sorted(v); print(k, len(v), 'median %.1f min %.1f'%(v[len(v)//2], v[0]))
PY
