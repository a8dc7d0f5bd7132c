R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o t -- python $R/scripts/probes/scatter_cm_bench.py > /tmp/sp.log 2>&1
cat /tmp/sp.log | tail -5
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/sp/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "chunked" in n or "inverse_index" in n:
        d[(n[:60], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
for k, v in d.items():
    v.sort()
    print(k, "n", len(v), "median us", round(v[len(v)//2], 1), "min", round(v[0], 1))
PY
