# builds sdf.hip with each flag set ON THE GPU BOX and prints rocprofv3 kernel averages (us) for bench_sdf.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  touch $R/hotrack_amd/csrc/sdf.hip
  (cd $R && PN2_EXTRA_HIPCC_FLAGS="$cfg" python -c "from hotrack_amd import _build; _build.build()" > /dev/null)
  echo "== ${cfg:-default}"
  rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/scripts/bench_sdf.py --no-cpu > /dev/null 2>&1
  f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "pn2::" in n: print("  %-34s calls %4s avg %8.1f us  min %8.1f" % (n.split("pn2::")[1].split("(")[0][:34], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
