R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06h; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py tests/test_gpu_modules.py -x -q 2>&1 | tail -3
python scripts/bench_ops.py --out $O/bench_ops.json > /dev/null 2>&1
python - <<'PY'
import json
for r in json.load(open("gpurun_out/r06h/bench_ops.json")):
    if r.get("op") in ("group_bwd", "interp_bwd"):
        print({k: r[k] for k in ("op", "us", "GBps", "B", "C") if k in r}, {k: r[k] for k in ("N", "P", "S", "M", "n") if k in r})
PY
