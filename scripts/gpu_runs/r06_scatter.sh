R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py -x -q -k "group or gather or interpolate or fuzz" 2>&1 | tail -3
python scripts/bench_ops.py --out $O/bench_ops.json > /dev/null 2>&1
python - <<'PY'
import json
for r in json.load(open("gpurun_out/r06d/bench_ops.json")):
    if r.get("op") in ("group_bwd", "interp_bwd"):
        print(r)
PY
