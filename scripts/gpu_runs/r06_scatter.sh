R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_ops.py -x -q -k "group or gather or interpolate" 2>&1 | tail -3
python scripts/bench_ops.py --out $O/bench_ops.json > /dev/null 2>&1
python - <<'PY'
import json
for r in json.load(open("gpurun_out/r06d/bench_ops.json")):
    if r.get("op") in ("group_bwd", "interp_bwd", "group_fwd"):
        print(r)
PY
for i in 1 2; do python scripts/bench_train.py --graph 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
python -m pytest tests/test_gpu_train.py tests/test_gpu_train_full.py -x -q 2>&1 | tail -2
