R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py tests/test_gpu_train.py tests/test_gpu_train_full.py -x -q 2>&1 | tail -3
python scripts/probes/scatter_cm_bench.py
for i in 1 2 3; do python scripts/bench_train.py --graph 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
python scripts/probes/train_step_parts.py 2>/dev/null | grep "ms / iteration"
