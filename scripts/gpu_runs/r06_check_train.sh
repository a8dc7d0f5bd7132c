R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python -m pytest tests/test_gpu_train.py tests/test_gpu_train_full.py tests/test_gpu_wgrad.py -x -q 2>&1 | tail -4
for i in 1 2; do python scripts/bench_train.py --graph 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
python scripts/bench_train.py --graph --dp-selftest 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print({k:r[k] for k in ('ms_per_step','bwd_segments','allreduce_us','segment_bytes','moved_bytes','launches')})"
