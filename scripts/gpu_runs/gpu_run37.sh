# layer backward kernels with unconditional next-tile requests: parity, step time, kernel times inside the replayed step
python -m pytest tests/test_gpu_train.py tests/test_gpu_train_full.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do python scripts/bench_train.py --graph 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
python scripts/probes/tg_bench.py 2>/dev/null | tail -30
