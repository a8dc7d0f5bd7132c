python -m pytest tests/test_gpu_fused.py -x -q -m gpu -k "ln_linear or tail" 2>&1 | tail -2
echo "folded:"; python scripts/bench_latency.py 2>/dev/null | tail -1 | cut -c1-200
echo "two launches (HOTRACK_LN_LINEAR_MAX_ROWS=0):"; HOTRACK_LN_LINEAR_MAX_ROWS=0 python scripts/bench_latency.py 2>/dev/null | tail -1 | cut -c1-200
echo "folded:"; python scripts/bench_latency.py 2>/dev/null | tail -1 | cut -c1-200
