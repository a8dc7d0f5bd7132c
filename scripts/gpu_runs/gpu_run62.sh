python -m pytest tests/test_gpu_fused.py tests/test_tracking_golden.py tests/test_gpu_tracking.py -x -q -m gpu 2>&1 | tail -2
python scripts/bench_latency.py 2>/dev/null | tail -1 | cut -c1-200
