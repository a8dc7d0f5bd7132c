python -m pytest tests/test_gpu_fused.py tests/test_gpu_tracking.py tests/test_tracking_golden.py -x -q -m gpu 2>&1 | tail -3
python scripts/bench_latency.py 2>/dev/null | tail -2 | cut -c1-600
