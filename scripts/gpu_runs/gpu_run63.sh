python -m pytest tests/test_gpu_fused.py tests/test_tracking_golden.py tests/test_gpu_tracking.py -x -q -m gpu --durations=6 2>&1 | tail -12
