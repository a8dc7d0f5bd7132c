(timeout 600 python -m pytest tests/test_gpu_wgrad.py -q) 2>&1 | tail -3
