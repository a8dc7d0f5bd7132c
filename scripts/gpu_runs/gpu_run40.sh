python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "fps_streamed" 2>&1 | grep -v "^$" | head -60
