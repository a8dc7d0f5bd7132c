python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "fps" 2>&1 | tail -3
python scripts/probes/fps_stream_bench.py
