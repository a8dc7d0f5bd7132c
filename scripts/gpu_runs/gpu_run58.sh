python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "fps" 2>&1 | tail -2
echo "resident (registers + LDS):"; python scripts/probes/fps_stream_bench.py 2>/dev/null | grep -v amdgpu | tr -d '\n' | sed 's/  */ /g'; echo
echo "plain streamed:"; PN2_FPS_STREAM_PLAIN=1 python scripts/probes/fps_stream_bench.py 2>/dev/null | grep -v amdgpu | tr -d '\n' | sed 's/  */ /g'; echo
