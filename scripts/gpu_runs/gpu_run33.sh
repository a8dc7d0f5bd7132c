O=gpurun_out/r05z; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_fused.py -q -k "linear_k128") 2>&1 | tail -4
python scripts/probes/linear_k128_bench.py 2>/dev/null | tee $O/linear_k128_bench.json
HOTRACK_LINEAR_K128_MIN_ROWS=0 python bench.py --no-cpu-baseline --no-legs 2>/dev/null | cut -c1-200
python bench.py --no-cpu-baseline --no-legs 2>/dev/null | cut -c1-200
