# mlp2_rows with transposed layer-3 product (16-byte row stores): parity, kernel time, headline
python -m pytest tests/test_gpu_fused.py -x -q -m gpu -k "mlp2 or fused or sa_" 2>&1 | tail -2
python scripts/probes/mlp2_bench.py
python bench.py --no-cpu-baseline --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step']); [print(k['kernel'][:50], k['frac'], k['us_per_launch']) for k in d['kernels']]"
