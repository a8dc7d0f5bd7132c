python -m pytest tests/test_gpu_fused.py tests/test_gpu_tracking.py -x -q -m gpu 2>&1 | tail -2
PN2_LIB_PATH=hotrack_amd/libpn2_hip.satrace.so python scripts/probes/sa1_trace.py 2>/dev/null | sed -n 1,8p
python scripts/probes/mlp2_bench.py 2>/dev/null | head -1
for i in 1 2; do python bench.py --no-cpu-baseline --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step']); [print('  ',k['kernel'][:50], k['frac'], k['us_per_launch']) for k in d['kernels'] if 'fps' not in k['kernel']]"; done
