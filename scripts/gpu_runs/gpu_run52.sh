python -m pytest tests/test_gpu_fused.py tests/test_tracking_golden.py tests/test_gpu_tracking.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do python scripts/bench_latency.py 2>/dev/null | tail -1 | cut -c1-200; done
python bench.py --no-cpu-baseline --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"
