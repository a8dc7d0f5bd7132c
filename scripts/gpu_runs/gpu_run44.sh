one() { python bench.py --no-cpu-baseline --no-legs "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['config'].get('batches_in_flight'), d['config'].get('hip_hw_queues'))"; }
one --inflight 4
one --inflight 5
one --inflight 6
GPU_MAX_HW_QUEUES=12 one --inflight 6
GPU_MAX_HW_QUEUES=12 one --inflight 8
one --inflight 3
