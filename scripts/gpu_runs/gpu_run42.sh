one() { python bench.py --no-cpu-baseline --no-legs "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['config'].get('geometry_on_priority_stream'), d['config'].get('hip_hw_queues'))"; }
echo "split, 16 queues:"; GPU_MAX_HW_QUEUES=16 one --split-geometry
echo "split, 12 queues:"; GPU_MAX_HW_QUEUES=12 one --split-geometry
echo "unsplit, 16 queues:"; GPU_MAX_HW_QUEUES=16 one
echo "split, inflight 2:"; one --split-geometry --inflight 2
echo "split, inflight 3:"; one --split-geometry --inflight 3
echo "unsplit, inflight 2:"; one --inflight 2
