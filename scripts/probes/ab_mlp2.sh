# A/B on the GPU box: the fused fp1 pair (pn2x_mlp2_rows) against the two library GEMMs, three alternating bench runs each
for i in 1 2 3; do
for v in 0 1; do
HOTRACK_MLP2=$v python bench.py --no-cpu-baseline --min-time 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('HOTRACK_MLP2=$v', d['value'], d['ms_per_step'], d['config']['single_stream_ms_per_step'])"
done; done
