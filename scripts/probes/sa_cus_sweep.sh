for n in 256 248 240 224 208; do
PN2_SA_CUS=$n python bench.py --no-cpu-baseline --min-time 1.5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('PN2_SA_CUS=$n', d['value'], d['ms_per_step'], d['config']['single_stream_ms_per_step'])"
done
