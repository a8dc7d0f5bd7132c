# fps level 1 with the winner's coordinates in the exchange entry (default) vs read from the LDS copy of the cloud (PN2_FPS_CENT_LDS=1)
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fused.py -x -q -m gpu -k "fps or two_level or fast_point or knn" 2>&1 | tail -2
for v in entry lds entry lds; do
  if [ $v = lds ]; then export PN2_FPS_CENT_LDS=1; else unset PN2_FPS_CENT_LDS; fi
  echo "$v: $(python scripts/bench_latency.py 2>/dev/null | tail -1 | cut -c1-120)"
done
unset PN2_FPS_CENT_LDS
python bench.py --no-cpu-baseline --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step']); [print('  ',k['kernel'][:50], k['us_per_launch']) for k in d['kernels'] if 'fps' in k['kernel']]"
