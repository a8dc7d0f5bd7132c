# builds sa_fused.hip with each flag set ON THE GPU BOX and prints bench.py's SA kernel timings
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  touch $R/hotrack_amd/csrc/sa_fused.hip
  (cd $R && PN2_EXTRA_HIPCC_FLAGS="$cfg" python -c "from hotrack_amd import _build; _build.build()" > /dev/null)
  echo "== ${cfg:-default}"
  for i in 1 2; do (cd $R && python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['single_stream_ms_per_step'], [(k['kernel'][18:50], k['us_per_launch']) for k in d['kernels'] if 'sa_mlp' in k['kernel']])"); done
done
