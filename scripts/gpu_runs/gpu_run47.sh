k() { python bench.py --no-cpu-baseline --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step']); [print('  ',k['kernel'][:50], k['frac'], k['us_per_launch']) for k in d['kernels'] if 'fps' not in k['kernel']]"; }
echo "H1 ring of 3 tiles:"; PN2_LIB_PATH=$PWD/hotrack_amd/libpn2_hip.nb3.so k
echo "ring of 2 (product):"; k
