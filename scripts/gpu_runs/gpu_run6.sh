set -x
O=gpurun_out/r05f; mkdir -p $O
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
for v in tgbprof tgbnow tgblate; do
PN2_LIB_PATH=$PWD/hotrack_amd/libpn2_hip.$v.so python scripts/probes/tgb_profile.py > $O/tgb_profile_$v.json 2> $O/tgb_profile_$v.err
python -c "
import json
d=json.load(open('$O/tgb_profile_$v.json'))
for k,v in d.items(): print('$v', k, v.get('prologue_parts_waves_0_3_7'), v['prologue'], v['commit'], v['total'])
"
done
python scripts/bench_legs.py stress > $O/stress.json 2> $O/stress.err; cat $O/stress.json; tail -3 $O/stress.err
