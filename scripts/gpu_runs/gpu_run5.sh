set -x
O=gpurun_out/r05e; mkdir -p $O
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
PN2_LIB_PATH=$PWD/hotrack_amd/libpn2_hip.tgbprof.so python scripts/probes/tgb_profile.py > $O/tgb_profile_v2.json 2> $O/tgb_profile_v2.err; tail -3 $O/tgb_profile_v2.err
python -c "
import json
d=json.load(open('$O/tgb_profile_v2.json'))
for k,v in d.items(): print(k, v.get('prologue_parts_waves_0_3_7'), v['prologue'], v['total'])
"
python scripts/bench_legs.py stress > $O/stress.json 2> $O/stress.err; cat $O/stress.json; tail -3 $O/stress.err
