set -x
O=gpurun_out/r05g; mkdir -p $O
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
(time timeout 1200 python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
HOTRACK_TGB2=1 python scripts/probes/tg_bench.py > $O/tg_bench_v2.json 2> $O/tg_bench_v2.err
python - <<'PY'
import json
b=json.load(open("gpurun_out/r05g/tg_bench_v2.json"))["shapes"]
for k in b: print(k, "v2", b[k]["fused"])
PY
HOTRACK_TGB2=0 python scripts/bench_train.py --graph > $O/train_v1.json 2> $O/train_v1.err; cat $O/train_v1.json
HOTRACK_TGB2=1 python scripts/bench_train.py --graph > $O/train_v2.json 2> $O/train_v2.err; cat $O/train_v2.json
PN2_LIB_PATH=$PWD/hotrack_amd/libpn2_hip.tgbprof.so python scripts/probes/tgb_profile.py > $O/tgb_profile_v2.json 2> $O/tgb_profile_v2.err; tail -3 $O/tgb_profile_v2.err
python -c "
import json
d=json.load(open('$O/tgb_profile_v2.json'))
for k,v in d.items(): print(k, v.get('prologue_parts_waves_0_3_7'), v['prologue'], v['commit'], v['total'])
"
