set -x
O=gpurun_out/r05c; mkdir -p $O
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
(time timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_full.py -x -q) > $O/pytest_train.log 2>&1; tail -5 $O/pytest_train.log
HOTRACK_TGB2=0 python scripts/probes/tg_bench.py > $O/tg_bench_v1.json 2> $O/tg_bench_v1.err
HOTRACK_TGB2=1 python scripts/probes/tg_bench.py > $O/tg_bench_v2.json 2> $O/tg_bench_v2.err
python - <<'PY'
import json
a=json.load(open("gpurun_out/r05c/tg_bench_v1.json"))["shapes"]; b=json.load(open("gpurun_out/r05c/tg_bench_v2.json"))["shapes"]
for k in a: print(k, "v1", a[k]["fused"], "v2", b[k]["fused"])
PY
HOTRACK_TGB2=0 python scripts/bench_train.py --graph > $O/train_v1.json 2> $O/train_v1.err; cat $O/train_v1.json
HOTRACK_TGB2=1 python scripts/bench_train.py --graph > $O/train_v2.json 2> $O/train_v2.err; cat $O/train_v2.json
PN2_LIB_PATH=$PWD/hotrack_amd/libpn2_hip.tgbprof.so python scripts/probes/tgb_profile.py > $O/tgb_profile_v2.json 2> $O/tgb_profile_v2.err; tail -3 $O/tgb_profile_v2.err
