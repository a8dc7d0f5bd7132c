set -x
mkdir -p gpurun_out/r05a
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
(time python -m pytest tests -m gpu -x -q) > gpurun_out/r05a/pytest.log 2>&1; tail -3 gpurun_out/r05a/pytest.log
python scripts/bench_train.py --graph > gpurun_out/r05a/train_base.json 2> gpurun_out/r05a/train_base.err; cat gpurun_out/r05a/train_base.json
python scripts/bench_train.py --graph --dp-selftest > gpurun_out/r05a/train_seg2.json 2> gpurun_out/r05a/train_seg2.err; cat gpurun_out/r05a/train_seg2.json; tail -3 gpurun_out/r05a/train_seg2.err
python scripts/bench_train.py --graph --dp-selftest --segments 1 > gpurun_out/r05a/train_seg1.json 2> gpurun_out/r05a/train_seg1.err; cat gpurun_out/r05a/train_seg1.json
python scripts/bench_train.py --graph --dp-selftest --no-overlap > gpurun_out/r05a/train_seg2_noov.json 2> gpurun_out/r05a/train_seg2_noov.err; cat gpurun_out/r05a/train_seg2_noov.json
PN2_LIB_PATH=$PWD/hotrack_amd/libpn2_hip.tgbprof.so python scripts/probes/tgb_profile.py > gpurun_out/r05a/tgb_profile.json 2> gpurun_out/r05a/tgb_profile.err; tail -3 gpurun_out/r05a/tgb_profile.err
(time bash scripts/scale_selftest.sh 2) > gpurun_out/r05a/scale_selftest.log 2>&1; tail -8 gpurun_out/r05a/scale_selftest.log
python bench.py > gpurun_out/r05a/bench.json 2> gpurun_out/r05a/bench.err; cut -c1-600 gpurun_out/r05a/bench.json
