set -x
O=gpurun_out/r05n; mkdir -p $O
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
(time timeout 900 python -m pytest tests/test_gpu_wgrad.py -q) > $O/pytest_wgrad.log 2>&1; tail -15 $O/pytest_wgrad.log
python scripts/probes/wgrad_bench.py > $O/wgrad_bench_nbuf1.json 2>$O/wgrad_bench.err; cat $O/wgrad_bench_nbuf1.json; tail -3 $O/wgrad_bench.err
PN2_LIB_PATH=$PWD/hotrack_amd/libpn2_hip.wgm2.so python scripts/probes/wgrad_bench.py > $O/wgrad_bench_nbuf2.json 2>>$O/wgrad_bench.err; cat $O/wgrad_bench_nbuf2.json
PN2_LIB_PATH=$PWD/hotrack_amd/libpn2_hip.wgmsame.so python scripts/probes/wgrad_bench.py > $O/wgrad_bench_same.json 2>>$O/wgrad_bench.err; cat $O/wgrad_bench_same.json
python scripts/bench_train.py --graph 2>$O/train.err | grep '^{' > $O/train.json; cut -c1-330 $O/train.json
