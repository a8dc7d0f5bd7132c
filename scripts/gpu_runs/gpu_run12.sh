set -x
O=gpurun_out/r05m; mkdir -p $O
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
python scripts/probes/dbg_wgrad.py 2>&1 | grep -v amdgpu.ids
(time timeout 900 python -m pytest tests/test_gpu_wgrad.py -q) > $O/pytest_wgrad.log 2>&1; tail -15 $O/pytest_wgrad.log
python scripts/probes/wgrad_bench.py > $O/wgrad_bench_nbuf1.json 2>$O/wgrad_bench.err; cat $O/wgrad_bench_nbuf1.json; tail -3 $O/wgrad_bench.err
PN2_LIB_PATH=$PWD/hotrack_amd/libpn2_hip.wgmsame.so python scripts/probes/wgrad_bench.py > $O/wgrad_bench_same.json 2>>$O/wgrad_bench.err; cat $O/wgrad_bench_same.json
(time timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_full.py tests/test_ddp_gloo.py -x -q) > $O/pytest_train.log 2>&1; tail -8 $O/pytest_train.log
python scripts/bench_train.py --graph 2>$O/train.err | grep '^{' > $O/train.json; cut -c1-330 $O/train.json
