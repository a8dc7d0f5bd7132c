O=gpurun_out/r05r; mkdir -p $O
R=$PWD
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
(timeout 1200 python -m pytest tests/test_gpu_wgrad.py tests/test_gpu_train.py tests/test_gpu_train_full.py tests/test_ddp_gloo.py -x -q) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python scripts/bench_train.py --graph 2>$O/train.err | grep '^{' > $O/train.json; cut -c1-330 $O/train.json
