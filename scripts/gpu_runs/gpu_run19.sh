O=gpurun_out/r05s; mkdir -p $O
R=$PWD
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
(timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_full.py tests/test_gpu_wgrad.py tests/test_ddp_gloo.py -x -q) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
HOTRACK_PAIR_SCALES=0 HOTRACK_STACK_PAIR_LAUNCH=0 python scripts/bench_train.py --graph 2>/dev/null | grep '^{' > $O/train_nopair.json; cut -c1-330 $O/train_nopair.json
python scripts/bench_train.py --graph 2>$O/train.err | grep '^{' > $O/train.json; cut -c1-330 $O/train.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ts && rocprofv3 --kernel-trace --output-format csv -d /tmp/ts -o t -- python $R/scripts/bench_train.py --graph --steps 12 --warmup 4 > /dev/null 2>&1
python $R/scripts/trace_one_step.py $(find /tmp/ts -name "*kernel_trace.csv" | head -1) > $R/$O/train_one_step.csv
head -1 $R/$O/train_one_step.csv
