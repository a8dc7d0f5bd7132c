set -x
O=gpurun_out/r05k; mkdir -p $O
R=$PWD
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
(time timeout 900 python -m pytest tests/test_gpu_wgrad.py -x -q) > $O/pytest_wgrad.log 2>&1; tail -15 $O/pytest_wgrad.log
(time timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_full.py tests/test_ddp_gloo.py -x -q) > $O/pytest_train.log 2>&1; tail -8 $O/pytest_train.log
HOTRACK_DEFER_WGRAD=0 python scripts/bench_train.py --graph 2>/dev/null | grep '^{' > $O/train_nodefer.json; cut -c1-330 $O/train_nodefer.json
python scripts/bench_train.py --graph 2>$O/train.err | grep '^{' > $O/train.json; cut -c1-330 $O/train.json; tail -3 $O/train.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ts && rocprofv3 --kernel-trace --output-format csv -d /tmp/ts -o t -- python $R/scripts/bench_train.py --graph --steps 12 --warmup 4 > /dev/null 2>&1
python $R/scripts/trace_one_step.py $(find /tmp/ts -name "*kernel_trace.csv" | head -1) > $R/$O/train_one_step.csv
cd $R
grep -i "wgm\|Cijk" $O/train_one_step.csv | cut -c1-120
