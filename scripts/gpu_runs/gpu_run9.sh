set -x
O=gpurun_out/r05j; mkdir -p $O
R=$PWD
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
(time timeout 1500 python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python scripts/bench_train.py --graph > $O/train.json 2> $O/train.err; cat $O/train.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ts && rocprofv3 --kernel-trace --output-format csv -d /tmp/ts -o t -- python $R/scripts/bench_train.py --graph --steps 12 --warmup 4 > /dev/null 2>&1
python $R/scripts/trace_one_step.py $(find /tmp/ts -name "*kernel_trace.csv" | head -1) > $R/$O/train_one_step.csv
cd $R
python scripts/probes/tg_bench.py --iters 10 > $O/tg_bench.json 2>/dev/null
head -1 $O/train_one_step.csv
