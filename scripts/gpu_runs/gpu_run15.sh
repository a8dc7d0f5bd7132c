O=gpurun_out/r05p; mkdir -p $O
R=$PWD
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
(time timeout 1500 python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python scripts/bench_train.py --graph 2>$O/train.err | grep '^{' > $O/train.json; cut -c1-330 $O/train.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ts && rocprofv3 --kernel-trace --output-format csv -d /tmp/ts -o t -- python $R/scripts/bench_train.py --graph --steps 12 --warmup 4 > /dev/null 2>&1
python $R/scripts/trace_one_step.py $(find /tmp/ts -name "*kernel_trace.csv" | head -1) > $R/$O/train_one_step.csv
cd $R
grep -i "sa_layer1_stats\|segment_sum\|wgm" $O/train_one_step.csv | cut -c1-100
