O=gpurun_out/r05t; mkdir -p $O
R=$PWD
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
(timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_full.py tests/test_gpu_wgrad.py tests/test_gpu_fused.py -x -q) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python scripts/bench_train.py --graph 2>$O/train.err | grep '^{' > $O/train.json; cut -c1-330 $O/train.json
HOTRACK_TUNE_GEMMS=1 HOTRACK_GEMM_CACHE=$O/gemm_cache.csv python scripts/bench_train.py --graph 2>$O/train_tune.err | grep '^{' > $O/train_tuned.json; cut -c1-330 $O/train_tuned.json
ls -la $O/gemm_cache.csv; wc -l $O/gemm_cache.csv
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ts && HOTRACK_GEMM_CACHE=$R/$O/gemm_cache.csv rocprofv3 --kernel-trace --output-format csv -d /tmp/ts -o t -- python $R/scripts/bench_train.py --graph --steps 12 --warmup 4 > /dev/null 2>&1
python $R/scripts/trace_one_step.py $(find /tmp/ts -name "*kernel_trace.csv" | head -1) > $R/$O/train_one_step.csv
head -1 $R/$O/train_one_step.csv
