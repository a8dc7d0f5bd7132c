O=gpurun_out/r05x; mkdir -p $O
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
for thr in 0 4096 8192; do
  echo "small rows <= $thr: $(HOTRACK_SMALL_STACK_ROWS=$thr HOTRACK_TUNE_GEMMS=1 HOTRACK_GEMM_CACHE=$O/cache_$thr.csv python scripts/bench_train.py --graph 2>/dev/null | grep '^{' | cut -c100-200)"
done
(HOTRACK_SMALL_STACK_ROWS=8192 timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_full.py -x -q) 2>&1 | tail -3
