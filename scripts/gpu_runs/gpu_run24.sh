O=gpurun_out/r05w; mkdir -p $O
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
(timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_full.py -x -q) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python scripts/probes/sa_layer1_bench.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print({k:(v['sa_layer1_us'],v['sa_layer1_stats_us'],v['bn_stats_us']) for k,v in d.items()})"
python scripts/bench_train.py --graph 2>/dev/null | grep '^{' | cut -c1-260
