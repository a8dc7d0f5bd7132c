export HOTRACK_DATA_ROOT=/tmp/hotrack_data
for v in 0 6 0 6; do echo "tgf384=$v: $(PN2_TGF_384=$v python scripts/bench_train.py --graph 2>/dev/null | grep '^{' | cut -c100-200)"; done
(PN2_TGF_384=6 timeout 600 python -m pytest tests/test_gpu_train.py -x -q -k "stack") 2>&1 | tail -2
