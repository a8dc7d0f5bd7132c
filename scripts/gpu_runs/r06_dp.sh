R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06c; mkdir -p $O; cd $R
for i in 1 2; do
python scripts/bench_train.py --graph 2>/dev/null | grep '^{' > $O/train_graph_$i.json; echo "single      $(grep -o '"ms_per_step": [0-9.]*' $O/train_graph_$i.json)"
python scripts/bench_train.py --graph --dp-selftest --segments 1 2>/dev/null | grep '^{' > $O/dp_seg1_$i.json; echo "dp 1 seg    $(grep -o '"ms_per_step": [0-9.]*' $O/dp_seg1_$i.json) $(grep -o '"allreduce_us": [0-9.]*' $O/dp_seg1_$i.json)"
python scripts/bench_train.py --graph --dp-selftest 2>/dev/null | grep '^{' > $O/dp_seg2_$i.json; echo "dp 2 seg ov $(grep -o '"ms_per_step": [0-9.]*' $O/dp_seg2_$i.json)"
python scripts/bench_train.py --graph --dp-selftest --no-overlap 2>/dev/null | grep '^{' > $O/dp_seg2_noov_$i.json; echo "dp 2 seg in order $(grep -o '"ms_per_step": [0-9.]*' $O/dp_seg2_noov_$i.json)"
done
python scripts/probes/train_step_parts.py --dp 2>&1 | grep "ms / iteration"
python -m pytest tests/test_gpu_train_full.py tests/test_gpu_train.py -x -q 2>&1 | tail -5
