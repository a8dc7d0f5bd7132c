O=gpurun_out/r05y; mkdir -p $O
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
python scripts/tune_gemms_train.py --batches 32 16 24 48 64 --out $O/tunableop_train.csv > $O/tune_train.json 2> $O/tune_train.err; cat $O/tune_train.json | cut -c1-300
python scripts/merge_gemm_tables.py hotrack_amd/tunableop_gfx950.csv $O/tunableop_train.csv --out $O/tunableop_gfx950.csv
cp $O/tunableop_gfx950.csv hotrack_amd/tunableop_gfx950.csv
for b in 32 16 48 64; do python scripts/bench_train.py --graph --batch $b 2>/dev/null | grep '^{' | cut -c100-230; done | tee $O/train_after.txt
(timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_full.py tests/test_gpu_gemm_table.py -x -q) 2>&1 | tail -3
