O=gpurun_out/r05u; mkdir -p $O
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
python scripts/tune_gemms_train.py --batches 32 16 24 48 64 --out $O/tunableop_train.csv > $O/tune_train.json 2> $O/tune_train.err; cat $O/tune_train.json; tail -2 $O/tune_train.err
python scripts/merge_gemm_tables.py hotrack_amd/tunableop_gfx950.csv $O/tunableop_train.csv --out $O/tunableop_gfx950.csv
cp $O/tunableop_gfx950.csv hotrack_amd/tunableop_gfx950.csv
for b in 32 16 48 64; do python scripts/bench_train.py --graph --batch $b 2>/dev/null | grep '^{' | cut -c1-260; done | tee $O/train_after.txt
