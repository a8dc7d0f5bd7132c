R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06f; mkdir -p $O; cd $R
python scripts/tune_gemms_train.py --batches 32 16 24 48 64 --out $O/tuned_all.csv 2>/dev/null | tail -1
python scripts/merge_gemm_tables.py hotrack_amd/tunableop_gfx950.csv $O/tuned_all.csv.train_only --out $O/tunableop_gfx950.csv
cp $O/tunableop_gfx950.csv hotrack_amd/tunableop_gfx950.csv
for i in 1 2; do python scripts/bench_train.py --graph 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
for b in 16 64; do python scripts/bench_train.py --graph --batch $b 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
