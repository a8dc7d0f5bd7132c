set -x
O=gpurun_out/r05i; mkdir -p $O
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
(time timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_full.py tests/test_gpu_ops.py -x -q) > $O/pytest_train.log 2>&1; tail -4 $O/pytest_train.log
python scripts/bench_train.py --graph > $O/train.json 2> $O/train.err; cat $O/train.json
python scripts/bench_train.py --graph --batch 16 2>/dev/null | grep '^{' > $O/train_b16_before.json; cut -c1-200 $O/train_b16_before.json
python scripts/bench_train.py --graph --batch 48 2>/dev/null | grep '^{' > $O/train_b48_before.json; cut -c1-200 $O/train_b48_before.json
python scripts/bench_train.py --graph --batch 64 2>/dev/null | grep '^{' > $O/train_b64_before.json; cut -c1-200 $O/train_b64_before.json
python scripts/tune_gemms_train.py --batches 16 48 64 24 --out $O/tunableop_train.csv > $O/tune_train.json 2> $O/tune_train.err; cat $O/tune_train.json; tail -2 $O/tune_train.err
python scripts/tune_gemms.py --batches 2 4 16 24 32 48 96 128 --out $O/tunableop_infer.csv > $O/tune_infer.json 2> $O/tune_infer.err; cat $O/tune_infer.json; tail -2 $O/tune_infer.err
python scripts/merge_gemm_tables.py $O/tunableop_train.csv $O/tunableop_infer.csv --out $O/tunableop_gfx950.csv
cp hotrack_amd/tunableop_gfx950.csv $O/tunableop_shipped_before.csv
cp $O/tunableop_gfx950.csv hotrack_amd/tunableop_gfx950.csv
python scripts/bench_train.py --graph 2>/dev/null | grep '^{' > $O/train_after.json; cut -c1-200 $O/train_after.json
python scripts/bench_train.py --graph --batch 16 2>/dev/null | grep '^{' > $O/train_b16_after.json; cut -c1-200 $O/train_b16_after.json
python scripts/bench_train.py --graph --batch 48 2>/dev/null | grep '^{' > $O/train_b48_after.json; cut -c1-200 $O/train_b48_after.json
python scripts/bench_train.py --graph --batch 64 2>/dev/null | grep '^{' > $O/train_b64_after.json; cut -c1-200 $O/train_b64_after.json
python bench.py --no-cpu-baseline > $O/bench_after.json 2> $O/bench_after.err; cut -c1-400 $O/bench_after.json
