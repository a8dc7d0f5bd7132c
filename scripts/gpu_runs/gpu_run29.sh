export HOTRACK_DATA_ROOT=/tmp/hotrack_data
for b in 32 16 48 64; do python scripts/bench_train.py --graph --batch $b 2>/dev/null | grep '^{' | cut -c100-230; done
