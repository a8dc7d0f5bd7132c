O=gpurun_out/r05v; mkdir -p $O
export HOTRACK_DATA_ROOT=/tmp/hotrack_data
python scripts/bench_train.py --graph 2>/dev/null | grep '^{' | cut -c1-260
HOTRACK_WGM_SIDE_STREAM=1 python scripts/bench_train.py --graph 2>$O/side.err | grep '^{' | cut -c1-260; tail -2 $O/side.err
python scripts/bench_train.py --graph 2>/dev/null | grep '^{' | cut -c1-260
HOTRACK_WGM_SIDE_STREAM=1 python scripts/bench_train.py --graph 2>/dev/null | grep '^{' | cut -c1-260
