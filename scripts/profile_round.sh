#!/bin/bash
# usage (GPU box, repo root): scripts/profile_round.sh rNN   -> gpurun_out/rNN_* (copy what is to be judged into profiles/)
#   rNN_bench.json               the default `python bench.py` line
#   rNN_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline`
#   rNN_bench_kernel_by_grid.csv the same trace per (kernel, grid)
#   rNN_pmc_traffic.json         FETCH_SIZE / WRITE_SIZE (separate --pmc passes, --kernel-trace only) -> HBM bytes per launch
#   rNN_pmc_mfma.txt             MfmaUtil of the SA kernels (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE: separate passes)
set -u
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R && python bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o b -- python $R/bench.py --no-cpu-baseline > $O/${tag}_stats.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/${tag}_bench_kernel_stats.csv
python $R/scripts/kernel_trace_by_grid.py $(find /tmp/prof_stats -name "*kernel_trace.csv" | head -1) > $O/${tag}_bench_kernel_by_grid.csv
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/bench.py --steps 3 --warmup 2 --min-time 0 --no-graph --no-cpu-baseline > $O/${tag}_pmc_$c.log 2>&1
done
python $R/scripts/pmc_summary.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/${tag}_pmc_traffic.json > /dev/null
python $R/scripts/pmc_extra.py $(find /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_GRBM_GUI_ACTIVE -name "*counter_collection.csv" | head -1) > $O/${tag}_pmc_mfma.txt 2>&1
ls -la $O | head -30
