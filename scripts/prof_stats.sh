#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/prof_stats.sh <tag> <command...>
# rocprofv3 --kernel-trace --stats of <command>; keeps only the kernel_stats CSV (gpurun_out/<tag>_kernel_stats.csv):
# the raw trace of a few hundred thousand launches would exceed gpurun's 64 MiB merge limit.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
args=()
for a in "$@"; do if [ -e "$R/$a" ]; then args+=("$R/$a"); else args+=("$a"); fi; done   # rocprofv3 runs from /tmp
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- "${args[@]}" > $R/gpurun_out/${tag}.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${tag}_kernel_stats.csv
head -40 $R/gpurun_out/${tag}_kernel_stats.csv | cut -c1-230
