#!/bin/bash
# usage (GPU box, repo root): scripts/prof_window.sh <tag> <frac> <steps_in_window> <command...>
# rocprofv3 --kernel-trace of <command>, summarised over the last <frac> of the traced span (steady state) by
# scripts/trace_window.py -> gpurun_out/<tag>_window.csv.  The raw trace stays on the box (too big for gpurun's merge limit).
tag=$1; frac=$2; steps=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
args=()
for a in "$@"; do if [ -e "$R/$a" ]; then args+=("$R/$a"); else args+=("$a"); fi; done   # rocprofv3 runs from /tmp
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -o $tag -- "${args[@]}" > $R/gpurun_out/${tag}.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
python $R/scripts/trace_window.py "$f" --frac $frac --steps-in-window $steps > $R/gpurun_out/${tag}_window.csv
head -75 $R/gpurun_out/${tag}_window.csv | cut -c1-200
