R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for w in 1024 512 256 2048; do export PN2_BN_WGS=$w; echo "bn reduce workgroups $w: $(python scripts/bench_train.py --graph 2>/dev/null | grep -o '"ms_per_step": [0-9.]*') $(python scripts/bench_train.py --graph 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; done
