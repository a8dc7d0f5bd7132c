for r in 4096 8192 0 4096; do echo "small stack rows <= $r: $(HOTRACK_SMALL_STACK_ROWS=$r python scripts/bench_train.py --graph 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; done
