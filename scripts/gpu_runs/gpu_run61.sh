for r in 64 256 64 256; do echo "ln_linear rows <= $r: $(HOTRACK_LN_LINEAR_MAX_ROWS=$r python scripts/bench_latency.py 2>/dev/null | tail -1 | cut -c1-200)"; done
