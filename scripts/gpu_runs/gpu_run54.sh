python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -2
for b in 16 48 64; do echo "batch $b: $(python scripts/bench_train.py --graph --batch $b 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; done
python scripts/soak.py 240 4242 2>&1 | tail -2
