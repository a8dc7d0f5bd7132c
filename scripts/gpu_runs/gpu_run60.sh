python scripts/soak.py 600 20260930 2>&1 | tail -2
