( time python -m pytest tests -x -q -m gpu 2>&1 | tail -3 ) 2>&1
python scripts/soak.py 150 777 2>&1 | tail -3
