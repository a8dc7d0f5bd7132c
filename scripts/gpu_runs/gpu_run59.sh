python -m pytest tests -q -m gpu 2>&1 | tail -3
