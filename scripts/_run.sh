python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python bench.py --only-primary 2>&1 | grep '^{'
