python -m pytest tests -x -q -m gpu -k "bkgd or stage1 or stage3 or edge or stress or fullsize or plane" 2>&1 | tail -3
python bench.py --primary stage1 --only-primary 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stage1', d['ms_per_step'], d['value'])"
python bench.py --only-primary 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stage3', d['ms_per_step'], d['value'])"
