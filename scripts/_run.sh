python -m pytest tests -x -q -m gpu -k "adam or stage2 or stage3 or dist or bkgd or launcher or soak" 2>&1 | tail -3
python bench.py --only-primary 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stage3', d['ms_per_step'], d['value'], d['final_loss'])"
python bench.py --primary stage2 --only-primary 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stage2', d['ms_per_step'], d['value'], d['final_loss'])"
