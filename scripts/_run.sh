python -m pytest tests -x -q -m gpu -k "bkgd or stage1 or stage3 or configs or edge or stress or fullsize" 2>&1 | tail -4
for h in 1 0 1 0; do
export HOS_ROWDOT_HEADS=$h
python bench.py --only-primary 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('heads=$h stage3', d['ms_per_step'], d['value'])"
python bench.py --primary stage1 --only-primary 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('heads=$h stage1', d['ms_per_step'], d['value'])"
done
