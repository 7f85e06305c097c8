for c in 0 1 0 1; do HOS_MLP_CHAIN=$c python bench.py --primary stage2 --only-primary --steps 20 --warmup 3 --no-kernel-events 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.readline()); print('chain', os.environ.get('C'), d['value'], d['ms_per_step'])"; done
