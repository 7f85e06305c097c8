#!/bin/bash
# A/B of this round's human-branch changes (folds + unpredicated thin kernels) at 512 and 4096 rays, alternating runs
for i in 1 2 3; do
  for f in 1 0; do
    for r in 512 4096; do
      env HOS_CHAIN_FOLD=$f HOS_CNL_FOLD=$f HOS_THIN_FAST=$f HOS_WGRAD_FAST=$f python bench.py --primary stage3 --only-primary --rays $r --no-kernel-events --steps 60 --warmup 10 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new=$f rays=$r', round(d['ms_per_step'],3), round(d['value']))"
    done
  done
done
