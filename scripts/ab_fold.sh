#!/bin/bash
# A/B of a byte-reduction switch (default HOS_CNL_FOLD; pass another variable name as $1) on the stage-2 and stage-3 steps;
# three alternating runs each.
VAR=${1:-HOS_CNL_FOLD}
mkdir -p gpurun_out
for i in 1 2 3; do
  for f in 1 0; do
    for st in stage2 stage3; do
      env $VAR=$f python bench.py --primary $st --only-primary --no-kernel-events --steps 60 --warmup 10 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$f', '$st', d['ms_per_step'], d['value'])"
    done
  done
done
