#!/bin/bash
# generic A/B of one environment switch on the stage-3 step at 512 and 4096 rays: scripts/ab_env.sh VAR [on] [off]
VAR=$1; ON=${2:-1}; OFF=${3:-0}
for i in 1 2 3; do
  for f in $ON $OFF; do
    for r in 512 4096; do
      env $VAR=$f python bench.py --primary stage3 --only-primary --rays $r --no-kernel-events --steps 60 --warmup 10 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$f rays=$r', round(d['ms_per_step'],3), round(d['value']))"
    done
  done
done
