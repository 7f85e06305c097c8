#!/bin/bash
# A/B of one environment variable with explicit values on stage 3 (4096 rays) and stage 1: scripts/ab_env1.sh VAR A B
VAR=$1; A=$2; B=$3
for i in 1 2 3; do
  for f in $A $B; do
    for st in stage3 stage1; do
      env $VAR=$f python bench.py --primary $st --only-primary --no-kernel-events --steps 60 --warmup 10 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$f $st', round(d['ms_per_step'],3), round(d['value']))"
    done
  done
done
