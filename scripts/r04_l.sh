#!/bin/bash
# round 4, GPU call L: soaks with NaN-poisoned allocations (stage 2 and stage 3, 300 steps each) + the GPU suite under HOS_POISON=1
cd /root/repo; mkdir -p gpurun_out/r04l; O=gpurun_out/r04l
timeout 900 python scripts/soak_poison.py 2 300 1 2048 > $O/soak_s2.log 2>&1; tail -3 $O/soak_s2.log
timeout 900 python scripts/soak_poison.py 3 200 2 2048 > $O/soak_s3.log 2>&1; tail -3 $O/soak_s3.log
HOS_POISON=1 timeout 1500 python -m pytest tests -m gpu -x -q > $O/poison_suite.log 2>&1; tail -4 $O/poison_suite.log
