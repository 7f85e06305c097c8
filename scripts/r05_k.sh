#!/bin/bash
# round 5, GPU call K: un-profiled timeline (in-graph stamp kernels) of the stage-3 step, two streams and one, 512 and 4096 rays
cd /root/repo; mkdir -p gpurun_out/r05k; O=gpurun_out/r05k
for r in 512 4096; do for ts in 1 0; do
  HOS_TWO_STREAMS=$ts timeout 600 python scripts/diag_overlap.py $r 20 2>&1 | grep -v amdgpu.ids | tail -16
done; done | tee $O/overlap.txt
