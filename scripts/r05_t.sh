#!/bin/bash
# round 5, GPU call T: un-profiled timeline (in-graph stamp kernels) of the stage-3 step at 4096 rays: two streams with the planes GEMMs on 256 / 128
# workgroups, one stream
cd /root/repo; mkdir -p gpurun_out/r05t; O=gpurun_out/r05t
for g in 256 128; do echo "=== HOS_GEMMP_GRID=$g"; HOS_GEMMP_GRID=$g timeout 600 python scripts/diag_overlap.py 4096 10 2>&1 | grep -v amdgpu.ids | tail -16; done | tee $O/overlap.txt
echo "=== one stream"; HOS_TWO_STREAMS=0 timeout 600 python scripts/diag_overlap.py 4096 10 2>&1 | grep -v amdgpu.ids | tail -16 | tee -a $O/overlap.txt
