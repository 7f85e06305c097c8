#!/bin/bash
# round 5, GPU call AD: per-tile timeline of the planes forward at the proposal MLP's hidden-layer shape [262144,256,256] (4 tiles per CU) and at [65536,256,256] (stage 1: one tile per CU)
cd /root/repo; mkdir -p gpurun_out/r05ad; O=gpurun_out/r05ad
for m in 262144 65536 4194304; do
echo "=== M=$m N=256 K=256"
GM=$m GN=256 GK=256 HOS_LIB_PATH=build/variants/trace2/libhosrender.so timeout 300 python scripts/trace_gemmp2.py 2>&1 | grep -v amdgpu.ids | grep -v "XCC"
done | tee $O/trace_small.txt
