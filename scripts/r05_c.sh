#!/bin/bash
# round 5, GPU call C: start / end wall clock of EVERY workgroup of a [131072,1024,1024] planes forward (persistent and per-tile launch)
cd /root/repo; mkdir -p gpurun_out/r05c; O=gpurun_out/r05c
for p in 1 0; do
  echo "=== trace2 persist=$p"
  HOS_LIB_PATH=build/variants/trace2/libhosrender.so HOS_GEMMP_PERSIST=$p timeout 300 python scripts/trace_gemmp2.py 2>&1 | grep -v amdgpu.ids | grep -v "tile " | grep -v "XCC"
done | tee $O/trace2_all.txt
