#!/bin/bash
# round 4, GPU call U: wave-priority variants of the planes GEMM K loop at [131072,1024,1024] (two rounds each)
cd /root/repo; mkdir -p gpurun_out/r04u; O=gpurun_out/r04u
for rep in 1 2; do
for v in "" gp_noprio gp_static1 gp_static3 gp_flip_static1; do
  if [ -z "$v" ]; then L=""; else L=build/variants/$v/libhosrender.so; fi
  echo "== ${v:-default} (round $rep)"
  env ${L:+HOS_LIB_PATH=$L} GM=131072 GONLY="fwd(2fmt),fwd(f16),dgrad(bits),wgrad" python scripts/bench_gemmp.py 20 2>&1 | grep planes
done; done | tee $O/prio.txt
