#!/bin/bash
# round 5, GPU call O: planes GEMM at [131072,1024,1024], same box: round-4 tree (build/r04tree) against HEAD, interleaved; HEAD on
# zero-filled operands (what the same binary does when the power budget is not the limit); bf16-format forward of HEAD
cd /root/repo; mkdir -p gpurun_out/r05o; O=gpurun_out/r05o
L="fwd(2fmt),fwd(f16),dgrad(bits),wgrad"
for rep in 1 2 3; do
  echo "== r04 tree (round $rep)"; (cd build/r04tree && GM=131072 GONLY="$L" timeout 300 python scripts/bench_gemmp.py 20 2>&1 | grep planes)
  echo "== HEAD (round $rep)"; GM=131072 GONLY="$L" timeout 300 python scripts/bench_gemmp.py 20 2>&1 | grep planes
done | tee $O/ab_r04_head.txt
echo "== HEAD, zero-filled operands" | tee -a $O/ab_r04_head.txt
GZERO=1 GM=131072 GONLY="$L" timeout 300 python scripts/bench_gemmp.py 20 2>&1 | grep planes | tee -a $O/ab_r04_head.txt
echo "== HEAD, HOS_GEMMP_PERSIST=0" | tee -a $O/ab_r04_head.txt
HOS_GEMMP_PERSIST=0 GM=131072 GONLY="$L" timeout 300 python scripts/bench_gemmp.py 20 2>&1 | grep planes | tee -a $O/ab_r04_head.txt
