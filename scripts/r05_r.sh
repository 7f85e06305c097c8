#!/bin/bash
# round 5, GPU call R: is the planes GEMM power-bound enough that fewer CUs cost (almost) nothing?  Persistent grid 256 / 240 / 224 / 208 / 192 / 160 / 128
# workgroups (FWD, DGRAD), WGRAD with 16 / 15 / 14 / 13 / 12 / 10 / 8 splits of its 16 tiles, [131072,1024,1024], random operands
cd /root/repo; mkdir -p gpurun_out/r05r; O=gpurun_out/r05r
for g in 256 240 224 208 192 160 128; do
  echo "== grid $g (wgrad splits $((g/16)))"
  HOS_GEMMP_GRID=$g HOS_WGRAD_SPLITS=$((g/16)) GM=131072 GONLY="fwd(f16),dgrad(bits),wgrad" timeout 300 python scripts/bench_gemmp.py 20 2>&1 | grep planes
done | tee $O/grid_sweep.txt
