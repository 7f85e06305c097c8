#!/bin/bash
# round 5, GPU call P: planes GEMM [131072,1024,1024], same box: r04 tree / HEAD (one-tile WGRAD address fix) / MFMA order variants /
# ablations (no DMA, no MFMA: results invalid), on random and on zero-filled operands
cd /root/repo; mkdir -p gpurun_out/r05p; O=gpurun_out/r05p
L="fwd(f16),dgrad(bits),wgrad"
run() { GM=131072 GONLY="$L" timeout 300 python scripts/bench_gemmp.py 20 2>&1 | grep planes; }
for rep in 1 2; do
  echo "== r04 tree (round $rep)"; (cd build/r04tree && run)
  echo "== HEAD (round $rep)"; run
  for v in order1 order2; do echo "== $v (round $rep)"; HOS_LIB_PATH=build/variants/$v/libhosrender.so run; done
done | tee $O/ab.txt
for v in abl_dma abl_mfma; do echo "== $v"; HOS_LIB_PATH=build/variants/$v/libhosrender.so run; echo "== $v zero-filled"; GZERO=1 HOS_LIB_PATH=build/variants/$v/libhosrender.so run; done | tee -a $O/ab.txt
for v in order1 order2; do echo "== $v zero-filled"; GZERO=1 HOS_LIB_PATH=build/variants/$v/libhosrender.so run; done | tee -a $O/ab.txt
