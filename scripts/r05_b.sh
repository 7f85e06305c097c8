#!/bin/bash
# round 5, GPU call B: where does a [131072,1024,1024] planes GEMM launch spend its time?  Per-tile timeline of the persistent and the
# per-tile launch (HOS_TRACE2 build), and the two ablations (no DMA / no MFMA; results invalid) at both launch forms.
cd /root/repo; mkdir -p gpurun_out/r05b; O=gpurun_out/r05b
for p in 1 0; do for f in "" 1; do
  echo "=== trace2 persist=$p twofmt=${f:-0}"
  HOS_LIB_PATH=build/variants/trace2/libhosrender.so HOS_GEMMP_PERSIST=$p TWOFMT=$f timeout 300 python scripts/trace_gemmp2.py 2>&1 | grep -v amdgpu.ids
done; done | tee $O/trace2.txt
for v in abl_dma abl_mfma; do for p in 1 0; do
  echo "=== $v persist=$p"
  HOS_LIB_PATH=build/variants/$v/libhosrender.so HOS_GEMMP_PERSIST=$p GM=131072 GONLY="fwd(2fmt),fwd(f16),dgrad(bits),wgrad" timeout 300 python scripts/bench_gemmp.py 20 2>&1 | grep planes
done; done | tee $O/ablate.txt
