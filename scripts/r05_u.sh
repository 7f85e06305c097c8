#!/bin/bash
# round 5, GPU call U: what does data movement cost the planes GEMM in (power-limited) time?  Every LDS fragment read issued twice / every
# LDS-DMA request issued twice, random and zero-filled operands, [131072,1024,1024]; then the stage-3 step's un-profiled timeline
cd /root/repo; mkdir -p gpurun_out/r05u; O=gpurun_out/r05u
run() { GM=131072 GONLY="fwd(f16),dgrad(bits),wgrad" timeout 300 python scripts/bench_gemmp.py 20 2>&1 | grep planes; }
for rep in 1 2; do
echo "== HEAD"; run
for v in dup_lds dup_dma; do echo "== $v"; HOS_LIB_PATH=build/variants/$v/libhosrender.so run; done
done | tee $O/dup.txt
echo "== HEAD zero-filled"; GZERO=1 run | tee -a $O/dup.txt
for v in dup_lds dup_dma; do echo "== $v zero-filled"; GZERO=1 HOS_LIB_PATH=build/variants/$v/libhosrender.so run; done | tee -a $O/dup.txt
for g in 256 128; do echo "=== HOS_GEMMP_GRID=$g"; HOS_GEMMP_GRID=$g timeout 600 python scripts/diag_overlap.py 4096 10 2>&1 | grep -v amdgpu.ids | tail -16; done | tee $O/overlap.txt
echo "=== one stream"; HOS_TWO_STREAMS=0 timeout 600 python scripts/diag_overlap.py 4096 10 2>&1 | grep -v amdgpu.ids | tail -16 | tee -a $O/overlap.txt
