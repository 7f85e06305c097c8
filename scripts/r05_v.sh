#!/bin/bash
# round 5, GPU call V: every LDS fragment read of the planes GEMM issued twice (random / zero-filled operands)
cd /root/repo; mkdir -p gpurun_out/r05v; O=gpurun_out/r05v
run() { GM=131072 GONLY="fwd(f16),fwd(bf16),dgrad(bits),wgrad" timeout 300 python scripts/bench_gemmp.py 20 2>&1 | grep -v amdgpu.ids | tail -6; }
for rep in 1 2; do
echo "== HEAD"; run
echo "== dup_lds"; HOS_LIB_PATH=build/variants/dup_lds/libhosrender.so run
done | tee $O/dup_lds.txt
echo "== dup_lds zero-filled"; GZERO=1 HOS_LIB_PATH=build/variants/dup_lds/libhosrender.so run | tee -a $O/dup_lds.txt
