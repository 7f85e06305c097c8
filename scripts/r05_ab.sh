#!/bin/bash
# round 5, GPU call AB: thin forward kernel on bf16 instead of fp16 (hi, lo) operand pairs (timing experiment: is the 256-wide forward power-bound like the planes GEMM?)
cd /root/repo; mkdir -p gpurun_out/r05ab; O=gpurun_out/r05ab
for rep in 1 2; do
echo "== HEAD (fp16 pairs)"; timeout 300 python scripts/bench_thin.py 20 2>&1 | grep -A3 '"thin_fwd' | grep -E 'thin_fwd|"us"'
echo "== thin_bf16"; HOS_LIB_PATH=build/variants/thin_bf16/libhosrender.so timeout 300 python scripts/bench_thin.py 20 2>&1 | grep -A3 '"thin_fwd' | grep -E 'thin_fwd|"us"'
done | tee $O/thin_bf16.txt
