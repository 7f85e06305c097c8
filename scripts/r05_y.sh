#!/bin/bash
# round 5, GPU call Y: IPE encoder, own sine with the range decided per workgroup
cd /root/repo; mkdir -p gpurun_out/r05y; O=gpurun_out/r05y
for rep in 1 2; do
echo "== HEAD"; timeout 300 python scripts/bench_encode.py 2>&1 | grep encode | grep -v "+bf16"
for v in enc_fast enc_own1 enc_own2; do echo "== $v"; HOS_LIB_PATH=build/variants/$v/libhosrender.so timeout 300 python scripts/bench_encode.py 2>&1 | grep encode | grep -v "+bf16"; done
done | tee $O/encode.txt
