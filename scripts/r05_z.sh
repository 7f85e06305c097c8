#!/bin/bash
# round 5, GPU call Z: IPE encoder with loop-invariant column assignment (library sine / own sine / own sine + exp / hardware, timing) against the per-item loop;
# background parity tests on the own-sine + own-exp build
cd /root/repo; mkdir -p gpurun_out/r05z; O=gpurun_out/r05z
for rep in 1 2; do
echo "== per-item loop, library sinf (in-tree library)"; timeout 300 python scripts/bench_encode.py 2>&1 | grep encode | grep -v "+bf16"
for v in enc_head2 enc_own1 enc_own2 enc_fast; do echo "== invariant columns, $v"; HOS_LIB_PATH=build/variants/$v/libhosrender.so timeout 300 python scripts/bench_encode.py 2>&1 | grep encode | grep -v "+bf16"; done
done | tee $O/encode.txt
for v in enc_head2 enc_own2; do
echo "=== parity tests on $v"
HOS_LIB_PATH=build/variants/$v/libhosrender.so timeout 1500 python -m pytest tests/test_gpu_bkgd.py tests/test_gpu_fullsize.py tests/test_gpu_multistate.py tests/test_gpu_round2_kernels.py -q -m gpu 2>&1 | tail -5
cp gpurun_out/parity_counts.json $O/parity_counts_$v.json
done | tee $O/pytest.txt
