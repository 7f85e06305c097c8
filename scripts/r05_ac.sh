#!/bin/bash
# round 5, GPU call AC: thin forward with the two waves of a SIMD half a tile apart (HOS_THIN_PP=1 default / 0 = lock-step), timing + tests
cd /root/repo; mkdir -p gpurun_out/r05ac; O=gpurun_out/r05ac
for rep in 1 2; do
for pp in 0 1; do echo "== HOS_THIN_PP=$pp"; HOS_THIN_PP=$pp timeout 300 python scripts/bench_thin.py 20 2>&1 | grep -A3 '"thin_fwd' | grep -E 'thin_fwd|"us"'; done
done | tee $O/thin_pp.txt
timeout 1500 python -m pytest tests/test_gpu_round2_kernels.py tests/test_gpu_human.py tests/test_gpu_stage2.py -x -q -m gpu 2>&1 | tail -8 | tee $O/pytest.txt
