#!/bin/bash
# round 5, GPU call I: LPIPS with the fixed-order split-K (tight bound + bit-reproducibility), its timing
cd /root/repo; mkdir -p gpurun_out/r05i; O=gpurun_out/r05i
timeout 900 python -m pytest tests/test_gpu_lpips.py -q -m gpu 2>&1 | tail -30 | tee $O/pytest.txt
timeout 300 python scripts/bench_lpips.py 2>&1 | tail -8 | tee $O/bench_lpips.txt
