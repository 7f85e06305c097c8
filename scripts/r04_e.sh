#!/bin/bash
# round 4, GPU call E: group backward of the non-rigid MLP -- parity against the layer launches, microbench
cd /root/repo; mkdir -p gpurun_out/r04e; O=gpurun_out/r04e
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q -k "group_backward or folded_chain_backward" > $O/test.log 2>&1; tail -15 $O/test.log
timeout 300 python scripts/bench_chainbwd.py 262144 20 2>&1 | grep -v amdgpu.ids | tee $O/bench_262144.txt
timeout 300 python scripts/bench_chainbwd.py 524288 20 2>&1 | grep -v amdgpu.ids | tee $O/bench_524288.txt
