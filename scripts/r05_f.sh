#!/bin/bash
# round 5, GPU call F: new IPE encoder output stage (timing + background parity tests), launcher eval/render tests, stress tests
cd /root/repo; mkdir -p gpurun_out/r05f; O=gpurun_out/r05f
timeout 300 python scripts/bench_encode.py 2>&1 | grep encode | tee $O/encode.txt
timeout 1800 python -m pytest tests/test_gpu_bkgd.py tests/test_gpu_scene.py tests/test_gpu_stress.py tests/test_gpu_multistate.py tests/test_gpu_eval.py -x -q -m gpu 2>&1 | tail -30 | tee $O/pytest.txt
