#!/bin/bash
# round 5, GPU call H: scene / launcher tests after the fixes; background tests on the restored encoder
cd /root/repo; mkdir -p gpurun_out/r05h; O=gpurun_out/r05h
timeout 1800 python -m pytest tests/test_gpu_scene.py tests/test_gpu_bkgd.py -x -q -m gpu 2>&1 | tail -30 | tee $O/pytest.txt
