#!/bin/bash
# round 5, GPU call L: sharded volume decoder (2 ranks on one GPU over gloo), HosComm in-graph exchange, decoder regressions
cd /root/repo; mkdir -p gpurun_out/r05l; O=gpurun_out/r05l
timeout 2400 python -m pytest tests/test_gpu_comm.py tests/test_gpu_dist.py tests/test_gpu_clip.py tests/test_gpu_stage2.py -x -q -m gpu 2>&1 | tail -40 | tee $O/pytest.txt
