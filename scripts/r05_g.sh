#!/bin/bash
# round 5, GPU call G: IPE encoder, round-4 output stage (variant library) against the LDS-staged whole-line stores, same box
cd /root/repo; mkdir -p gpurun_out/r05g; O=gpurun_out/r05g
for rep in 1 2; do
echo "== old"; HOS_LIB_PATH=build/variants/oldenc/libhosrender.so timeout 300 python scripts/bench_encode.py 2>&1 | grep encode
echo "== new"; timeout 300 python scripts/bench_encode.py 2>&1 | grep encode
done | tee $O/encode_ab.txt
timeout 1800 python -m pytest tests/test_gpu_scene.py -x -q -m gpu 2>&1 | tail -30 | tee $O/pytest.txt
