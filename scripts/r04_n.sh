#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r04n
for r in 512 4096; do timeout 600 python scripts/exp_multigraph.py $r 30 2>&1 | grep -v amdgpu.ids | tail -5; done | tee gpurun_out/r04n/multigraph.txt
