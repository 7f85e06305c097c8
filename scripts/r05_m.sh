#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r05m
timeout 900 python scripts/diag_shard.py 2>&1 | grep -v amdgpu.ids | tail -30 | tee gpurun_out/r05m/diag_shard.txt
