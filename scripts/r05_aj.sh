#!/bin/bash
# round 5, GPU call AJ: step-level check of the grid / split knobs of the persistent human-branch kernels and of the main on/off switches on
# the final tree (the persistent planes GEMM taught that an isolated launch's verdict need not be the step's): stage 3 (two streams) and stage 2
cd /root/repo; mkdir -p gpurun_out/r05aj; O=gpurun_out/r05aj
t() { timeout 600 python bench.py --only-primary --steps 20 --warmup 3 --no-kernel-events "$@" 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.readline())['ms_per_step'],3))"; }
run() { echo "$1: stage3 $(env $1 bash -c "$(declare -f t); t")  stage2 $(env $1 bash -c "$(declare -f t); t --primary stage2")"; }
for rep in 1 2; do
run "HOS_X=default"
for kv in HOS_PERSIST_GRID=128 HOS_PERSIST_GRID=512 HOS_PERSIST_GRID=1024 HOS_CHAIN_GRID=512 HOS_CHAIN_GRID=2048 HOS_MB_RSPLIT=1 HOS_MB_RSPLIT=4 HOS_CHAIN_BWD=0 HOS_THIN_FAST=0 HOS_WGRAD_WS=0 HOS_MLP_CHAIN=0 HOS_DEFER_REDUCE=0 HOS_EMBED_BWD_TILED=0; do run $kv; done
done 2>&1 | tee $O/knobs.txt
