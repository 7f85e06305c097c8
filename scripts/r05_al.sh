#!/bin/bash
# round 5, GPU call AL: second step-level knob sweep (decoder GEMM split-K / few-row thresholds, gemm3 prefetch distance, narrow-wgrad width, head row-dots, compact first deconv layer)
cd /root/repo; mkdir -p gpurun_out/r05al; O=gpurun_out/r05al
t() { timeout 600 python bench.py --only-primary --steps 20 --warmup 3 --no-kernel-events "$@" 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.readline())['ms_per_step'],3))"; }
run() { echo "$1: stage3 $(env $1 bash -c "$(declare -f t); t")  stage2 $(env $1 bash -c "$(declare -f t); t --primary stage2")  stage3@512 $(env $1 bash -c "$(declare -f t); t --rays 512")"; }
for rep in 1 2; do
run "HOS_X=default"
for kv in HOS_SPLITK_TARGET=128 HOS_SPLITK_TARGET=512 HOS_SPLITK_FEW_M=32 HOS_SPLITK_FEW_M=128 HOS_FEWROW_M=32 HOS_FEWROW_M=128 HOS_GEMM_PF=2 HOS_GEMM_PF=4 HOS_WGRAD_NARROW_MAX=128 HOS_ROWDOT_HEADS=0 HOS_SAMPLE_WARP_BWD_REUSE=0 HOS_DGRAD_SPLIT=1; do run $kv; done
done 2>&1 | tee $O/knobs2.txt
