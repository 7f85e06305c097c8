#!/bin/bash
# round 5, GPU call AG: persistent (default) vs one-workgroup-per-tile planes GEMM launches at the STEP level, same box, alternating; stage 3 (two streams and one), stage 1, 512 rays
cd /root/repo; mkdir -p gpurun_out/r05ag; O=gpurun_out/r05ag
t() { timeout 600 python bench.py --only-primary --steps 20 --warmup 3 --no-kernel-events "$@" 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.readline())['ms_per_step'],3))"; }
for rep in 1 2 3; do
for p in 1 0; do
  echo "persist=$p stage3 4096: $(HOS_GEMMP_PERSIST=$p t)   one stream: $(HOS_GEMMP_PERSIST=$p HOS_TWO_STREAMS=0 t)   512 rays: $(HOS_GEMMP_PERSIST=$p t --rays 512)   stage1: $(HOS_GEMMP_PERSIST=$p t --primary stage1)"
done; done | tee $O/persist_step_ab.txt
