#!/bin/bash
# round 5, GPU call AK: HOS_PERSIST_GRID (workgroups of the persistent backward kernels of the human branch's warp / LBS stages), step level, three alternations
cd /root/repo; mkdir -p gpurun_out/r05ak; O=gpurun_out/r05ak
t() { timeout 600 python bench.py --only-primary --steps 20 --warmup 3 --no-kernel-events "$@" 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.readline())['ms_per_step'],3))"; }
for rep in 1 2 3; do for g in 256 384 512 768; do
  echo "HOS_PERSIST_GRID=$g: stage2 $(HOS_PERSIST_GRID=$g t --primary stage2)  stage3 $(HOS_PERSIST_GRID=$g t)  stage3@512 $(HOS_PERSIST_GRID=$g t --rays 512)"
done; done 2>&1 | tee $O/persist_grid.txt
