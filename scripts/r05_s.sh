#!/bin/bash
# round 5, GPU call S: the two-stream stage-3 step with the planes GEMMs on fewer than all CUs (HOS_GEMMP_GRID), so that the human
# branch's kernels find free CUs while a GEMM runs; 4096 and 512 rays; one stream as the reference
cd /root/repo; mkdir -p gpurun_out/r05s; O=gpurun_out/r05s
for r in 4096 512; do
for g in 256 224 192 160 128; do
  echo -n "rays $r grid $g two streams: "
  HOS_GEMMP_GRID=$g timeout 600 python bench.py --primary stage3 --only-primary --rays $r --steps 20 --warmup 3 --no-kernel-events 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"
done
echo -n "rays $r grid 256 one stream: "
HOS_TWO_STREAMS=0 timeout 600 python bench.py --primary stage3 --only-primary --rays $r --steps 20 --warmup 3 --no-kernel-events 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"
done | tee $O/grid_step.txt
