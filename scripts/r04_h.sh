#!/bin/bash
# round 4, GPU call H: step-level A/B of the group backward (HOS_CHAIN_BWD=0/1): stage 3 at 4096 rays, stage 2
cd /root/repo; mkdir -p gpurun_out/r04h; O=gpurun_out/r04h
for cb in 0 1 0 1; do
  for st in stage3 stage2; do
    HOS_CHAIN_BWD=$cb python bench.py --primary $st --only-primary --steps 20 --warmup 5 --no-kernel-events 2>/dev/null | tail -1 > $O/${st}_cb$cb.json
    python - <<PY
import json
d=json.loads(open("$O/${st}_cb$cb.json").read())
print("$st chain_bwd=$cb", d["ms_per_step"], d["value"])
PY
  done
done
