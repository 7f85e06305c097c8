#!/bin/bash
# round 4, GPU call M: decoder backward inside the human branch's backward (one rank) vs split off behind the join: A/B
cd /root/repo; mkdir -p gpurun_out/r04m; O=gpurun_out/r04m
for sp in 1 0 1 0; do
  for cfg in "stage3 4096" "stage3 512" "stage2 2048"; do
    set -- $cfg
    HOS_BENCH_SPLIT_DECODER=$sp python bench.py --primary $1 --only-primary --rays $2 --steps 20 --warmup 5 --no-kernel-events 2>/dev/null | tail -1 > $O/t.json
    python - <<PY
import json
d=json.loads(open("$O/t.json").read())
print("$1 rays $2 split=$sp", round(d["ms_per_step"],3), "loss", d["final_loss"])
PY
  done
done | tee $O/ab.txt
