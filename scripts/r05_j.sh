#!/bin/bash
# round 5, GPU call J: kernel timeline of the replayed stage-3 step at 512 rays (two queues) + its critical path; sweep 512..4096
cd /root/repo; mkdir -p gpurun_out/r05j; O=gpurun_out/r05j
bash scripts/trace_step_timeline.sh 512
python scripts/analyse_timeline.py gpurun_out/timeline_512.csv | tee $O/timeline_512.txt
for r in 512 1024 2048 4096; do timeout 600 python bench.py --primary stage3 --only-primary --rays $r --steps 20 --warmup 3 --no-kernel-events 2>/dev/null | tail -1; done > $O/sweep.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r05j/sweep.jsonl'):
    d=json.loads(l); print(d['config']['global_rays'], round(d['ms_per_step'],3))
PY
