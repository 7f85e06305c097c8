#!/bin/bash
# round 5, GPU call N: where the tree stands at session start: bench line, one-stream and two-stream kernel stats of the stage-3 step,
# stage-2 stats, 512..4096 sweep
cd /root/repo; mkdir -p gpurun_out/r05n; O=gpurun_out/r05n
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
HOS_TWO_STREAMS=0 bash scripts/prof_step.sh 4096 r05_stage3_one_stream
bash scripts/prof_step.sh 4096 r05_stage3
PRIMARY=stage2 bash scripts/prof_step.sh 2048 r05_stage2
bash scripts/prof_step.sh 512 r05_stage3_512rays
for r in 512 1024 2048 4096; do timeout 600 python bench.py --primary stage3 --only-primary --rays $r --steps 20 --warmup 3 --no-kernel-events 2>/dev/null | tail -1; done > $O/sweep.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r05n/sweep.jsonl'):
    d=json.loads(l); print(d['config']['global_rays'], round(d['ms_per_step'],3))
PY
