#!/bin/bash
# round 4, GPU call V: end-of-round soaks of the captured steps (group backward inside): 600 replays each, loss must stay finite
cd /root/repo; mkdir -p gpurun_out/r04v; O=gpurun_out/r04v
for st in stage2 stage3; do
  python bench.py --primary $st --only-primary --steps 600 --warmup 5 --no-kernel-events 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$st 600 replays', round(d['ms_per_step'],3), 'ms, final loss', d['final_loss'])"
done | tee $O/soak.txt
timeout 900 python scripts/soak_graph.py 3 1500 2048 2>&1 | tail -3 | tee -a $O/soak.txt
