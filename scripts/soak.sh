#!/bin/bash
# End-of-round soaks of a tree:  scripts/soak.sh r06   -> gpurun_out/<round>soak/soak.txt (copy to profiles/<round>_soak.txt)
#   600 replays of the captured stage-1 / 2 / 3 steps, 1500 of soak_graph, 300 + 200 NaN-poisoned steps, the -m gpu suite under HOS_POISON=1
RND=${1:-r06}
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/${RND}soak; mkdir -p $O
for st in stage1 stage2 stage3; do
  python bench.py --primary $st --only-primary --steps 600 --warmup 5 --no-kernel-events 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$st 600 replays', round(d['ms_per_step'],3), 'ms, final loss', d['final_loss'])"
done | tee $O/soak.txt
timeout 900 python scripts/soak_graph.py 3 1500 2048 2>&1 | tail -2 | tee -a $O/soak.txt
timeout 900 python scripts/soak_poison.py 2 300 2>&1 | tail -2 | tee -a $O/soak.txt
timeout 900 python scripts/soak_poison.py 3 200 2>&1 | tail -2 | tee -a $O/soak.txt
HOS_POISON=1 timeout 2700 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2 | sed "s/^/HOS_POISON=1 pytest -m gpu: /" | tee -a $O/soak.txt
