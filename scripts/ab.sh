#!/bin/bash
# Step-level A/B of one runtime switch, alternated REPS times so that box drift shows up as spread, not as a result:
#   scripts/ab.sh HOS_PERSIST_GRID "256 384 512" [REPS] [-- extra bench.py args]
# prints ms/step of stage 2 (2048 rays), stage 3 (4096 rays) and stage 3 at 512 rays (one rank's share at N = 8) per value.
VAR=$1; VALUES=$2; REPS=${3:-3}; shift 3 2>/dev/null; [ "$1" = "--" ] && shift
cd ${GRAFT_REPO_ROOT:-/root/repo}
t() { timeout 600 python bench.py --only-primary --steps 20 --warmup 3 --no-kernel-events "$@" 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.readline())['ms_per_step'],3))"; }
for rep in $(seq $REPS); do for v in $VALUES; do
  echo "$VAR=$v: stage2 $(env $VAR=$v bash -c "$(declare -f t); t --primary stage2 $*")  stage3 $(env $VAR=$v bash -c "$(declare -f t); t $*")  stage3@512 $(env $VAR=$v bash -c "$(declare -f t); t --rays 512 $*")"
done; done
