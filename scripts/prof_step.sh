#!/bin/bash
# rocprofv3 kernel statistics of the replayed stage-3 step at a given ray count: scripts/prof_step.sh <rays> <tag>
# (summary -> gpurun_out/<tag>_kernel_stats.csv; copy into profiles/ to keep it)
RAYS=${1:-4096}; TAG=${2:-r03_stage3_${RAYS}}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -- python $ROOT/bench.py --primary ${PRIMARY:-stage3} --only-primary --rays $RAYS --steps 20 --warmup 3 --no-kernel-events > $ROOT/gpurun_out/prof_$TAG.log 2>&1
f=$(find $ROOT/gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1)
cp "$f" $ROOT/gpurun_out/${TAG}_kernel_stats.csv
find $ROOT/gpurun_out/prof_$TAG -name '*.csv' ! -name '*kernel_stats.csv' -delete
find $ROOT/gpurun_out/prof_$TAG -name '*.db' -delete
tail -1 $ROOT/gpurun_out/prof_$TAG.log | cut -c1-400
