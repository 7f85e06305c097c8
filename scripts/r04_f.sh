#!/bin/bash
# round 4, GPU call F: kernel stats of the group backward microbench.  usage: r04_f.sh [lib path]
ROOT=/root/repo; mkdir -p $ROOT/gpurun_out/r04f; O=$ROOT/gpurun_out/r04f
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $ROOT/scripts/bench_chainbwd.py 262144 10 > $O/run.log 2>&1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
grep "rows" $O/run.log
find $O/prof -name '*.csv' ! -name '*kernel_stats.csv' -delete; find $O/prof -name '*.db' -delete
