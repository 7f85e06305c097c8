#!/bin/bash
ROOT=/root/repo; mkdir -p $ROOT/gpurun_out/r04s; O=$ROOT/gpurun_out/r04s
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $ROOT/scripts/bench_lpips.py 20 > $O/run.log 2>&1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel us per iteration", tot / 23 / 1e3)
for r in rows[:14]:
    print(f"{r['Name'][:95]:95s} {int(r['Calls'])/23:5.1f}/it avg {float(r['AverageNs'])/1e3:8.1f} us  tot/it {float(r['TotalDurationNs'])/23/1e3:8.1f}")
PY
grep LPIPS $O/run.log
find $O/prof -name '*.csv' ! -name '*kernel_stats.csv' -delete; find $O/prof -name '*.db' -delete
