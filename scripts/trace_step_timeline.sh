#!/bin/bash
# Kernel timeline of the replayed stage-3 step (rocprofv3 --kernel-trace): scripts/trace_step_timeline.sh <rays>
# -> gpurun_out/timeline_<rays>.csv (start, end, stream/queue, name), analysed by scripts/analyse_timeline.py
RAYS=${1:-512}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/tl_$RAYS
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/tl_$RAYS -- python $ROOT/bench.py --primary stage3 --only-primary --rays $RAYS --steps 12 --warmup 3 --no-kernel-events > $ROOT/gpurun_out/tl_$RAYS.log 2>&1
f=$(find $ROOT/gpurun_out/tl_$RAYS -name '*kernel_trace.csv' | head -1)
python - "$f" $ROOT/gpurun_out/timeline_$RAYS.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-4000:]                      # the last steps only
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["start_ns", "end_ns", "queue", "stream", "kernel"])
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        w.writerow([int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r.get("Queue_Id", ""), r.get("Stream_Id", ""), r["Kernel_Name"][:90]])
PY
rm -rf $ROOT/gpurun_out/tl_$RAYS
tail -1 $ROOT/gpurun_out/tl_$RAYS.log | cut -c1-200
