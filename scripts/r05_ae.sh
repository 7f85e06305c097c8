#!/bin/bash
# round 5, GPU call AE: kernel timeline of the replayed stage-3 step under rocprofv3 --kernel-trace: how much of a step is gaps between launches?
cd /root/repo; mkdir -p gpurun_out/r05ae; O=gpurun_out/r05ae
for r in 4096 512; do for ts in 0 1; do
  echo "=== rays $r two_streams $ts"
  HOS_TWO_STREAMS=$ts bash scripts/trace_step_timeline.sh $r > /dev/null 2>&1
  python scripts/analyse_timeline.py gpurun_out/timeline_$r.csv | head -12
  python - gpurun_out/timeline_$r.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["start_ns"]), int(r["end_ns"]), r["kernel"]) for r in rows)
ends = [i for i, e in enumerate(ev) if "adam_multi" in e[2]]
lo, hi = ends[-3] + 1, ends[-2] + 1
step = ev[lo:hi]
# gaps between consecutive launches in start order (meaningful for one stream)
gaps = [max(0, step[i + 1][0] - max(e[1] for e in step[:i + 1])) for i in range(len(step) - 1)]
gaps_sorted = sorted(gaps)
print(f"  launches {len(step)}, idle between launches: total {sum(gaps) / 1e3:.1f} us, median {gaps_sorted[len(gaps) // 2] / 1e3:.2f} us, p90 {gaps_sorted[int(len(gaps) * 0.9)] / 1e3:.2f} us")
PY
done; done 2>&1 | tee $O/timeline.txt
