"""Critical path of a replayed step from gpurun_out/timeline_<rays>.csv (scripts/trace_step_timeline.sh):
per step (delimited by the first kernel of the captured graph): wall time, busy time per queue, time with 0 / 1 / 2 queues busy, the
largest gaps, and the kernels that run while the OTHER queue is idle (they are the serial part).
  python scripts/analyse_timeline.py gpurun_out/timeline_512.csv"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["start_ns"]), int(r["end_ns"]), r["queue"], r["kernel"]) for r in rows]
ev.sort()
# step boundaries: the optimiser kernel ends a step
ends = [i for i, e in enumerate(ev) if "adam_multi" in e[3]]
if len(ends) < 3:
    print("fewer than 3 steps in the trace"); sys.exit(0)
lo, hi = ends[-3] + 1, ends[-2] + 1          # one full step between two optimiser launches
step = ev[lo:hi]
t0 = step[0][0]
t1 = max(e[1] for e in step)
print(f"step: {len(step)} launches, {(t1 - t0) / 1e3:.1f} us from the first start to the last end; previous adam end -> first start {(step[0][0] - ev[lo - 1][1]) / 1e3:.1f} us")
queues = sorted({e[2] for e in step})
for q in queues:
    b = sum(e[1] - e[0] for e in step if e[2] == q)
    print(f"  queue {q}: {sum(1 for e in step if e[2] == q)} launches, busy {b / 1e3:.1f} us")
# sweep: number of queues busy
pts = []
for s, e, q, k in step:
    pts.append((s, 1, q, k)); pts.append((e, -1, q, k))
pts.sort()
busy = defaultdict(int)
tprev = t0
hist = defaultdict(int)
solo = defaultdict(int)
active = {}
for t, d, q, k in pts:
    n = sum(1 for v in busy.values() if v > 0)
    hist[n] += t - tprev
    if n == 1:
        for kk in list(active.values()):
            solo[kk.split("(")[0][:70]] += t - tprev
    tprev = t
    busy[q] += d
    if d > 0:
        active[(q, k, t)] = k
    else:
        for key in list(active):
            if key[0] == q and key[1] == k:
                del active[key]; break
for n in sorted(hist):
    print(f"  {n} queue(s) busy: {hist[n] / 1e3:.1f} us")
print("kernels running alone (other queue idle), us per step:")
for k, v in sorted(solo.items(), key=lambda kv: -kv[1])[:25]:
    print(f"   {v / 1e3:8.1f}  {k}")
