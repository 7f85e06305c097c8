#!/bin/bash
# HBM traffic of a whole training step (PMC, separate passes, no tracing domains): FETCH_SIZE and WRITE_SIZE summed over every
# kernel of `bench.py --primary stageN --only-primary --no-graph`, per step.  Usage: scripts/pmc_step_traffic.sh stage2 [steps]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ST=${1:-stage2}; STEPS=${2:-4}; WARM=2
OUT=$R/gpurun_out/pmc_step_$ST
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o t -- python $R/bench.py --primary $ST --only-primary --no-graph --no-kernel-events --steps $STEPS --warmup $WARM > $OUT/$c.log 2>&1
done
python - <<PY
import csv, glob, json, collections, re
steps = $STEPS + $WARM
tot = {}
per = collections.defaultdict(lambda: [0.0, 0.0])
for i, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    s = 0.0
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            v = float(r["Counter_Value"]); s += v
            k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))[:60]
            per[k][i] += v
    tot[c] = s
# KB; 2 x FETCH_SIZE: gfx950 wide-load correction (MI355X_MICROARCH.md)
hbm = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / steps
line = [l for l in open("$OUT/FETCH_SIZE.log") if l.startswith("{")][-1]
d = json.loads(line)
rays = d["config"]["global_rays"]
out = {"stage": "$ST", "rays": rays, "steps_profiled": steps, "fetch_kb_per_step": tot["FETCH_SIZE"] / steps, "write_kb_per_step": tot["WRITE_SIZE"] / steps,
       "hbm_bytes_per_step": hbm, "hbm_bytes_per_ray": hbm / rays,
       "top_kernels_bytes_per_step": {k: (2 * v[0] + v[1]) * 1024 / steps for k, v in sorted(per.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1]))[:14]},
       "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --primary $ST --only-primary --no-graph; (2 x FETCH + WRITE) x 1024 / steps (first steps include one-time packing)"}
json.dump(out, open("$OUT/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
