#!/bin/bash
# Effective shader clock of the planes GEMM on random and on zero-filled operands: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / dispatch duration.
# scripts/pmc_clock.sh  ->  gpurun_out/pmc_clock/clock.json
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_clock; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for z in 0 1; do
  if [ $z = 1 ]; then export GZERO=1; else unset GZERO; fi
  GM=131072 GONLY="fwd(bf16),dgrad(bits),wgrad" rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d $OUT/z$z -o t -- python $R/scripts/bench_gemmp.py 5 > $OUT/z$z.log 2>&1
done
python - <<PY
import csv, glob, json, collections
out = {}
for z in (0, 1):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/z%d/*counter_collection.csv" % z):
        rows = list(csv.DictReader(open(f)))
        if rows: cols = rows[0].keys()
        for r in rows:
            if "gemmp_kernel" not in r["Kernel_Name"]: continue
            tag = "wgrad" if "3, " in r["Kernel_Name"] or "ELi3E" in r["Kernel_Name"] else ("dgrad" if "ELi2E" in r["Kernel_Name"] else "fwd")
            dur = None
            if "Start_Timestamp" in r and "End_Timestamp" in r:
                dur = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
            acc[tag][r["Counter_Name"]].append((float(r["Counter_Value"]), dur))
    res = {}
    for tag, cs in acc.items():
        g = cs.get("GRBM_GUI_ACTIVE", [])
        if not g: continue
        cyc = sum(v for v, _ in g) / len(g) / 8.0
        durs = [d for _, d in g if d]
        d = sum(durs) / len(durs) if durs else None
        res[tag] = {"gui_active_cycles_per_xcd": cyc, "dispatch_seconds": d, "effective_clock_ghz": (cyc / d / 1e9) if d else None,
                    "insts_mfma": (sum(v for v, _ in cs.get("SQ_INSTS_MFMA", [])) / max(1, len(cs.get("SQ_INSTS_MFMA", []))))}
    out["zero_filled" if z else "random"] = res
out["how"] = "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA of scripts/bench_gemmp.py at [131072,1024,1024] (GZERO=1: zero-filled operands); clock = GRBM_GUI_ACTIVE / 8 XCDs / (End_Timestamp - Start_Timestamp) of the same dispatch"
json.dump(out, open("$OUT/clock.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
head -2 $OUT/z0/*counter_collection.csv | cut -c1-400
