#!/bin/bash
# PMC passes for the planes-GEMM micro-benchmark (separate passes; no tracing domains combined with --pmc).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_gemmp
mkdir -p $OUT
cd /tmp
run() { rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -o t -- python $R/scripts/bench_gemmp.py 3 > /dev/null 2>&1; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"
run sq2 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
run sq3 "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS"
run grbm "GRBM_GUI_ACTIVE"
run tcc "TCC_HIT_sum TCC_MISS_sum"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*")):
    for f in glob.glob(d + "/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "gemmp_kernel" in r["Kernel_Name"]:
                key = r["Kernel_Name"].split("gemmp_kernel")[1][:22]
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(d.split("/")[-1], k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
# per-launch HBM traffic of the three [32768,1024,1024] planes kernels: 2 x FETCH_SIZE (gfx950 wide-load correction,
# MI355X_MICROARCH.md "HBM") + WRITE_SIZE, both reported in KB
import json
def mean(path, tag):
    vals = []
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % path):
        for r in csv.DictReader(open(f)):
            if "gemmp_kernel" in r["Kernel_Name"] and tag in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals) if vals else None
out = {}
for name, tag in (("gemmp_fwd", "ELi1E"), ("gemmp_dgrad", "ELi2E"), ("gemmp_wgrad", "3, ")):
    f, w = mean("fetch", tag), mean("write", tag)
    if f is not None and w is not None:
        out[name + "[32768x1024x1024]"] = {"fetch_size_kb": f, "write_size_kb": w, "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024}
json.dump(out, open("$OUT/traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
