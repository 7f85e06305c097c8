#!/bin/bash
# HBM traffic (PMC) of the thin-layer kernels; separate passes, no tracing domains combined with --pmc.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_thin
mkdir -p $OUT
cd /tmp
run() { rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -o t -- python $R/scripts/bench_thin.py 3 > /dev/null 2>&1; }
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
python - <<PY
import csv, glob, json
def mean(path, tag):
    vals = []
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % path):
        for r in csv.DictReader(open(f)):
            if tag in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals) if vals else None
out = {}
M = 262144
for name, tag, alg in (("thin_fwd[262144,256,256]", "thin_gemm_kernel<16, false>", M * 256 * 8), ("thin_dgrad[262144,256,256]", "thin_gemm_kernel<16, true>", M * 256 * 12),
                       ("mlp_bwd_fused[262144,128,128]", "mlp_bwd_kernel<4, 4, true>", M * 128 * 12), ("wgrad_tr[256,256,262144]", "mlp_bwd_kernel<8, 8, false>", M * 256 * 8)):
    f, w = mean("fetch", tag), mean("write", tag)
    if f is not None and w is not None:
        # 2 x FETCH_SIZE: gfx950 wide-load correction (MI355X_MICROARCH.md "HBM"); both counters in KB
        out[name] = {"fetch_size_kb": f, "write_size_kb": w, "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024, "algorithmic_bytes": alg}
json.dump(out, open("$OUT/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
