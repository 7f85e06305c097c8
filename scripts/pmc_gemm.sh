#!/bin/bash
# PMC passes for the GEMM micro-benchmark (separate passes, no tracing domains combined with --pmc).
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_gemm
mkdir -p $OUT
cd /tmp
MODE=${1:-bf16x3}; WHICH=${2:-fwd}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_${MODE}_${WHICH} -o t -- python $R/scripts/bench_gemm.py $MODE $WHICH 10 > $OUT/trace_${MODE}_${WHICH}.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_${MODE}_${WHICH} -o t -- python $R/scripts/bench_gemm.py $MODE $WHICH 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_${MODE}_${WHICH} -o t -- python $R/scripts/bench_gemm.py $MODE $WHICH 5 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc_${MODE}_${WHICH} -o t -- python $R/scripts/bench_gemm.py $MODE $WHICH 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $OUT/sq_${MODE}_${WHICH} -o t -- python $R/scripts/bench_gemm.py $MODE $WHICH 5 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm_${MODE}_${WHICH} -o t -- python $R/scripts/bench_gemm.py $MODE $WHICH 5 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*_${MODE}_${WHICH}")):
    for f in glob.glob(d + "/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(d.split("/")[-1], k, "n=%d mean=%.4g" % (len(v), sum(v) / len(v)))
PY
cat $OUT/trace_${MODE}_${WHICH}.log | tail -3
