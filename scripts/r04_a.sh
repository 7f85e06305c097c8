#!/bin/bash
# round 4, GPU call A: multistate parity, persistent planes GEMM A/B (microbench + parity), baseline bench line
cd /root/repo; mkdir -p gpurun_out/r04a; O=gpurun_out/r04a
python -m pytest tests/test_gpu_multistate.py -x -q > $O/multistate.log 2>&1; tail -3 $O/multistate.log
HOS_GEMMP_PERSIST=256 python -m pytest tests/test_gpu_bkgd.py tests/test_gpu_round2_kernels.py -x -q > $O/bkgd_persist.log 2>&1; tail -3 $O/bkgd_persist.log
for P in 0 256 512; do
  echo "== PERSIST=$P [131072,1024,1024]"; HOS_GEMMP_PERSIST=$P GM=131072 python scripts/bench_gemmp.py 20 2>&1 | grep -v "^$" | tee -a $O/gemmp_131072_p$P.txt
done
for P in 0 256; do
  echo "== PERSIST=$P [262144,256,256]"; HOS_GEMMP_PERSIST=$P GM=262144 GN=256 GK=256 python scripts/bench_gemmp.py 30 2>&1 | grep "planes" | tee -a $O/gemmp_prop_p$P.txt
  echo "== PERSIST=$P [262144,256,576]"; HOS_GEMMP_PERSIST=$P GM=262144 GN=256 GK=576 python scripts/bench_gemmp.py 30 2>&1 | grep "planes fwd" | tee -a $O/gemmp_prop576_p$P.txt
done
for P in 0 256; do
  echo "== nostore variant PERSIST=$P"; HOS_LIB_PATH=build/variants/nostore/libhosrender.so HOS_GEMMP_PERSIST=$P GM=131072 python scripts/bench_gemmp.py 20 2>&1 | grep "planes" | tee -a $O/gemmp_nostore_p$P.txt
done
for P in 0 256; do
  echo "== bench.py PERSIST=$P"; HOS_GEMMP_PERSIST=$P python bench.py --no-cpu-baseline --no-torch-baseline --no-infer > $O/bench_p$P.json 2> $O/bench_p$P.err; python - <<PY
import json
d=json.loads(open("$O/bench_p$P.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d.get("roofline"), {k:(v.get("ms_per_step") if isinstance(v,dict) else v) for k,v in d.get("stages",{}).items()})
PY
done
