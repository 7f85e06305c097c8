#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r04c; O=gpurun_out/r04c
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -15 $O/gpu_tests.log
python scripts/torch_ops_in_step.py 512 > $O/torch_ops_512.txt 2>&1; grep -v amdgpu $O/torch_ops_512.txt | head -60
for R in 512 4096; do
python bench.py --no-cpu-baseline --no-torch-baseline --no-infer --rays $R > $O/bench_$R.json 2> $O/bench_$R.err; python - <<PY
import json
d=json.loads(open("$O/bench_$R.json").read().strip().splitlines()[-1])
print($R, {k:d[k] for k in ("value","ms_per_step")}, {k:(v.get("ms_per_step") if isinstance(v,dict) else v) for k,v in d.get("stages",{}).items()})
PY
done
