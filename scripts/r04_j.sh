#!/bin/bash
# round 4, GPU call J: comm library on the GPU, decoder-gradient factors, fresh-items leg
cd /root/repo; mkdir -p gpurun_out/r04j; O=gpurun_out/r04j
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_human.py tests/test_gpu_stage2.py -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 900 python bench.py --no-cpu-baseline --no-torch-baseline --no-infer > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04j/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"])
for k,v in d["stages"].items(): print(k, {a:v.get(a) for a in ("value","ms_per_step","vs_resident_batch","error")})
p=json.load(open("gpurun_out/parity_counts.json"))
for k,v in p.items():
    if "grad" in k: print(k, v)
PY
