#!/bin/bash
# round 4, GPU call R: LPIPS inside a step (test) + the bench legs
cd /root/repo; mkdir -p gpurun_out/r04r; O=gpurun_out/r04r
timeout 900 python -m pytest tests/test_gpu_lpips.py -x -q 2>&1 | tail -6
timeout 900 python bench.py --no-cpu-baseline --no-torch-baseline --no-infer > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04r/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["traffic"])
for k,v in d["stages"].items(): print(k, {a:v.get(a) for a in ("value","ms_per_step","vs_resident_batch","vs_without_lpips","error")})
PY
