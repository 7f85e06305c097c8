#!/bin/bash
# round 4, GPU call I: new tests (stage-2 scene items, group backward) + the default bench line with the new legs
cd /root/repo; mkdir -p gpurun_out/r04i; O=gpurun_out/r04i
timeout 900 python -m pytest tests/test_gpu_scene.py tests/test_gpu_chain.py -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04i/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"])
print(json.dumps(d.get("cpu_baseline"))[:1500])
for k,v in d["stages"].items(): print(k, {a:v.get(a) for a in ("value","ms_per_step","vs_resident_batch","error")})
PY
