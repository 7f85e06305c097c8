#!/bin/bash
# round 5, GPU call AA: bench line + full GPU suite on the tree with the new IPE encoder
cd /root/repo; mkdir -p gpurun_out/r05aa; O=gpurun_out/r05aa
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05aa/bench.json') if l.startswith('{')][-1])
print('stage3 ms', d['ms_per_step'], 'roofline', d['roofline']['kernel'], d['roofline']['avg_us'], d['roofline']['frac'], d['roofline']['traffic'])
print('stage2 ms', d['stages']['stage2']['ms_per_step'], 'stage1 ms', d['stages']['stage1']['ms_per_step'], 'infer rays/s', d['stages']['infer_1080p']['value'])
print(d['cpu_baseline'])
PY
timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee $O/pytest.txt
