#!/bin/bash
# Everything profiles/r05_* of the FINAL tree is made from, in one GPU call: scripts/collect_r05.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05final; O=gpurun_out/r05final
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_r05_final.json 2> $O/bench.err
for r in 512 1024 2048 4096; do
  timeout 600 python bench.py --primary stage3 --only-primary --rays $r --steps 20 --warmup 3 --no-kernel-events 2>/dev/null | tail -1
done > $O/r05_strong_scaling_sweep.jsonl
for r in 512 4096; do HOS_MODEL_SHARD=8 timeout 600 python bench.py --primary stage3 --only-primary --rays $r --steps 20 --warmup 3 --no-kernel-events 2>/dev/null | tail -1; done > $O/r05_model_shard8.jsonl
HOS_TWO_STREAMS=0 bash scripts/prof_step.sh 4096 r05_stage3_one_stream
bash scripts/prof_step.sh 4096 r05_stage3
bash scripts/prof_step.sh 512 r05_stage3_512rays
PRIMARY=stage2 bash scripts/prof_step.sh 2048 r05_stage2
PRIMARY=stage1 bash scripts/prof_step.sh 1024 r05_stage1
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05final/bench_r05_final.json') if l.startswith('{')][-1])
print('stage3 ms', d['ms_per_step'], d['value'], 'roofline', d['roofline']['kernel'], d['roofline']['avg_us'], d['roofline']['frac'], d['roofline']['traffic'], 'x torch', d.get('speedup_vs_torch_rocm'))
for k,v in d['stages'].items(): print(k, v.get('ms_per_step'), v.get('value'), v.get('speedup_vs_torch_rocm'))
for k in d['kernels'][:6]: print('   ', k['kernel'], k['launches'], round(k['avg_us'],1), round(k['tflops'],1))
print(d['cpu_baseline']['value'], d['cpu_baseline']['thread_sweep_rays_per_s'], d['cpu_baseline']['step_seconds'])
for f in ('r05_strong_scaling_sweep.jsonl','r05_model_shard8.jsonl'):
    for l in open('gpurun_out/r05final/'+f):
        x=json.loads(l); print(f, x['config']['global_rays'], round(x['ms_per_step'],3))
PY
